"""Tile-mask producers of DQO-MAP's mapping loop on MI355X.

Same names, argument meaning and return types as the reference helpers
    SLAM/utils.py:720-799            meanpool, pixelmask2tilemask, transmission2tilemask, colorerror2tilemask
    SLAM/multiprocess/mapper.py:930-988   Mapping.evaluate_render_range  (here a free function over the render outputs)
but one fused HIP kernel per mask (libdqoraster.so: dqo_tile_count_mask / dqo_transmission_mask / dqo_tile_color_error)
instead of pad + pool + compare chains of eager torch kernels.  The rasteriser's tiles are 16x16, and so is the only
stride the reference ever passes; other strides raise.  GPU only: there is no CPU path.
"""
import torch

import _dqo_native as N

_STRIDE = 16


def _check_stride(stride):
    if stride != _STRIDE:
        raise ValueError(f"stride {stride}: the tile-mask kernels are built for the rasteriser's 16x16 tiles")


def _grid(h, w):
    return (h + _STRIDE - 1) // _STRIDE, (w + _STRIDE - 1) // _STRIDE


def _tile_count(pixelmask):
    N.require_gpu(pixelmask)
    if not pixelmask.is_cuda:
        raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
    h, w = pixelmask.shape[:2]
    m = pixelmask if pixelmask.dtype == torch.uint8 else (pixelmask != 0).to(torch.uint8)
    m = m.contiguous()
    gy, gx = _grid(h, w)
    cnt = torch.empty((gy, gx), dtype=torch.int32, device=pixelmask.device)
    with torch.cuda.device(pixelmask.device):
        N.check(N.lib().dqo_tile_count_mask(w, h, N.ptr(m), N.ptr(cnt), N.current_stream()))
    return cnt


def pixelmask2tilemask(pixelmask, stride):
    """SLAM/utils.py:731-743: 1 where any pixel of the tile is set (max-pool of the zero-padded mask); int32 [gy, gx]."""
    _check_stride(stride)
    return (_tile_count(pixelmask) > 0).int()


def transmission2tilemask(pixelmask, stride, tile_mask_ratio=0.5):
    """SLAM/utils.py:752-763: 1 where more than tile_mask_ratio of the tile's 256 pixels are set; int32 [gy, gx]."""
    _check_stride(stride)
    return (_tile_count(pixelmask).float() / float(stride * stride) > tile_mask_ratio).int()


def meanpool(matrix, stride, padding_value=0):
    """SLAM/utils.py:720-729 for a [H, W] image: mean over stride x stride tiles of the padded image."""
    _check_stride(stride)
    h, w = matrix.shape[:2]
    gy, gx = _grid(h, w)
    pad = torch.full((gy * stride, gx * stride), float(padding_value), dtype=torch.float32, device=matrix.device)
    pad[:h, :w] = matrix
    return pad.view(gy, stride, gx, stride).sum((1, 3)) / float(stride * stride)


def color_error_tiles(render, gt):
    """color_error [H, W] = sum_c |render - gt| with the pixels whose rendered colour sums to 0 zeroed
    (mapper.py:949-956), and its 16x16 mean-pool [gy, gx] (meanpool of SLAM/utils.py:720-729), in one pass."""
    N.require_gpu(render, gt)
    if not render.is_cuda:
        raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
    _, h, w = render.shape
    render, gt = render.float().contiguous(), gt.float().contiguous()
    gy, gx = _grid(h, w)
    err = torch.empty((h, w), dtype=torch.float32, device=render.device)
    tsum = torch.empty((gy, gx), dtype=torch.float32, device=render.device)
    with torch.cuda.device(render.device):
        N.check(N.lib().dqo_tile_color_error(w, h, N.ptr(render), N.ptr(gt), N.ptr(err), N.ptr(tsum), N.current_stream()))
    return err, tsum / float(_STRIDE * _STRIDE)


def colorerror2tilemask(color_error, stride, top_ratio=0.4, _pooled=None):
    """SLAM/utils.py:766-799: the top_ratio share of tiles with the largest mean colour error; int32 [gy, gx].
    (`_pooled`: the mean-pooled error from color_error_tiles(), to skip the pooling pass.)"""
    _check_stride(stride)
    if _pooled is None:
        h, w = color_error.shape[:2]
        gy, gx = _grid(h, w)
        pad = torch.zeros((gy * stride, gx * stride), dtype=torch.float32, device=color_error.device)
        pad[:h, :w] = color_error
        _pooled = pad.view(gy, stride, gx, stride).sum((1, 3)) / float(stride * stride)
    sample_num = int(torch.numel(_pooled) * top_ratio)
    _, idx = torch.topk(_pooled.reshape(-1), k=sample_num)
    tile_mask = torch.zeros_like(_pooled, dtype=torch.int32)
    tile_mask.view(-1)[idx] = 1
    return tile_mask


def evaluate_render_range(T_map, render=None, gt=None, global_opt=False, sample_ratio=-1):
    """Mapping.evaluate_render_range (mapper.py:930-988) over the rasteriser's outputs: returns
    (render_mask bool [H, W], tile_mask int32 [gy, gx] or None, render_ratio 0-dim tensor)."""
    N.require_gpu(T_map)
    if not T_map.is_cuda:
        raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
    t = T_map.reshape(T_map.shape[-2], T_map.shape[-1]).float().contiguous()
    h, w = t.shape
    if global_opt and sample_ratio > 0:
        err, pooled = color_error_tiles(render, gt)
        tile_mask = colorerror2tilemask(err, _STRIDE, sample_ratio, _pooled=pooled)
        render_mask = tile_mask.bool().repeat_interleave(_STRIDE, 0).repeat_interleave(_STRIDE, 1)[:h, :w]
        return render_mask, tile_mask, render_mask.sum() / (h * w)
    gy, gx = _grid(h, w)
    mask = torch.empty((h, w), dtype=torch.uint8, device=t.device)
    cnt = torch.empty((gy, gx), dtype=torch.int32, device=t.device)
    total = torch.empty((1,), dtype=torch.int32, device=t.device)
    with torch.cuda.device(t.device):
        N.check(N.lib().dqo_transmission_mask(w, h, N.ptr(t), N.ptr(mask), N.ptr(cnt), N.ptr(total), N.current_stream()))
    render_mask = mask.bool()
    if global_opt:
        return render_mask, None, total[0] / (h * w)
    tile_mask = (cnt.float() / float(_STRIDE * _STRIDE) > 0.5).int()
    return render_mask, tile_mask, total[0] / (h * w)
