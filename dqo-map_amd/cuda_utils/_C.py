"""`cuda_utils._C` — same surface as the reference's pybind module (two functions; DQO-MAP calls the first)
(/root/reference/submodules/cuda_utils/ext.cpp, cuda_utils.cu:17-62; caller SLAM/multiprocess/mapper.py:1034-1047).

accumulate_gaussian_error(H, W, P, color_error, depth_error, normal_error, color_index, depth_index, color_thr, depth_thr,
                          normal_thr, check_max) -> (gs_color_error[P,1], gs_depth_error[P,1], gs_normal_error[P,1], rescale_counter[P,1])
accumulate_gaussian_confidence(H, W, P, gaussian_index_map, gaussian_confidence_map)
                          -> (gs_confidence_max[P,1], gs_confidence_min[P,1], gs_confidence_mean[P,1])      (cuda_utils.cu:62-83; the
                          reference exports it but has no Python caller — provided because the extension's surface has it)
Backed by dqo_accumulate_gaussian_error / dqo_accumulate_gaussian_confidence in libdqoraster.so; GPU only.
"""
import torch

import _dqo_native as N


def accumulate_gaussian_error(H, W, P, screen_color_error, screen_depth_error, screen_normal_error, screen_color_index,
                              screen_depth_index, color_threshold, depth_threshold, normal_threshold, check_max):
    ts = (screen_color_error, screen_depth_error, screen_normal_error, screen_color_index, screen_depth_index)
    if not all(t.is_cuda for t in ts):
        raise RuntimeError("accumulate_gaussian_error needs GPU (ROCm) tensors; there is no CPU path.")
    for t in ts[:3]:
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected scalar type Float but found {t.dtype}")
    for t in ts[3:]:
        if t.dtype != torch.int32:
            raise RuntimeError(f"expected scalar type Int but found {t.dtype}")
    ce, de, ne, ci, di = (t.contiguous() for t in ts)
    if min(t.numel() for t in (ce, de, ne, ci, di)) < H * W:
        raise RuntimeError("error / index maps must hold H*W elements")
    dev = ce.device
    outs = [torch.empty((P, 1), dtype=torch.float32, device=dev) for _ in range(4)]
    if P == 0:
        return tuple(outs)
    with torch.cuda.device(dev):
        counters = None if check_max else torch.empty((2 * P,), dtype=torch.int32, device=dev)
        N.check(N.lib().dqo_accumulate_gaussian_error(int(H), int(W), int(P), N.ptr(ce), N.ptr(de), N.ptr(ne), N.ptr(ci), N.ptr(di),
                                                      float(color_threshold), float(depth_threshold), float(normal_threshold),
                                                      1 if check_max else 0, N.ptr(outs[0]), N.ptr(outs[1]), N.ptr(outs[2]),
                                                      N.ptr(outs[3]), N.ptr(counters), N.current_stream()))
    return tuple(outs)


def accumulate_gaussian_confidence(H, W, P, gaussian_index_map, gaussian_confidence_map):
    if not (gaussian_index_map.is_cuda and gaussian_confidence_map.is_cuda):
        raise RuntimeError("accumulate_gaussian_confidence needs GPU (ROCm) tensors; there is no CPU path.")
    if gaussian_confidence_map.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {gaussian_confidence_map.dtype}")
    if gaussian_index_map.dtype != torch.int32:
        raise RuntimeError(f"expected scalar type Int but found {gaussian_index_map.dtype}")
    idx, conf = gaussian_index_map.contiguous(), gaussian_confidence_map.contiguous()
    if min(idx.numel(), conf.numel()) < H * W:
        raise RuntimeError("index / confidence maps must hold H*W elements")
    dev = conf.device
    outs = [torch.empty((P, 1), dtype=torch.float32, device=dev) for _ in range(3)]
    if P == 0:
        return tuple(outs)
    with torch.cuda.device(dev):
        counter = torch.empty((P,), dtype=torch.int32, device=dev)
        N.check(N.lib().dqo_accumulate_gaussian_confidence(int(H), int(W), int(P), N.ptr(idx), N.ptr(conf), N.ptr(outs[0]), N.ptr(outs[1]),
                                                           N.ptr(outs[2]), N.ptr(counter), N.current_stream()))
    return tuple(outs)
