"""Opt-in fused pieces for callers that stay on the reference's API (drop-in op + their own torch optimiser): two autograd Functions
that replace eager torch op sequences of Mapping.loss_update (SLAM/multiprocess/mapper.py:799-905) by the HIP kernels of row f2 —

    loss = masked_mapping_loss(out, gt_color, gt_depth, render_mask)[0]            # mapper.py:836-875 with a render mask
    attach = fused_attach_loss(_scaling, _xyz, _rotation, init_stat)               # mapper.py:812-829
    (loss + attach).backward(); optimizer.step()                                    # unchanged

Each is a two-line change in mapper.py; values and gradients equal the eager statements (dqo_harness.mapping.mapping_loss /
attach_loss) to float rounding (tests/test_gpu_fused_ops.py).  And one optimiser:

    optimizer = DqoAdam(param_groups, lr=0.0, eps=1e-15)                            # gaussian_pointcloud.py:331-378, unchanged groups

a torch.optim.Optimizer with torch.optim.Adam's arithmetic whose step() is ONE launch over all groups (dqo_adam_multi).
GPU only: there is no CPU path.
"""
import ctypes

import torch

import _dqo_native as N
from dqo_harness import mapping


class _MaskedMappingLoss(torch.autograd.Function):
    """dqo_map_loss_fwd_bwd as an autograd Function: loss values AND the gradient images come out of the forward's two launches; the
    backward only scales them by the incoming gradient of the total."""

    @staticmethod
    def forward(ctx, render, depth, depth_index, gt_color, gt_depth, render_mask, color_weight, depth_weight, add_depth_thres):
        lib = N.lib()
        N.require_gpu(render, depth, depth_index, gt_color, gt_depth, render_mask)
        if not render.is_cuda:
            raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
        f = lambda t: t.detach().float().contiguous()
        render, depth, gt_color, gt_depth = f(render), f(depth), f(gt_color), f(gt_depth)
        depth_index = depth_index.to(torch.int32).contiguous()
        mask = None if render_mask is None else render_mask.to(torch.uint8).contiguous()
        H, W = render.shape[-2], render.shape[-1]
        dev = render.device
        loss = torch.empty(8, dtype=torch.float32, device=dev)
        dC, dD = torch.empty_like(render), torch.empty_like(depth)
        ws = torch.empty(lib.dqo_map_loss_workspace_bytes(), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.dqo_map_loss_fwd_bwd(W, H, N.ptr(render), N.ptr(depth), N.ptr(depth_index), N.ptr(gt_color), N.ptr(gt_depth),
                                             N.ptr(mask), float(color_weight), float(depth_weight), float(add_depth_thres), N.ptr(loss),
                                             N.ptr(dC), N.ptr(dD), N.ptr(ws), ws.numel(), N.current_stream()))
        ctx.save_for_backward(dC, dD)
        ctx.mark_non_differentiable(loss)
        ctx.set_materialize_grads(False)  # (no zero tensor for the gradient of `loss`, which nothing differentiates through)
        return loss[0].clone(), loss

    @staticmethod
    def backward(ctx, g_total, _g_loss):
        if g_total is None:
            return (None,) * 9
        dC, dD = ctx.saved_tensors
        return g_total * dC, g_total * dD, None, None, None, None, None, None, None


class _Ssim(torch.autograd.Function):
    """utils/loss_utils.py:60-100 (ssim, window 11, size_average=True) in three launches: the value and d ssim / d img1 come out of the forward."""

    @staticmethod
    def forward(ctx, img1, img2):
        lib = N.lib()
        N.require_gpu(img1, img2)
        if not img1.is_cuda:
            raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
        a, b = img1.detach().float().contiguous(), img2.detach().float().contiguous()
        if a.dim() != 3 or a.shape[0] != 3 or a.shape != b.shape:
            raise RuntimeError(f"fused_ssim: two [3,H,W] images expected, got {tuple(a.shape)} and {tuple(b.shape)}")
        H, W, dev = a.shape[1], a.shape[2], a.device
        out = torch.empty(2, dtype=torch.float32, device=dev)
        grad = torch.empty_like(a)
        ws = torch.empty(lib.dqo_map_ssim_workspace_bytes(W, H), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            # weight -1: the term is -(1 - ssim) = ssim - 1, its gradient d ssim / d img1
            N.check(lib.dqo_map_ssim_fwd_bwd(W, H, N.ptr(a), N.ptr(b), -1.0, N.ptr(out), N.ptr(grad), 0, None, N.ptr(ws), ws.numel(),
                                             N.current_stream()))
        ctx.save_for_backward(grad)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None


def fused_ssim(img1, img2):
    """ssim(img1, img2) of utils/loss_utils.py:60-100 for [3,H,W] images, differentiable w.r.t. img1 (img2 = the ground truth)."""
    return _Ssim.apply(img1, img2)


def masked_mapping_loss(out, gt_color, gt_depth, render_mask, add_depth_thres=0.1, color_weight=mapping.COLOR_WEIGHT,
                        depth_weight=mapping.DEPTH_WEIGHT):
    """mapping.mapping_loss (mapper.py:836-875: 0.8 L1 colour + 1.0 depth L1 in two launches; the SSIM term is skipped when a render
    mask is given, B14, and added by fused_ssim — three launches — when render_mask is None).  `out` = the dict of mapping.render / Renderer.render.  Returns (total, parts) like mapping_loss."""
    total, loss = _MaskedMappingLoss.apply(out["render"], out["depth"], out["depth_index_map"], gt_color, gt_depth, render_mask,
                                           color_weight, depth_weight, add_depth_thres)
    if render_mask is None:  # mapper.py:839-845: the SSIM term exists only without a render mask
        ssim_loss = 1 - fused_ssim(out["render"], gt_color)
        total = total + mapping.SSIM_WEIGHT * ssim_loss
        return total, dict(total_loss=total.detach(), color_loss=loss[1], depth_loss=loss[2], ssim_loss=ssim_loss.detach())
    return total, dict(total_loss=loss[0], color_loss=loss[1], depth_loss=loss[2], ssim_loss=loss[3])


class _AttachLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, xyz, rotation, scaling0, xyz0, rotation0, attach_mask, attach_count):
        lib = N.lib()
        N.require_gpu(scaling, xyz, rotation, scaling0, xyz0, rotation0, attach_mask)
        if not xyz.is_cuda:
            raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
        f = lambda t: t.detach().float().contiguous()
        s, x, q, s0, x0, q0 = f(scaling), f(xyz), f(rotation), f(scaling0), f(xyz0), f(rotation0)
        P, dev = x.shape[0], x.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        gs, gx, gq = torch.empty_like(s), torch.empty_like(x), torch.empty_like(q)
        ws = torch.empty(lib.dqo_map_attach_workspace_bytes(P), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            N.check(lib.dqo_map_attach_loss_fwd_bwd(P, N.ptr(s), N.ptr(x), N.ptr(q), N.ptr(s0), N.ptr(x0), N.ptr(q0), N.ptr(attach_mask),
                                                    int(attach_count), N.ptr(loss), N.ptr(gs), N.ptr(gx), N.ptr(gq), N.ptr(ws), ws.numel(),
                                                    N.current_stream()))
        ctx.save_for_backward(gs, gx, gq)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        gs, gx, gq = ctx.saved_tensors
        return g * gs, g * gx, g * gq, None, None, None, None, None


class AttachSet:
    """The attach set of one mapping call (mapper.py:812-813: sigmoid(opacity at the start of the call) < 0.9), evaluated ONCE per call —
    the reference re-derives it in every iteration from the same snapshot."""

    def __init__(self, init_stat):
        self.init_stat = init_stat
        self.mask = (torch.sigmoid(init_stat["opacity"]) < 0.9).reshape(-1).to(torch.uint8).contiguous()
        self.count = int(self.mask.sum().item())


def fused_attach_loss(scaling, xyz, rotation, attach_set):
    """mapping.attach_loss (mapper.py:812-829) in two launches; attach_set = AttachSet(params.init_stat())."""
    st = attach_set.init_stat
    return _AttachLoss.apply(scaling, xyz, rotation, st["scaling"], st["xyz"], st["rotation_raw"], attach_set.mask, attach_set.count)


class DqoAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) — weight_decay = 0, amsgrad = False, the configuration of gaussian_pointcloud.py:331-378 —
    with the whole step as ONE kernel launch over every parameter tensor of every group (dqo_adam_multi; up to 16 tensors per launch),
    instead of torch's per-group launches and their host time (six groups: 0.28 ms per iteration on the drop-in path).  Same param
    groups (per-group `lr`, scheduler-compatible), same state keys (`step`, `exp_avg`, `exp_avg_sq`), same element arithmetic;
    parameters whose `.grad` is None are skipped like torch skips them.  betas / eps must be common to the groups.
    capturable=True: the step count lives on the device (ONE int32 shared by all parameters: `state[p]["step"]` is that tensor), so
    step() can be captured into a torch.cuda.graph and replayed — the kernel forms the bias corrections of the replayed step itself
    (dqo_adam_multi_dev); every parameter must then have a gradient at every step (one count for all of them)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        if not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("DqoAdam: bad betas / eps")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = bool(capturable)
        self._step_dev = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = N.lib()
        by_step = {}  # tensors that take the same step number go into one launch (normally: all of them)
        keep = []
        betas = eps = dev = None
        for group in self.param_groups:
            if betas is None:
                betas, eps = tuple(group["betas"]), float(group["eps"])
            elif tuple(group["betas"]) != betas or float(group["eps"]) != eps:
                raise RuntimeError("DqoAdam: betas and eps must be the same in every group")
            for p in group["params"]:
                if p.grad is None:
                    if self.capturable:
                        raise RuntimeError("DqoAdam(capturable=True): every parameter needs a gradient at every step (one device-side "
                                           "step count serves all of them)")
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise RuntimeError("DqoAdam: dense contiguous float32 GPU parameters only; there is no CPU path")
                if dev is None:
                    dev = p.device
                elif p.device != dev:
                    raise RuntimeError("DqoAdam: all parameters must live on one GPU (one launch on that device's current stream)")
                st = self.state[p]
                if len(st) == 0:
                    if self.capturable:
                        if self._step_dev is None:
                            self._step_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
                        st["step"] = self._step_dev
                    else:
                        st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                if self.capturable:
                    key = -1
                else:
                    st["step"] += 1
                    key = int(st["step"])
                by_step.setdefault(key, []).append(
                    N.DqoAdamTensor(p=p.data_ptr(), g=g.data_ptr(), m=st["exp_avg"].data_ptr(), v=st["exp_avg_sq"].data_ptr(), n=p.numel(),
                                    lr=float(group["lr"])))
        for step, ts in by_step.items():
            with torch.cuda.device(dev):
                for i in range(0, len(ts), 16):
                    chunk = ts[i:i + 16]
                    arr = (N.DqoAdamTensor * len(chunk))(*chunk)
                    if self.capturable:
                        N.check(lib.dqo_adam_multi_dev(ctypes.cast(arr, ctypes.c_void_p), len(chunk), N.ptr(self._step_dev),
                                                       1 if i + 16 >= len(ts) else 0, betas[0], betas[1], eps, N.current_stream()))
                    else:
                        N.check(lib.dqo_adam_multi(ctypes.cast(arr, ctypes.c_void_p), len(chunk), step, betas[0], betas[1], eps,
                                                   N.current_stream()))
        return loss


class CapturedIteration:
    """One mapping iteration of a caller on the reference's API — `iteration()` = render through the drop-in op, loss, `.backward()`,
    `optimizer.step()` — captured into ONE torch.cuda.graph and replayed (the op's 'graph' mode):

        it = CapturedIteration(iteration, optimizer)      # warm-up iterations (they train too), then the capture
        for _ in range(n): it.replay()
        it.check()                                         # one small D2H read: raises if a replayed frame outgrew the capacity

    `iteration` must not synchronise with the host (no `.item()`, no boolean-mask indexing: fused_ops.masked_mapping_loss /
    fused_attach_loss instead of the eager loss), the optimiser must be capturable (DqoAdam(capturable=True) or torch.optim.Adam(
    capturable=True)), and every tensor the iteration reads from outside (ground truth, masks, camera) is read in place at every
    replay.  The op's synchronisation mode is restored after the capture."""

    def __init__(self, iteration, optimizer, warmup=3):
        import diff_gaussian_rasterization_depth as dgr
        self._dgr, self.iteration, self.optimizer = dgr, iteration, optimizer
        self.warmup = max(1, int(warmup))
        self.graph = None
        self.replays = 0
        self._capture()

    def _capture(self):
        dgr = self._dgr
        mode = dgr._sync_mode
        try:
            dgr.set_sync_mode("lazy")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            # (the parameters' AccumulateGrad nodes were created on the caller's stream and the warm-up runs on a side stream, as
            # torch's capture recipe asks: the mismatch is intended, its warning is switched off for the warm-up)
            quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if quiet is not None:
                quiet(False)
            try:
                with torch.cuda.stream(side):  # the warm-up also measures the op's instance capacity (kept with 25 % headroom)
                    for _ in range(self.warmup):
                        self.optimizer.zero_grad(set_to_none=True)
                        self.iteration()
            finally:
                if quiet is not None:
                    quiet(True)
            torch.cuda.current_stream().wait_stream(side)
            dgr.verify_pending()
            dgr.set_sync_mode("graph")
            self.graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.iteration()
            self._header = dgr._last["header"]  # the captured forward's own geometry buffer: its header is valid after every replay
            # the op's pooled contexts that the captured forwards run on are this graph's now: they live and die with this object
            self._contexts = dgr.take_captured_contexts()
        finally:
            dgr.set_sync_mode(mode)

    def replay(self):
        self.graph.replay()
        self.replays += 1

    def header(self):
        h = self._header[:32].view(torch.int32).cpu().tolist()
        return dict(num_rendered=h[0], num_tiles=h[1], overflow=h[2], max_tile_count=h[3], num_visible=h[4], num_candidates=h[5])

    def check(self):
        """Raises if the last replayed frame outgrew the captured capacity (its outputs were background, its gradients zeros)."""
        h = self.header()
        if h["overflow"]:
            raise RuntimeError(f"CapturedIteration: the frame produced more Gaussian-tile instances than the captured capacity "
                               f"({h['num_candidates']} candidates); capture again (a new CapturedIteration measures the capacity anew)")
        return h

