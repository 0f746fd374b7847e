"""Host-side mirror of the dual-quadric fit of DQO-MAP (SLAM/multiprocess/quadrics.py) on top of libdqoraster.so.

`Ellipsoid_tensor` keeps the reference's class name and `forward(P)` meaning (quadrics.py:2144-2220: parameters axes_, R_,
center_; returns the projected axis-aligned bbox), `bboxes_iou` is quadrics.py:285-290, and `optimize_objects` is the
inner loop of `Object_Optimize_only` (quadrics.py:2251-2285) for ALL objects of a keyframe in one kernel launch instead
of ~60 eager launches x 20 iterations x objects.  GPU only.
"""
import torch

import _dqo_native as N


def _gpu_f32(x, dev):
    return torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()


def quadric_iou_fwd_bwd(axes, R, center, P34, obs_bbox):
    """Batched residual over B (object, view) pairs.  Returns dict(bbox[B,4], loss[B], valid[B], g_axes, g_R, g_center)."""
    if not (torch.is_tensor(axes) and axes.is_cuda):
        raise RuntimeError("quadric_iou_fwd_bwd needs GPU (ROCm) tensors; there is no CPU path.")
    dev = axes.device
    axes = _gpu_f32(axes, dev).reshape(-1, 3)
    B = axes.size(0)
    R, center = _gpu_f32(R, dev).reshape(B, 3, 3), _gpu_f32(center, dev).reshape(B, 3)
    P34, obs = _gpu_f32(P34, dev).reshape(B, 3, 4), _gpu_f32(obs_bbox, dev).reshape(B, 4)
    out = dict(bbox=torch.empty((B, 4), device=dev), loss=torch.empty((B,), device=dev),
               valid=torch.empty((B,), dtype=torch.int32, device=dev), g_axes=torch.empty((B, 3), device=dev),
               g_R=torch.empty((B, 3, 3), device=dev), g_center=torch.empty((B, 3), device=dev))
    with torch.cuda.device(dev):
        N.check(N.lib().dqo_quadric_iou_fwd_bwd(B, N.ptr(axes), N.ptr(R), N.ptr(center), N.ptr(P34), N.ptr(obs), N.ptr(out["bbox"]),
                                                N.ptr(out["loss"]), N.ptr(out["valid"]), N.ptr(out["g_axes"]), N.ptr(out["g_R"]),
                                                N.ptr(out["g_center"]), N.current_stream()))
    return out


class _QuadricResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axes, R, center, P34, obs):
        o = quadric_iou_fwd_bwd(axes, R, center, P34, obs)
        ctx.save_for_backward(o["g_axes"], o["g_R"], o["g_center"])
        ctx.shapes = (axes.shape, R.shape, center.shape)
        return o["loss"], o["bbox"]

    @staticmethod
    def backward(ctx, g_loss, g_bbox):
        ga, gR, gc = ctx.saved_tensors
        sa, sR, sc = ctx.shapes
        return ((ga * g_loss[:, None]).reshape(sa), (gR * g_loss[:, None, None]).reshape(sR), (gc * g_loss[:, None]).reshape(sc),
                None, None)


def bboxes_iou(bb1, bb2):
    """quadrics.py:285-290 on tensors/floats (python min/max semantics)."""
    inter_w = max(min(bb1[2], bb2[2]) - max(bb1[0], bb2[0]), 0)
    inter_h = max(min(bb1[3], bb2[3]) - max(bb1[1], bb2[1]), 0)
    area_inter = inter_w * inter_h
    return area_inter / ((bb1[2] - bb1[0]) * (bb1[3] - bb1[1]) + (bb2[2] - bb2[0]) * (bb2[3] - bb2[1]) - area_inter)


class Ellipsoid_tensor(torch.nn.Module):
    """quadrics.py:2144-2220.  forward(P) returns the projected bbox; `residual(P, obs)` the fused 1 - IoU loss."""

    def __init__(self, axes, R, center, bbox=None, device="cuda"):
        super().__init__()
        self.axes_ = torch.nn.Parameter(torch.as_tensor(axes, dtype=torch.float32, device=device).clone())
        self.R_ = torch.nn.Parameter(torch.as_tensor(R, dtype=torch.float32, device=device).clone())
        self.center_ = torch.nn.Parameter(torch.as_tensor(center, dtype=torch.float32, device=device).clone())
        self.bbox = bbox

    def residual(self, P, obs_bbox):
        dev = self.axes_.device
        loss, bbox = _QuadricResidual.apply(self.axes_[None], self.R_[None], self.center_[None], _gpu_f32(P, dev)[None],
                                            _gpu_f32(obs_bbox, dev)[None])
        return loss[0], bbox[0]

    def forward(self, P):
        dev = self.axes_.device
        o = quadric_iou_fwd_bwd(self.axes_.detach()[None], self.R_.detach()[None], self.center_.detach()[None],
                                _gpu_f32(P, dev)[None], torch.zeros((1, 4), device=dev))
        return o["bbox"][0]


def optimize_objects(axes, R, center, P34_views, obs_views, view_offset, view_schedule):
    """Object_Optimize_only inner loops (quadrics.py:2251-2285) for n_obj objects in one launch.

    axes[n,3], R[n,3,3], center[n,3]: initial ellipsoids (updated copies are returned);
    P34_views[V,3,4], obs_views[V,4]: all stored (K @ Rt, detection bbox) pairs, object o owning rows
    view_offset[o]:view_offset[o+1]; view_schedule[n, n_iters] int32: view used at each iteration (negative = from the end;
    the reference draws random.randint for iter <= 5 and uses -1 afterwards).  Returns (axes, R, center, loss_hist[n, n_iters]).
    """
    if not (torch.is_tensor(axes) and axes.is_cuda):
        raise RuntimeError("optimize_objects needs GPU (ROCm) tensors; there is no CPU path.")
    dev = axes.device
    axes = _gpu_f32(axes, dev).reshape(-1, 3).clone()
    n = axes.size(0)
    R, center = _gpu_f32(R, dev).reshape(n, 3, 3).clone(), _gpu_f32(center, dev).reshape(n, 3).clone()
    P34_views, obs_views = _gpu_f32(P34_views, dev).reshape(-1, 3, 4), _gpu_f32(obs_views, dev).reshape(-1, 4)
    view_offset = torch.as_tensor(view_offset, dtype=torch.int32, device=dev).contiguous()
    view_schedule = torch.as_tensor(view_schedule, dtype=torch.int32, device=dev).reshape(n, -1).contiguous()
    n_iters = view_schedule.size(1)
    hist = torch.empty((n, n_iters), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        N.check(N.lib().dqo_quadric_adam(n, n_iters, N.ptr(view_offset), N.ptr(P34_views), N.ptr(obs_views), N.ptr(view_schedule),
                                         N.ptr(axes), N.ptr(R), N.ptr(center), N.ptr(hist), N.current_stream()))
    return axes, R, center, hist
