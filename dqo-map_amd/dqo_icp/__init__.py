"""Point-to-plane ICP of DQO-MAP's tracker on MI355X.

`ICP` mirrors the class of /root/reference/SLAM/icp.py:16-129 (same constructor arguments, same `icp(...)` signature and return
value).  Per Gauss-Newton iteration the reference builds residuals, Jacobians and the 6x6 normal equations with ~40 eager torch
kernels; here that is one call of libdqoraster.so's dqo_icp_normal_equations (csrc/icp.hip).  The damped 6x6 solve and the
se(3) exponential (icp.py:248-337) are restated in double precision on the host, where the reference also runs them
(its invH moves the matrix to the CPU).  GPU only: there is no CPU path.
"""
import ctypes
import math

import numpy as np
import torch

import _dqo_native as N


def normal_equations(vertex0, vertex1, normal0, normal1, pose10, K, distance_threshold, normal_threshold):
    """(JtJ [6,6], JtR [6,1], valid_count 0-dim int32) of ICP.compute_residuals_jacobian + compute_jtj + compute_jtr."""
    N.require_gpu(vertex0, vertex1, normal0, normal1)
    if not vertex0.is_cuda:
        raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
    dev = vertex0.device
    H, W = vertex0.shape[:2]
    v0, v1, n0, n1 = (t.float().contiguous() for t in (vertex0, vertex1, normal0, normal1))
    pose = pose10.to(device=dev, dtype=torch.float32).contiguous()
    Kc = K.detach().cpu() if torch.is_tensor(K) else torch.as_tensor(K)
    fx, fy, cx, cy = float(Kc[0, 0]), float(Kc[1, 1]), float(Kc[0, 2]), float(Kc[1, 2])
    lib = N.lib()
    JtJ = torch.empty((6, 6), dtype=torch.float32, device=dev)
    JtR = torch.empty((6, 1), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.dqo_icp_workspace_bytes(),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        N.check(lib.dqo_icp_normal_equations(H, W, N.ptr(v0), N.ptr(v1), N.ptr(n0), N.ptr(n1), N.ptr(pose), fx, fy, cx, cy,
                                             float(distance_threshold), float(normal_threshold), N.ptr(JtJ), N.ptr(JtR), N.ptr(cnt),
                                             N.ptr(ws), ws.numel(), N.current_stream()))
    return JtJ, JtR, cnt[0]


def lev_mar_H(JtWJ, damping):
    """icp.py:248-256: JtJ + damping * trace(JtJ) * I."""
    eye = torch.eye(6, dtype=JtWJ.dtype, device=JtWJ.device)
    return JtWJ + (torch.sum(eye * JtWJ) * damping) * eye


def exp_se3(xi):
    """icp.py:272-312 (rotation first, then translation through the left Jacobian), float64 on the host."""
    xi = np.asarray(xi, np.float64).reshape(6)
    w, v = xi[:3], xi[3:]
    wh = np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])
    wh2 = wh @ wh
    th = float(np.linalg.norm(w))
    if th <= 1e-8:
        ew, j = np.eye(3), np.eye(3)
    else:
        ew = np.eye(3) + wh * math.sin(th) / th + wh2 * (1.0 - math.cos(th)) / th ** 2
        j = np.eye(3) + (1.0 - math.cos(th)) / th ** 2 * wh + (th - math.sin(th)) / th ** 3 * wh2
    T = np.eye(4)
    T[:3, :3] = ew
    T[:3, 3] = j @ v
    return T


def forward_update_pose(H, Rhs, pose):
    """icp.py:259-269, 331-337: xi = -H^-1 Rhs (pinv when singular), pose <- exp(xi) pose."""
    Hn = H.detach().double().cpu().numpy()
    inv = np.linalg.pinv(Hn) if np.linalg.det(Hn) == 0 else np.linalg.inv(Hn)
    xi = -inv @ Rhs.detach().double().cpu().numpy().reshape(6)
    return torch.as_tensor(exp_se3(xi), dtype=pose.dtype, device=pose.device) @ pose


class ICP:
    def __init__(self, max_iter=3, damping=1e-6, distance_threshold=0.2, normal_threshold=20, verbose=False):
        self.max_iterations = max_iter
        self.distance_threshold = distance_threshold
        self.normal_threshold = np.cos(np.deg2rad(normal_threshold))
        self.damping = damping
        self.verbose = verbose

    def icp(self, pose10, vertex_t0, vertex_t1, normal_t0, normal_t1, K):
        cnt = None
        for _ in range(self.max_iterations):
            JtWJ, JtR, cnt = normal_equations(vertex_t0, vertex_t1, normal_t0, normal_t1, pose10, K, self.distance_threshold,
                                              self.normal_threshold)
            pose10 = forward_update_pose(lev_mar_H(JtWJ, self.damping), JtR, pose10)
        H, W = vertex_t0.shape[:2]
        return pose10, cnt / H / W

    __call__ = icp
