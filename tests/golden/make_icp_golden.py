"""Generates tests/golden/icp_golden.npz by importing the REFERENCE module /root/reference/SLAM/icp.py in the authoring container
(CPU, no GPU) and calling ICP.compute_residuals_jacobian / compute_jtj / compute_jtr (icp.py:51-123) and one full ICP.icp
(three Gauss-Newton iterations, icp.py:33-47) on seeded synthetic vertex / normal maps.

Only runs where /root/reference exists; the produced .npz (inputs + expected outputs = data) is committed, the reference source
never is.  icp.py does `from SLAM.utils import *`; SLAM/utils.py is loaded exactly as in make_tilemask_golden.py (empty stubs for
the absent third-party modules its other helpers import).  Inputs are stored as float16-exact values to keep the fixture small.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

from make_tilemask_golden import REF, import_reference_utils


def import_reference_icp():
    u = import_reference_utils()
    pkg = types.ModuleType("SLAM")
    pkg.__path__ = [os.path.join(REF, "SLAM")]
    sys.modules["SLAM"] = pkg
    sys.modules["SLAM.utils"] = u
    spec = importlib.util.spec_from_file_location("SLAM.icp", os.path.join(REF, "SLAM", "icp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def scene(rng, H, W, K):
    """Two depth images of a smooth surface seen from slightly different poses; vertex / normal maps in camera frames."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))

    def maps(shift):
        z = 2.0 + 0.3 * np.sin((jj + shift) / 17.0) + 0.2 * np.cos(ii / 23.0) + 0.002 * rng.normal(size=(H, W))
        z[rng.uniform(size=(H, W)) < 0.03] = 0.0  # holes
        z = z.astype(np.float16).astype(np.float32)
        v = np.stack([(jj - cx) / fx * z, (ii - cy) / fy * z, z], -1).astype(np.float16).astype(np.float32)
        dzdx, dzdy = np.gradient(z, axis=1), np.gradient(z, axis=0)
        n = np.stack([-dzdx * fx / np.maximum(z, 1e-3), -dzdy * fy / np.maximum(z, 1e-3), np.ones_like(z)], -1)
        n = -(n / np.linalg.norm(n, axis=-1, keepdims=True))
        return v, n.astype(np.float16).astype(np.float32)

    v0, n0 = maps(0.0)
    v1, n1 = maps(1.5)
    return v0, v1, n0, n1


def main():
    icp = import_reference_icp()
    rng = np.random.default_rng(20250118)
    out = {}
    for ci, (H, W) in enumerate([(60, 80), (48, 64), (17, 23)]):
        K = np.array([[W * 0.9, 0, (W - 1) / 2.0], [0, W * 0.9, (H - 1) / 2.0], [0, 0, 1]], np.float32)
        v0, v1, n0, n1 = scene(rng, H, W, K)
        ang = 0.01
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        pose[:3, 3] = [0.01, -0.005, 0.008]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        dist_thr, nthr = 0.2, float(np.cos(np.deg2rad(20)))
        mask0 = t(v0)[..., -1] > 0.0
        res, J, valid = icp.ICP.compute_residuals_jacobian(t(v0), t(v1), t(n0), t(n1), mask0, t(pose), t(K), dist_thr, nthr)
        JtJ, JtR = icp.ICP.compute_jtj(J), icp.ICP.compute_jtr(J, res)
        tracker = icp.ICP(max_iter=3, damping=1e-6, distance_threshold=dist_thr, normal_threshold=20)
        pose_out, ratio = tracker.icp(t(pose), t(v0), t(v1), t(n0), t(n1), t(K))
        for k, a in dict(v0=v0, v1=v1, n0=n0, n1=n1).items():
            out[f"c{ci}_{k}_f16"] = a.astype(np.float16)
        out[f"c{ci}_K"], out[f"c{ci}_pose"] = K, pose
        out[f"c{ci}_JtJ"], out[f"c{ci}_JtR"] = JtJ.numpy(), JtR.numpy()
        out[f"c{ci}_valid"] = valid.numpy()
        out[f"c{ci}_pose_out"], out[f"c{ci}_valid_ratio"] = pose_out.numpy(), np.float32(ratio)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icp_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes; valid fractions", [float(out[f"c{i}_valid"].mean()) for i in range(3)])


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
