"""GPU: row f4 — the fused ICP normal-equations kernel (dqo_icp, through the C ABI) against goldens from the reference's own
SLAM/icp.py, against the numpy oracle at tracking resolution, and the three-iteration Gauss-Newton loop end to end."""
import numpy as np
import pytest

from oracle import map_oracle as mo
from test_oracle_icp import CASES, DIST_THR, G, NORMAL_THR, case, close_normal_equations

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def icp():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import _dqo_native
    _dqo_native.lib()
    import dqo_icp
    return torch, dqo_icp


def test_normal_equations_vs_reference_goldens(icp):
    torch, M = icp
    for c in CASES:
        v0, v1, n0, n1, pose, K = case(c)
        t = lambda a: torch.tensor(a, device="cuda")
        JtJ, JtR, cnt = M.normal_equations(t(v0), t(v1), t(n0), t(n1), t(pose), torch.tensor(K), DIST_THR, NORMAL_THR)
        close_normal_equations(JtJ.cpu().numpy(), JtR.cpu().numpy(), c)
        assert abs(int(cnt) - int(G[f"{c}_valid"].sum())) <= max(2, 2e-3 * G[f"{c}_valid"].size)
        np.testing.assert_array_equal(JtJ.cpu().numpy(), JtJ.cpu().numpy().T)  # symmetric by construction


def test_three_iterations_vs_reference_goldens(icp):
    torch, M = icp
    for c in CASES:
        v0, v1, n0, n1, pose, K = case(c)
        t = lambda a: torch.tensor(a, device="cuda")
        tr = M.ICP(max_iter=3, damping=1e-6, distance_threshold=DIST_THR, normal_threshold=20)
        pose_out, ratio = tr.icp(t(pose), t(v0), t(v1), t(n0), t(n1), torch.tensor(K))
        np.testing.assert_allclose(pose_out.cpu().numpy(), G[f"{c}_pose_out"], rtol=0, atol=2e-4)
        assert abs(float(ratio) - float(G[f"{c}_valid_ratio"])) < 5e-3


def test_tracking_resolution_vs_oracle_and_reproducible(icp):
    torch, M = icp
    rng = np.random.default_rng(4)
    H, W = 340, 600
    K = np.array([[520.0, 0, 299.5], [0, 520.0, 169.5], [0, 0, 1]], np.float32)
    jj, ii = np.meshgrid(np.arange(W), np.arange(H))
    def maps(shift):
        z = (2.0 + 0.3 * np.sin((jj + shift) / 37.0) + 0.2 * np.cos(ii / 41.0)).astype(np.float32)
        z[rng.uniform(size=(H, W)) < 0.05] = 0
        v = np.stack([(jj - K[0, 2]) / K[0, 0] * z, (ii - K[1, 2]) / K[1, 1] * z, z], -1).astype(np.float32)
        n = np.stack([-np.gradient(z, axis=1) * K[0, 0] / np.maximum(z, 1e-3), -np.gradient(z, axis=0) * K[1, 1] / np.maximum(z, 1e-3),
                      np.ones_like(z)], -1)
        return v, (-(n / np.linalg.norm(n, axis=-1, keepdims=True))).astype(np.float32)
    (v0, n0), (v1, n1) = maps(0.0), maps(2.0)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, 3] = [0.01, 0.0, -0.01]
    t = lambda a: torch.tensor(a, device="cuda")
    JtJ, JtR, cnt = M.normal_equations(t(v0), t(v1), t(n0), t(n1), t(pose), torch.tensor(K), DIST_THR, NORMAL_THR)
    oJ, oR, valid = mo.icp_normal_equations(v0, v1, n0, n1, pose, K, DIST_THR, NORMAL_THR)
    assert abs(int(cnt) - int(valid.sum())) <= 1e-3 * valid.size
    np.testing.assert_allclose(JtJ.cpu().numpy(), oJ, rtol=0, atol=2e-4 * np.abs(oJ).max())
    np.testing.assert_allclose(JtR.cpu().numpy().reshape(-1), oR, rtol=0, atol=2e-4 * np.abs(oR).max() + 1e-6)
    JtJ2, JtR2, cnt2 = M.normal_equations(t(v0), t(v1), t(n0), t(n1), t(pose), torch.tensor(K), DIST_THR, NORMAL_THR)
    assert torch.equal(JtJ, JtJ2) and torch.equal(JtR, JtR2) and int(cnt) == int(cnt2)  # fixed-order reduction
