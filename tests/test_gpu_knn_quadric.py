"""GPU parity tests for dqo_knn3 (bit-exact vs the oracle) and the quadric kernels (vs oracle and reference goldens)."""
import os

import numpy as np
import pytest

from dqo_harness import scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quadric_golden.npz")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    import _dqo_native
    _dqo_native.lib()
    return torch


@pytest.mark.parametrize("P", [1, 3, 4, 100, 4097, 40800, 100000])
def test_knn_exact(torch_cuda, oracle, P):
    torch = torch_cuda
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(P)
    pts = (rng.normal(size=(P, 3)) * np.array([3, 1.5, 2])).astype(np.float32)
    if P >= 100:
        pts[P // 2:P // 2 + 5] = pts[0]  # duplicates -> distance ties resolved by Morton order
    d, idx = distCUDA2(torch.tensor(pts, device="cuda"))
    od, oidx = oracle.knn3(pts)
    assert d.dtype == torch.float32 and idx.dtype == torch.int32 and tuple(idx.shape) == (P, 3)
    np.testing.assert_array_equal(idx.cpu().numpy(), oidx)   # index work: bit-exact
    np.testing.assert_array_equal(d.cpu().numpy(), od)       # same IEEE ops in the same order: bit-exact


def test_knn_surfel_room_and_scale_init(torch_cuda, oracle):
    """update_geometry (gaussian_pointcloud.py:519-570) on the workload's point distribution: 40 800 new points + existing."""
    torch = torch_cuda
    from dqo_harness import mapping
    cam, sc = scenes.make_config(2, P=60000)
    xyz = torch.tensor(sc["xyz"][:40800], device="cuda")
    extra = torch.tensor(sc["xyz"][40800:], device="cuda")
    rad = torch.full((40800,), 0.002, device="cuda")
    erad = torch.full((extra.shape[0],), 0.002, device="cuda")
    log_scales, invalid = mapping.update_geometry_scales(xyz, rad, extra, erad)
    # numpy restatement on top of the oracle's knn
    lo, hi = sc["xyz"][:40800].min(0), sc["xyz"][:40800].max(0)
    ex = sc["xyz"][40800:]
    ex = ex[((ex >= lo) & (ex <= hi)).all(1)]
    tot = np.concatenate([sc["xyz"][:40800], ex])
    _, oidx = oracle.knn3(tot)
    oidx = oidx[:40800]
    d = np.stack([np.linalg.norm(sc["xyz"][:40800] - tot[oidx[:, k]], axis=1) - 3 * 0.002 for k in range(3)], 1)
    exp = np.clip(np.sqrt((d ** 2).sum(1) / 3), 0.001, 0.05)[:, None] * np.array([1, 1, 0.1])
    np.testing.assert_allclose(log_scales.cpu().numpy(), np.log(exp), rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(invalid.cpu().numpy(), (d < 0).any(1))


def test_quadric_residual_vs_oracle_and_golden(torch_cuda, oracle):
    torch = torch_cuda
    import dqo_quadrics as dq
    g = np.load(GOLD)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device="cuda")
    o = dq.quadric_iou_fwd_bwd(t(g["single_axes"]), t(g["single_R"]), t(g["single_center"]), t(g["single_P"]), t(g["single_obs"]))
    r = oracle.quadric_iou_fwd_bwd(g["single_axes"], g["single_R"], g["single_center"], g["single_P"], g["single_obs"], np.float32)
    np.testing.assert_array_equal(o["valid"].cpu().numpy(), g["single_valid"])
    np.testing.assert_allclose(o["bbox"].cpu().numpy(), r["bbox"], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(o["loss"].cpu().numpy(), g["single_loss"], atol=5e-5)
    v = g["single_valid"].astype(bool)
    for k in ("g_axes", "g_R", "g_center"):
        a = o[k].cpu().numpy()[v].reshape(v.sum(), -1)
        for ref in (r[k], g["single_" + k]):  # oracle and the reference's own autograd
            b = ref[v].reshape(v.sum(), -1)
            rel = np.abs(a - b).max(1) / (np.abs(b).max(1) + 1e-12)
            assert rel.max() < 1e-3, (k, rel.max())
    # autograd wrapper with the reference's class name
    ell = dq.Ellipsoid_tensor(g["single_axes"][0], g["single_R"][0], g["single_center"][0])
    loss, bbox = ell.residual(g["single_P"][0], g["single_obs"][0])
    loss.backward()
    np.testing.assert_allclose(ell.axes_.grad.cpu().numpy(), g["single_g_axes"][0], rtol=2e-3)
    np.testing.assert_allclose(ell.center_.grad.cpu().numpy(), g["single_g_center"][0], rtol=2e-3)
    np.testing.assert_allclose(ell(g["single_P"][0]).cpu().numpy(), g["single_bbox"][0], rtol=2e-5)


def test_quadric_adam_batched(torch_cuda, oracle):
    torch = torch_cuda
    import dqo_quadrics as dq
    g = np.load(GOLD)
    n, nv = g["adam_axes0"].shape[0], g["adam_Pviews"].shape[1]
    off = np.arange(n + 1, dtype=np.int32) * nv
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device="cuda")
    axes, R, c, hist = dq.optimize_objects(t(g["adam_axes0"]), t(g["adam_R0"]), t(g["adam_center0"]),
                                           t(g["adam_Pviews"].reshape(-1, 3, 4)), t(g["adam_obsviews"].reshape(-1, 4)), off,
                                           g["adam_sched"])
    np.testing.assert_allclose(axes.cpu().numpy(), g["adam_axes"], atol=1e-5)   # reference trajectory (torch.optim.Adam)
    np.testing.assert_allclose(R.cpu().numpy(), g["adam_R"], atol=1e-5)
    np.testing.assert_allclose(c.cpu().numpy(), g["adam_center"], atol=1e-5)
    np.testing.assert_allclose(hist.cpu().numpy(), g["adam_loss"], atol=1e-4)
    # large batch: 8 objects x 5 views x 20 iterations of BASELINE config 3, repeated -> one launch
    reps = 64
    big = dq.optimize_objects(t(np.tile(g["adam_axes0"], (reps, 1))), t(np.tile(g["adam_R0"], (reps, 1, 1))),
                              t(np.tile(g["adam_center0"], (reps, 1))), t(np.tile(g["adam_Pviews"].reshape(-1, 3, 4), (reps, 1, 1))),
                              t(np.tile(g["adam_obsviews"].reshape(-1, 4), (reps, 1))), np.arange(n * reps + 1, dtype=np.int32) * nv,
                              np.tile(g["adam_sched"], (reps, 1)))
    np.testing.assert_array_equal(big[0].cpu().numpy()[:n], axes.cpu().numpy())
