"""GPU parity of cuda_utils._C.accumulate_gaussian_error / accumulate_gaussian_confidence (row f1) against the numpy oracle; exact in
max / min mode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _maps(seed, H, W, P):
    rng = np.random.default_rng(seed)
    ce = rng.uniform(-0.2, 1.0, (H, W)).astype(np.float32)
    de = rng.uniform(-0.1, 0.5, (H, W)).astype(np.float32)
    ne = rng.uniform(0, 1.0, (H, W)).astype(np.float32)
    ne[0, :5] = np.nan
    # spatially coherent index maps (as the rasteriser produces) with -1 holes and out-of-range ids
    ci = (rng.integers(0, P, (H // 8 + 1, W // 8 + 1)).repeat(8, 0).repeat(8, 1)[:H, :W]).astype(np.int32)
    di = (rng.integers(0, P, (H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W]).astype(np.int32)
    ci[rng.uniform(size=(H, W)) < 0.1] = -1
    di[rng.uniform(size=(H, W)) < 0.1] = -1
    di[3, 3] = P + 7
    return ce, de, ne, ci, di


@pytest.mark.parametrize("check_max", [True, False])
def test_accumulate_gaussian_error(check_max):
    import torch
    from cuda_utils._C import accumulate_gaussian_error
    from oracle import map_oracle as mo
    H, W, P = 211, 333, 5000
    ce, de, ne, ci, di = _maps(1, H, W, P)
    t = lambda a: torch.tensor(a, device="cuda")
    out = accumulate_gaussian_error(H, W, P, t(ce), t(de), t(ne), t(ci), t(di), 0.25, 0.1, 0.3, check_max)
    ref = mo.accumulate_gaussian_error(H, W, P, ce, de, ne, ci, di, 0.25, 0.1, 0.3, check_max)
    assert all(tuple(o.shape) == (P, 1) and o.dtype == torch.float32 for o in out)
    np.testing.assert_array_equal(out[3].cpu().numpy(), ref[3])  # counts: exact
    for a, b in zip(out[:3], ref[:3]):
        if check_max:
            np.testing.assert_array_equal(a.cpu().numpy(), b)     # max of the same floats: bit-exact
        else:
            np.testing.assert_allclose(a.cpu().numpy(), b, rtol=2e-5, atol=1e-6)
    with pytest.raises(RuntimeError, match="no CPU path"):
        accumulate_gaussian_error(H, W, P, torch.tensor(ce), t(de), t(ne), t(ci), t(di), 0.25, 0.1, 0.3, True)


def test_on_rasteriser_index_maps():
    """End to end like SLAM/multiprocess/mapper.py:1005-1047: errors of a render against a target, scattered by its hit maps."""
    import torch
    from cuda_utils._C import accumulate_gaussian_error
    from dqo_harness import scenes, mapping
    from oracle import map_oracle as mo
    cam, scene = scenes.make_config(1, P=8000)
    dev = torch.device("cuda")
    st = mapping.make_settings(cam, dev)
    with torch.no_grad():
        out = mapping.render(st, mapping.GaussianParams(scene, dev).activated())
    rng = np.random.default_rng(0)
    color_error = torch.abs(out["render"] - torch.tensor(rng.uniform(0, 1, (3, cam.H, cam.W)).astype(np.float32), device=dev)).mean(0)
    depth_error = torch.abs(out["depth"][0] - 2.0)
    normal_error = torch.zeros_like(depth_error)
    res = accumulate_gaussian_error(cam.H, cam.W, 8000, color_error, depth_error, normal_error, out["color_index_map"][0],
                                    out["depth_index_map"][0], 0.25, 0.1, 0.3, True)
    ref = mo.accumulate_gaussian_error(cam.H, cam.W, 8000, color_error.cpu().numpy(), depth_error.cpu().numpy(),
                                       normal_error.cpu().numpy(), out["color_index_map"][0].cpu().numpy(),
                                       out["depth_index_map"][0].cpu().numpy(), 0.25, 0.1, 0.3, True)
    for a, b in zip(res, ref):
        np.testing.assert_array_equal(a.cpu().numpy(), b)
    assert (res[0] > 0).sum() > 100


@pytest.mark.parametrize("H,W,P,seed", [(211, 333, 5000, 3), (17, 9, 4, 4), (64, 48, 100000, 5)])
def test_accumulate_gaussian_confidence(H, W, P, seed):
    """cuda_utils.cu:62-83: (max, min, mean) per Gaussian over the pixels that name it; negative, zero and NaN confidences, -1 holes
    and out-of-range ids in the index map, Gaussians that no pixel names."""
    import torch
    from cuda_utils._C import accumulate_gaussian_confidence
    from oracle import map_oracle as mo
    rng = np.random.default_rng(seed)
    conf = rng.normal(0.2, 1.0, (H, W)).astype(np.float32)
    conf[rng.uniform(size=(H, W)) < 0.05] = 0.0
    conf[0, :3] = np.nan
    conf[1, 1] = -0.0
    conf[H - 1, W - 4:] = [-1.5, -0.25, -3.0, -0.0]  # -0.0 among negatives (as an integer it is INT_MIN: must not win the minimum)
    idx = (rng.integers(0, P, (H // 4 + 1, W // 4 + 1)).repeat(4, 0).repeat(4, 1)[:H, :W]).astype(np.int32)
    idx[rng.uniform(size=(H, W)) < 0.1] = -1
    idx[H // 2, W // 2] = P + 3
    idx[idx == P - 1] = -1
    idx[H - 1, W - 4:] = P - 1  # ... and alone on their Gaussian
    t = lambda a: torch.tensor(a, device="cuda")
    out = accumulate_gaussian_confidence(H, W, P, t(idx), t(conf))
    ref = mo.accumulate_gaussian_confidence(H, W, P, idx, conf)
    assert all(tuple(o.shape) == (P, 1) and o.dtype == torch.float32 for o in out)
    for k in (0, 1):  # max / min of the same floats: exact (+0 and -0 compare equal)
        np.testing.assert_array_equal(out[k].cpu().numpy(), ref[k])
    np.testing.assert_allclose(out[2].cpu().numpy(), ref[2], rtol=2e-5, atol=1e-6, equal_nan=True)
    unseen = np.setdiff1d(np.arange(P), idx[(idx >= 0) & (idx < P)])
    if len(unseen):
        assert all((o.cpu().numpy()[unseen] == 0).all() for o in out)
    with pytest.raises(RuntimeError, match="no CPU path"):
        accumulate_gaussian_confidence(H, W, P, torch.tensor(idx), t(conf))
