"""GPU: the HIP tile-mask producers (dqo_tilemask, through the C ABI) against the reference goldens and the numpy oracle, plus the
full-size (1200x680) case and the fused evaluate_render_range helper."""
import os

import numpy as np
import pytest

from oracle import map_oracle as mo
from test_oracle_tilemask import CASES, G, case, topk_mask_agrees

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tm():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import _dqo_native
    _dqo_native.lib()
    import dqo_tilemask
    return torch, dqo_tilemask


def test_masks_vs_reference_goldens(tm):
    torch, M = tm
    for c in CASES:
        T, render, gt = case(c)
        mask = torch.tensor(T != 1, device="cuda")
        np.testing.assert_array_equal(M.pixelmask2tilemask(mask, 16).cpu().numpy(), G[f"{c}_pixelmask2tilemask"])
        for r in (0.5, 0.25, 0.9):
            np.testing.assert_array_equal(M.transmission2tilemask(mask, 16, r).cpu().numpy(), G[f"{c}_transmission2tilemask_{r}"])


def test_color_error_vs_reference_goldens(tm):
    torch, M = tm
    for c in CASES:
        T, render, gt = case(c)
        err, pooled = M.color_error_tiles(torch.tensor(render, device="cuda"), torch.tensor(gt, device="cuda"))
        np.testing.assert_array_equal(err.cpu().numpy(), G[f"{c}_color_error"])  # bit-exact: same fp32 operation order
        np.testing.assert_allclose(pooled.cpu().numpy(), G[f"{c}_meanpool"], rtol=2e-6, atol=1e-7)
        for r in (0.4, 0.1):
            m = M.colorerror2tilemask(err, 16, r, _pooled=pooled).cpu().numpy()
            k = int(pooled.numel() * r)
            assert topk_mask_agrees(m, G[f"{c}_colorerror2tilemask_{r}"], pooled.cpu().numpy(), k), (c, r)


def test_evaluate_render_range_full_size_vs_oracle(tm):
    torch, M = tm
    rng = np.random.default_rng(3)
    h, w = 680, 1200
    T = np.ones((h, w), np.float32)
    for _ in range(40):
        y0, x0, hh, ww = rng.integers(0, h), rng.integers(0, w), rng.integers(1, 300), rng.integers(1, 500)
        T[y0:y0 + hh, x0:x0 + ww] = 0.3
    render = rng.uniform(0, 1, (3, h, w)).astype(np.float32)
    render[:, T == 1] = 0
    gt = rng.uniform(0, 1, (3, h, w)).astype(np.float32)
    tT, tr, tg = (torch.tensor(a, device="cuda") for a in (T[None], render, gt))
    rm, tmask, ratio = M.evaluate_render_range(tT)
    np.testing.assert_array_equal(rm.cpu().numpy(), T != 1)
    np.testing.assert_array_equal(tmask.cpu().numpy(), mo.transmission2tilemask(T != 1, 16, 0.5))
    assert abs(float(ratio) - (T != 1).mean()) < 1e-6
    rm2, tmask2, ratio2 = M.evaluate_render_range(tT, global_opt=True)  # "after training" branch: no tile mask
    assert tmask2 is None and np.array_equal(rm2.cpu().numpy(), T != 1)
    rm3, tmask3, ratio3 = M.evaluate_render_range(tT, tr, tg, global_opt=True, sample_ratio=0.3)
    ref_mask, pooled, k = mo.colorerror2tilemask(mo.color_error_image(render, gt), 16, 0.3)
    assert topk_mask_agrees(tmask3.cpu().numpy(), ref_mask, pooled, k)
    up = np.repeat(np.repeat(tmask3.cpu().numpy().astype(bool), 16, 0), 16, 1)[:h, :w]
    np.testing.assert_array_equal(rm3.cpu().numpy(), up)
    assert abs(float(ratio3) - up.mean()) < 1e-6


def test_no_cpu_path(tm):
    torch, M = tm
    with pytest.raises(RuntimeError):
        M.transmission2tilemask(torch.zeros((32, 32), dtype=torch.bool), 16)
