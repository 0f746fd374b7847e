"""GPU parity on the rarely-taken paths: per-tile lists longer than the LDS sort capacity (in-place global sort), screen-filling
splats (multi-window binning, whole-image tile rects), image sizes that are not multiples of 16, lazy-mode capacity overflow."""
import numpy as np
import pytest

from dqo_harness import scenes
import util_rast as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    import _dqo_native
    _dqo_native.lib()
    return torch


def _dL(cam, seed=0):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32)


def _check(oracle, cam, sc, dL, **kw):
    hr = U.HipRun(cam, sc, **kw)
    o, r, _ = U.run_oracle(oracle, cam, sc, **kw)
    bad = U.flipped_pixels(hr.res, r)
    st = U.compare_forward(hr.res, r)
    keep = (~bad).astype(np.float32)
    dLm = (dL[0] * keep[None], dL[1] * keep[None])
    precomp = kw.get("colors_precomp") is not None
    o64, _, _ = U.run_oracle(oracle, cam, sc, dtype=np.float64, **kw)  # (only consulted for rows beyond the bar: compare_grads)
    gs = U.compare_grads(hr.backward(dLm, retain=False), U.oracle_backward(o, dLm, precomp), U.oracle_backward(o64, dLm, precomp))
    return o, st, gs


@pytest.mark.parametrize("P,longer_than,at_most", [(1300, 512, 1024), (2800, 1024, None), (9000, 4096, None)])
def test_long_tile_lists_global_sort(torch_cuda, oracle, P, longer_than, at_most):
    """Lists of 513..1024 entries (two waves: two register sorts + one exchange through LDS), beyond that (> 1024: block sort in
    LDS) and beyond the LDS capacity (> 4096 instances in one tile: the sort kernel's in-place global path); semi-transparent
    so the whole list is walked."""
    cam = scenes.Camera(96, 64, 80.0, 80.0, 47.5, 31.5)
    rng = np.random.default_rng(0)
    sc = scenes.frustum_cloud(5, P, cam, zmin=1.0, zmax=4.0)
    # squeeze all centres into a 20x20-pixel window around the image centre
    pc = np.stack([rng.uniform(-0.12, 0.12, P), rng.uniform(-0.12, 0.12, P), rng.uniform(1.0, 4.0, P)], 1)
    pc[:, :2] *= pc[:, 2:3]
    sc["xyz"] = pc.astype(np.float32)
    sc["opacity"] = rng.uniform(0.02, 0.08, (P, 1)).astype(np.float32)
    o, st, gs = _check(oracle, cam, sc, _dL(cam))
    rg = o.ctx("ranges")
    assert (rg[:, 1] - rg[:, 0]).max() > longer_than, "scene does not exercise the long-list path"
    if at_most is not None:
        assert (rg[:, 1] - rg[:, 0]).max() <= at_most, "scene is past the path this case is meant for"
    print("long lists:", (rg[:, 1] - rg[:, 0]).max(), st)


def test_screen_filling_splats_multi_window(torch_cuda, oracle):
    """64 huge, close splats (tile rect = whole image) among small ones: > 16384 candidate pairs in one binning chunk."""
    cam = scenes.Camera(640, 352, 300.0, 300.0, 319.5, 175.5)
    sc = scenes.frustum_cloud(7, 1500, cam, zmin=1.0, zmax=4.0)
    rng = np.random.default_rng(1)
    big = rng.choice(1024, 64, replace=False)  # all inside the first chunk of 1024 Gaussians
    sc["scales"][big] = np.array([1.5, 1.5, 0.15], np.float32)
    sc["xyz"][big, 2] = rng.uniform(0.6, 1.2, 64).astype(np.float32)
    sc["opacity"][big] = 0.05
    o, st, gs = _check(oracle, cam, sc, _dL(cam, 1))
    T = ((cam.W + 15) // 16) * ((cam.H + 15) // 16)
    tt = o.ctx("tiles_touched")[big]
    assert (tt >= T * 0.9).sum() >= 16 and tt.sum() > 16384  # whole-image rects, more candidates than one binning window
    print("giant splats:", st)


@pytest.mark.parametrize("W,H", [(17, 9), (100, 75), (333, 47)])
def test_odd_image_sizes(torch_cuda, oracle, W, H):
    cam = scenes.Camera(W, H, 0.8 * W, 0.8 * W, W / 2 - 0.3, H / 2 + 0.2, scenes.rot_yx(3.0, -2.0), np.array([0.01, 0.02, 0.0]))
    sc = scenes.frustum_cloud(W, 800, cam, zmin=0.8, zmax=3.0)
    sc["scales"] = (sc["scales"] * 3).astype(np.float32)
    _check(oracle, cam, sc, _dL(cam, 2))


def test_lazy_mode_overflow_is_detected_and_recovers(torch_cuda, oracle):
    import torch
    import diff_gaussian_rasterization_depth as dgr
    cam, sc = scenes.make_config(1, P=3000)
    try:
        dgr.set_context_pool(False)  # (the carried instance capacity of contexts allocated per call; pooled: test_gpu_context_pool.py)
        dgr.set_sync_mode("lazy")
        h1, _ = U.run_hip(cam, sc)  # first call of this shape measures N and sets the capacity hint
        key = (0, 3000, cam.W, cam.H)
        assert key in dgr._cap_hint
        dgr._verify_pending(block=True)
        dgr._cap_hint[key] = 64  # far too small on purpose
        U.run_hip(cam, sc)
        with pytest.raises(RuntimeError, match="only 64 fitted"):
            dgr._verify_pending(block=True)
        assert dgr._cap_hint[key] > 64  # capacity raised: the re-run is valid again
        h3, _ = U.run_hip(cam, sc)
        dgr._verify_pending(block=True)
        for k in ("color", "depth", "hit_depth", "T_map"):
            np.testing.assert_array_equal(h3[k], h1[k])
    finally:
        dgr.set_sync_mode("exact")
        dgr.set_context_pool(True)
    o, r, _ = U.run_oracle(oracle, cam, sc)
    U.compare_forward(h3, r)


def test_deferred_mode_never_waits_and_raises_an_overflow_later(torch_cuda, oracle):
    """set_sync_mode('deferred'): the backward does not wait for its forward's header.  A frame that outgrew the capacity renders
    background (lists emptied, nothing out of bounds) and the error comes from a later call — verify_pending() at the latest; a frame
    that fits gives the gradients of the exact mode bit for bit."""
    import torch
    import diff_gaussian_rasterization_depth as dgr
    cam, sc = scenes.make_config(1, P=3000)
    dL = _dL(cam, 4)
    h0, g0 = U.run_hip(cam, sc, dL=dL)  # exact mode
    try:
        dgr.set_context_pool(False)
        dgr.set_sync_mode("deferred")
        h1, g1 = U.run_hip(cam, sc, dL=dL)
        dgr.verify_pending()
        for k in g0:
            assert np.array_equal(g0[k], g1[k]), k
        key = (0, 3000, cam.W, cam.H)
        dgr._cap_hint[key] = 64  # far too small on purpose
        r = U.HipRun(cam, sc)  # the forward goes through without an exception: its header has not been looked at ...
        assert (r.res["hit_depth"] <= 0).all() and (r.res["T_map"] == 1).all()  # ... the invalid frame is background (lists emptied)
        with pytest.raises(RuntimeError, match="only 64 fitted"):
            r.backward(dL)         # ... and a LATER call raises (here the backward: reading the outputs back has let the header arrive;
            dgr.verify_pending()   # in a loop that never reads anything back, the next forward or verify_pending() does)
        assert dgr._cap_hint[key] > 64
        h3, g3 = U.run_hip(cam, sc, dL=dL)
        dgr.verify_pending()
        for k in g0:
            assert np.array_equal(g0[k], g3[k]), k
    finally:
        dgr.set_sync_mode("exact")
        dgr.set_context_pool(True)


@pytest.mark.parametrize("n", [64, 65, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8193])
def test_sort_paths_at_their_boundaries(torch_cuda, oracle, n):
    """A tile list of EXACTLY n entries — the lengths at which the per-tile sort changes path (one wave with 1..8 keys per lane,
    two waves, the queued block sort with 2048- / 4096-key segments, the two-segment global merge): n tiny surfels inside one tile,
    semi-transparent so that the whole list is walked, depths with many exact ties (ties fall back to the Gaussian id).  Forward and
    gradients against the oracle; the hit ids make the comparison sensitive to the order."""
    cam = scenes.Camera(48, 32, 60.0, 60.0, 23.5, 15.5)
    rng = np.random.default_rng(n)
    sc = scenes.frustum_cloud(11, n, cam, zmin=1.0, zmax=4.0)
    # centres inside the middle of tile (1, 0): pixels 20..27 x 4..11, radius <= 3 px => every Gaussian lists exactly that tile
    z = np.round(rng.uniform(1.0, 4.0, n) * 8) / 8  # 25 distinct depths: ties by the hundred
    u, v = rng.uniform(20.5, 27.5, n), rng.uniform(4.5, 11.5, n)
    pc = np.stack([(u - cam.cx) / cam.fx * z, (v - cam.cy) / cam.fy * z, z], 1)
    sc["xyz"] = ((pc - cam.t) @ cam.Rw2c).astype(np.float32)
    sc["scales"] = (rng.uniform(0.004, 0.012, (n, 3)) * z[:, None]).astype(np.float32)
    sc["opacity"] = rng.uniform(0.01, 0.04, (n, 1)).astype(np.float32)
    o, st, gs = _check(oracle, cam, sc, _dL(cam, 2))
    rg = o.ctx("ranges")
    lens = rg[:, 1] - rg[:, 0]
    assert lens.max() == n and (lens > 0).sum() == 1, "the scene must put all n Gaussians into one tile"
    import diff_gaussian_rasterization_depth as dgr
    assert dgr.last_header()["max_tile_count"] == n  # ... and none of them is culled as dead: the HIP list has n entries as well
