"""GPU: DqoRastCtx.list_split — the forward blend with the long tile lists shared between eight waves (rast_forward_blend.hip).
The split kernel groups the transmittance products by chunk, so it is compared with the serial kernel within float rounding (and with the
oracle within north_star's bar, like the serial kernel), not bit for bit; integer outputs may differ only on a handful of pixels that
sit on a threshold."""
import numpy as np
import pytest

from dqo_harness import scenes
import util_rast as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import _dqo_native
    _dqo_native.lib()
    import diff_gaussian_rasterization_depth as dgr
    yield dgr
    dgr.set_list_split(0)


def _dL(cam, seed):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))


def _close(a, b, what, flips_allowed):
    """Float images equal within rounding except on at most `flips_allowed` pixels."""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) > 2e-5 + 2e-5 * np.abs(b)
    bad = int(d.reshape(-1, d.shape[-2], d.shape[-1]).any(0).sum()) if d.ndim >= 2 else int(d.sum())
    assert bad <= flips_allowed, (what, bad)
    return bad


@pytest.mark.parametrize("cfg,P,runs", [(3, 30000, 64), (3, 30000, 200), (5, 120000, 128), (5, 120000, 1024), (1, None, 1)])
def test_split_forward_is_the_serial_forward_within_rounding(env, cfg, P, runs):
    dgr = env
    cam, sc = scenes.make_config(cfg, P=P)
    dL = _dL(cam, 3)
    dgr.set_list_split(0)
    a = U.HipRun(cam, sc)
    dgr.set_list_split(runs)
    b = U.HipRun(cam, sc)
    dgr.set_list_split(0)  # (the backward does not split; the flag is read by the forward only)
    flips = 4
    for k in ("color", "depth", "hit_color_weight", "hit_depth_weight", "T_map"):
        _close(a.res[k], b.res[k], k, flips)
    for k in ("hit_color", "hit_depth"):
        assert int((a.res[k] != b.res[k]).sum()) <= flips, k
    assert np.array_equal(a.res["radii"], b.res["radii"])
    assert int((a.res["n_touched"] != b.res["n_touched"]).sum()) <= 8 * flips
    # the backward reads the forward's per-pixel state (final T, last contributor, hit position): gradients agree like the images
    same = (np.abs(a.res["color"] - b.res["color"]).max(0) < 1e-4) & (a.res["hit_depth"][0] == b.res["hit_depth"][0])
    dLm = (dL[0] * same, dL[1] * same)
    ga = a.backward(dLm, retain=False)
    gb = b.backward(dLm, retain=False)
    U.assert_grads_match(ga, gb, "split vs serial forward")


@pytest.mark.parametrize("runs", [64, 256])
def test_split_forward_against_the_oracle(env, oracle, runs):
    """The same parity case the serial kernel passes (forward 1e-4, gradients 1e-3, flipped pixels within budget), object gate on."""
    dgr = env
    cam, sc = scenes.make_config(3, P=30000)
    res, _ = U.run_hip(cam, sc)
    go = np.asarray(sc["obj_id"], np.int32)
    hit = res["hit_depth"][0]
    po = np.where(hit >= 0, go[np.clip(hit, 0, None)], -1).astype(np.int32)
    dgr.set_list_split(runs)
    try:
        U.parity_case(oracle, cam, sc, _dL(cam, 9), fp64=True, object_gate=(go, po))
        U.parity_case(oracle, cam, sc, _dL(cam, 10), fp64=True)
    finally:
        dgr.set_list_split(0)


def test_split_forward_edge_lists(env, oracle):
    """Lists shorter than the number of runs, empty tiles, a tile mask, and an image whose size is no multiple of the tile."""
    dgr = env
    cam, sc = scenes.make_config(1)
    few = {k: (v[:40] if hasattr(v, "shape") and v.shape[:1] == (len(sc["xyz"]),) else v) for k, v in sc.items()}
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    tm = (np.arange(gy * gx).reshape(gy, gx) % 3 != 0).astype(np.int32)
    for scene, mask in ((few, None), (sc, tm)):
        dgr.set_list_split(0)
        a = U.HipRun(cam, scene, tile_mask=mask, grad=False)
        dgr.set_list_split(1)
        b = U.HipRun(cam, scene, tile_mask=mask, grad=False)
        dgr.set_list_split(0)
        for k in ("color", "depth", "T_map"):
            _close(a.res[k], b.res[k], k, 2)
        assert int((a.res["hit_depth"] != b.res["hit_depth"]).sum()) <= 2


def test_fused_mapper_with_split_lists_trains_like_the_serial_one(env):
    """FusedMapper.capture(list_split=128): a few captured iterations end within rounding of the serial capture's parameters."""
    import torch
    from dqo_harness import mapping
    from dqo_harness.fused_mapping import FusedMapper
    cam, sc = scenes.make_config(3, P=20000)
    dev = torch.device("cuda")
    st = mapping.make_settings(cam, dev)
    with torch.no_grad():
        r = mapping.render(st, mapping.GaussianParams(sc, dev).activated())
    gt_c = (r["render"] * 0.9 + 0.05).contiguous()
    gt_d = (r["depth"] * 1.01).contiguous()
    mask = torch.ones((cam.H, cam.W), dtype=torch.uint8, device=dev)
    outs = []
    for runs in (0, 128, (128, 0)):  # (forward only: the single-wave backward walks every list and ignores the forward's queue)
        fm = FusedMapper(sc, st, dev)
        fm.capture(gt_c, gt_d, mask, list_split=runs)
        assert (int(fm._g.ls_fwd), int(fm._g.ls_bwd)) == (runs if isinstance(runs, tuple) else (runs, runs))
        for _ in range(5):
            fm.replay()
        torch.cuda.synchronize()
        outs.append((fm.xyz.detach().cpu().numpy().copy(), fm.shs.detach().cpu().numpy().copy(), float(fm.loss[0].item())))
    for o in outs[1:]:
        assert abs(o[2] - outs[0][2]) <= 1e-5 * abs(outs[0][2])
        # Adam turns a gradient that changes sign near zero into a full step: compare in the norm, not element by element
        assert np.abs(o[0] - outs[0][0]).mean() < 1e-6 and np.abs(o[1] - outs[0][1]).mean() < 1e-6


def test_auto_list_split_follows_the_number_of_rendered_tiles(env):
    import torch
    from dqo_harness import mapping
    from dqo_harness.fused_mapping import FusedMapper
    cam, _ = scenes.make_config(3, P=1000)
    st = mapping.make_settings(cam, torch.device("cuda"))
    pick = FusedMapper.pick_list_split
    tiles = lambda n: torch.ones(n, dtype=torch.int32)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    assert pick("auto", None, st, 5000) == pick("auto", tiles(gy * gx), st, 5000)
    assert pick("auto", tiles(200), st, 5000) == 256 and pick("auto", tiles(200), st, 800) == 0  # no list worth cutting
    assert pick("auto", tiles(600), st, 1500) == 512
    assert pick("auto", tiles(1200), st, 5000) == 1024 and pick("auto", tiles(4000), st, 6000) == 0  # a frame that fills the GPU
    assert pick(300, None, st) == 300 and pick(0, None, st) == 0
    pair = FusedMapper.pick_list_split_pair
    assert pair("auto", tiles(4000), st, 6000) == (2048, 0) and pair("auto", tiles(4000), st, 3000) == (0, 0)  # forward only on a full frame
    assert pair("auto", tiles(2000), st, 3000) == (1024, 0)  # half a frame
    assert pair("auto", tiles(200), st, 5000) == (256, 256) and pair(300, None, st) == (300, 300) and pair((512, 0), None, st) == (512, 0)
    with pytest.raises(ValueError):
        pair((512, 256), None, st)
    with pytest.raises(ValueError):
        pick(-1, None, st)


def test_split_backward_with_quadrants_that_get_no_gradient(env):
    """Half of the image carries no incoming gradient: those quadrants take the backward's early exit (in the eight-wave blocks: wave 0
    takes back the validity marks, the others leave), and a second backward over the same forward with a full gradient must not see
    anything of the first — the split backward is a function of its arguments only, like the single-wave one."""
    dgr = env
    cam, sc = scenes.make_config(5, P=120000)
    dL = _dL(cam, 12)
    half = np.zeros((cam.H, cam.W), np.float32)
    half[:, : cam.W // 2] = 1
    dLh = (dL[0] * half, dL[1] * half)
    runs_ = {}
    for runs in (0, 128):
        dgr.set_list_split(runs)
        runs_[runs] = U.HipRun(cam, sc)
    dgr.set_list_split(0)
    a, b = runs_[0], runs_[128]
    # (a pixel on a threshold may fall on the other side with the split on: no gradient comes in there, on either side)
    same = ((np.abs(a.res["color"] - b.res["color"]).max(0) < 1e-4) & (a.res["hit_depth"][0] == b.res["hit_depth"][0])).astype(np.float32)
    assert same.mean() > 0.999
    dL, dLh = (dL[0] * same, dL[1] * same), (dLh[0] * same, dLh[1] * same)
    res = {}
    for runs, r in runs_.items():
        g_half = r.backward(dLh)
        g_full = r.backward(dL)       # records of the first call must not leak into this one
        g_half2 = r.backward(dLh, retain=False)
        for k in g_half:
            assert np.array_equal(g_half[k], g_half2[k]), (runs, k)  # bitwise: a function of its arguments
        res[runs] = (g_half, g_full)
    for ga, gb in zip(res[0], res[128]):
        U.assert_grads_match(ga, gb, "split vs single-wave backward")


def test_fused_mapper_with_split_lists_and_a_partial_render_mask(env):
    """The loss tap with a render mask that covers a third of the frame, long lists shared between eight waves: trains like the
    single-wave kernels (quadrants outside the mask have no gradient: the early exit of the split backward)."""
    import torch
    from dqo_harness import mapping
    from dqo_harness.fused_mapping import FusedMapper
    cam, sc = scenes.make_config(5, P=100000)
    dev = torch.device("cuda")
    st = mapping.make_settings(cam, dev)
    with torch.no_grad():
        r = mapping.render(st, mapping.GaussianParams(sc, dev).activated())
    gt_c = (r["render"] * 0.9 + 0.05).contiguous()
    gt_d = (r["depth"] * 1.01).contiguous()
    mask = torch.zeros((cam.H, cam.W), dtype=torch.uint8, device=dev)
    mask[cam.H // 3: 2 * cam.H // 3, 100:900] = 1
    outs = []
    for runs in (0, 128):
        fm = FusedMapper(sc, st, dev)
        fm.capture(gt_c, gt_d, mask, list_split=runs)
        for _ in range(4):
            fm.replay()
        torch.cuda.synchronize()
        assert not fm.graph_overflowed()
        outs.append((fm.xyz.detach().cpu().numpy().copy(), fm.shs.detach().cpu().numpy().copy(), float(fm.loss[0].item())))
    assert abs(outs[1][2] - outs[0][2]) <= 1e-5 * abs(outs[0][2])
    assert np.abs(outs[1][0] - outs[0][0]).mean() < 1e-6 and np.abs(outs[1][1] - outs[0][1]).mean() < 1e-6


def test_the_split_kernels_walk_short_lists_exactly_like_the_unsplit_ones(env):
    """With a threshold that no list reaches the split launch has no long list: its short-list blocks — the serial forward wave, and in
    the backward the ROW walk of blend_backward_kernel<7, GATE, true> since round 6 (the union walk until then: last-bit differences
    in the record sums) — must give the outputs and the gradients of list_split = 0 bit for bit."""
    dgr = env
    cam, sc = scenes.make_config(3, P=30000)
    dL = _dL(cam, 9)
    dgr.set_list_split(0)
    a = U.HipRun(cam, sc)
    ga = a.backward(dL, retain=False)
    hdr = dgr.last_header()
    dgr.set_list_split(max(64, 2 * int(hdr["max_tile_count"])))
    b = U.HipRun(cam, sc)
    gb = b.backward(dL, retain=False)
    dgr.set_list_split(0)
    for k in a.res:
        assert np.array_equal(a.res[k], b.res[k]), k
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
