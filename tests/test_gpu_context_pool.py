"""The drop-in op's context pool (diff_gaussian_rasterization_depth._pool): in 'lazy' / 'deferred' mode a forward runs on a pooled
context — per-tile list buckets, the previous frame's tile launch order, counters cleared by the previous backward — instead of
three fresh buffers as the reference allocates them (rasterize_points.cu:37-155).  Same lists, same order, same kernels behind them:
every output and every gradient must be the bits of the unpooled op, whatever the caller does with the graphs in between."""
import numpy as np
import pytest

from dqo_harness import scenes
import util_rast as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _dL(cam, seed=0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((3, cam.H, cam.W)).astype(np.float32), rng.standard_normal((1, cam.H, cam.W)).astype(np.float32)


def _moved(sc, i):
    """The map after i optimiser-like steps (every frame of a loop sees other parameters)."""
    rng = np.random.default_rng(100 + i)
    out = dict(sc)
    out["xyz"] = (sc["xyz"] + 0.002 * i * rng.standard_normal(sc["xyz"].shape)).astype(np.float32)
    out["opacity"] = np.clip(sc["opacity"] * (1.0 - 0.01 * i), 0.01, 0.99).astype(np.float32)
    return out


def _same(a, b):
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def _kernel_calls(torch, fn):
    import _dqo_native as N
    N.profile_enable(True)
    N.profile_collect(reset=True)
    fn()
    torch.cuda.synchronize()
    prof = N.profile_collect(reset=True)
    N.profile_enable(False)
    return {k: v[1] for k, v in prof.items()}


def test_a_loop_on_pooled_contexts_gives_the_bits_of_fresh_contexts_with_four_launches_less(torch_cuda):
    import diff_gaussian_rasterization_depth as dgr
    torch = torch_cuda
    cam, sc = scenes.make_config(1, P=20000)
    dL = _dL(cam, 1)
    ref = [U.run_hip(cam, _moved(sc, i), dL=dL) for i in range(6)]  # exact mode, contexts of the call's own
    try:
        dgr.set_sync_mode("deferred")
        got = []
        for i in range(6):
            got.append(U.run_hip(cam, _moved(sc, i), dL=dL))
            dgr.verify_pending()
        for (h0, g0), (h1, g1) in zip(ref, got):
            _same(h0, h1), _same(g0, g1)
        key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, 20000, cam.W, cam.H)
        sets = dgr._pool[key]
        assert len(sets) == 1 and sets[0].order_valid and sets[0].clean and not sets[0].leased  # one context served the loop
        # steady state: no zero fill, no preprocess launch, no scan, no placement pass
        calls = _kernel_calls(torch, lambda: U.run_hip(cam, _moved(sc, 6), dL=dL))
        for k in ("zero_words_kernel", "preprocess_kernel", "tile_scan_kernel", "bin_place_kernel"):
            assert calls.get(k, 0) == 0, (k, calls)
        assert calls["bin_count_kernel"] == 1 and calls["blend_forward_kernel"] == 1 and calls["gaussian_rows_kernel"] == 1
        dgr.verify_pending()
        dgr.set_context_pool(False)
        calls = _kernel_calls(torch, lambda: U.run_hip(cam, _moved(sc, 6), dL=dL))
        assert calls["preprocess_kernel"] == 1 and calls["tile_scan_kernel"] == 1 and calls["bin_place_kernel"] == 1
        dgr.verify_pending()
    finally:
        dgr.set_sync_mode("exact")
        dgr.set_context_pool(True)


def test_graphs_that_outlive_their_iteration_keep_their_context(torch_cuda):
    """Outputs held across the next forward (DQO-MAP's loop does), a second backward through a retained graph, a forward without grad
    in between, a forward whose backward never runs: a context is reused only when the graph that used it is gone, and a frame that
    finds counters nobody cleared zeroes them itself."""
    import diff_gaussian_rasterization_depth as dgr
    torch = torch_cuda
    cam, sc = scenes.make_config(1, P=12000)
    dLa, dLb = _dL(cam, 2), _dL(cam, 3)
    refs = []
    for i in range(4):
        r = U.HipRun(cam, _moved(sc, i))
        refs.append((r.res, r.backward(dLa), r.backward(dLb)))
    try:
        dgr.set_sync_mode("lazy")
        U.run_hip(cam, sc, dL=dLa)  # (the shape's statistics: the pool starts with the second call)
        dgr.verify_pending()
        runs = [U.HipRun(cam, _moved(sc, i)) for i in range(3)]  # three graphs alive at once: three contexts
        key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, 12000, cam.W, cam.H)
        assert len(dgr._pool[key]) == 3 and all(cs.leased for cs in dgr._pool[key])
        r3 = U.HipRun(cam, _moved(sc, 3))  # a fourth: every pooled context is leased -> a context of its own
        assert r3.out[0].grad_fn.pooled is None and len(dgr._pool[key]) == 3
        runs.append(r3)
        for i in (2, 0, 3, 1):  # backward in another order than forward, twice through each graph
            _same(refs[i][0], runs[i].res)
            _same(refs[i][1], runs[i].backward(dLa))
            _same(refs[i][2], runs[i].backward(dLb))
        del runs, r3
        assert not any(cs.leased for cs in dgr._pool[key])
        # a forward without grad leaves its context's counters in use; the next frame on it must not trust them
        with torch.no_grad():
            h, _ = U.run_hip(cam, _moved(sc, 1))
        _same(refs[1][0], h)
        r = U.HipRun(cam, _moved(sc, 2))  # ... and one whose backward never runs
        _same(refs[2][0], r.res)
        del r
        for i in range(4):
            h, g = U.run_hip(cam, _moved(sc, i), dL=dLa)
            _same(refs[i][0], h), _same(refs[i][1], g)
        dgr.verify_pending()
    finally:
        dgr.set_sync_mode("exact")


def test_a_pooled_context_that_is_outgrown_is_flagged_and_replaced(torch_cuda):
    import diff_gaussian_rasterization_depth as dgr
    torch = torch_cuda
    cam, sc = scenes.make_config(1, P=8000)
    dL = _dL(cam, 5)
    h0, g0 = U.run_hip(cam, sc, dL=dL)
    try:
        dgr.set_sync_mode("deferred")
        U.run_hip(cam, sc, dL=dL)
        dgr.verify_pending()
        skey = (torch.cuda.current_device(), 8000, cam.W, cam.H)
        assert dgr._shape_hint[skey][0] > 8000 and dgr._shape_hint[skey][1] > 16
        dgr._shape_hint[skey] = [64, 1]  # a context sized for a map of 64 (Gaussian, tile) pairs
        r = U.HipRun(cam, sc)
        assert r.out[0].grad_fn.pooled is not None
        assert (r.res["hit_depth"] <= 0).all() and (r.res["T_map"] == 1).all()  # the invalid frame is background, nothing out of bounds
        with pytest.raises(RuntimeError, match="pooled context"):
            dgr.verify_pending()
        del r
        h1, g1 = U.run_hip(cam, sc, dL=dL)  # sizes raised from the flagged frame's own header: valid again, on a new context
        dgr.verify_pending()
        _same(h0, h1), _same(g0, g1)
    finally:
        dgr.set_sync_mode("exact")


def test_the_pool_forgets_the_shapes_a_growing_map_has_left_behind(torch_cuda):
    import diff_gaussian_rasterization_depth as dgr
    torch = torch_cuda
    try:
        dgr.set_sync_mode("deferred")
        for P in range(3000, 3000 + 100 * 7, 100):  # seven map sizes, two frames each
            cam, sc = scenes.make_config(1, P=P)
            for _ in range(3):
                U.run_hip(cam, sc, dL=_dL(cam, 7))
                dgr.verify_pending()
        keys = list(dgr._pool)
        assert len(keys) == dgr._POOL_KEYS and [k[2] for k in keys] == [3300, 3400, 3500, 3600]
    finally:
        dgr.set_sync_mode("exact")
    assert len(dgr._pool) == 0


def test_a_forward_only_caller_checks_its_frame_and_renders_again(torch_cuda):
    """dqo_harness.mapping.perturbed_target (the targets of bench.py and of the full-size parity tests) is a forward-only caller: in the
    carrying modes nothing behind it would look at the frame's header, so it does — a frame that outgrew its pooled context (here: a
    context sized for far shorter lists) is rendered again with the sizes the flagged frame raised, and the target is the exact mode's."""
    import diff_gaussian_rasterization_depth as dgr
    from dqo_harness import mapping
    torch = torch_cuda
    cam, sc = scenes.make_config(1, P=9000)
    dev = torch.device("cuda")
    st = mapping.make_settings(cam, dev)
    full = dict(sc)
    full["obj_id"] = np.zeros(9000, np.int32)
    ref = mapping.perturbed_target(full, st, dev, 3)
    try:
        dgr.set_sync_mode("deferred")
        mapping.perturbed_target(full, st, dev, 3)  # (the shape's statistics)
        skey = (torch.cuda.current_device(), 9000, cam.W, cam.H)
        dgr._shape_hint[skey] = [64, 1]
        dgr._pool.clear()
        got = mapping.perturbed_target(full, st, dev, 3)
        assert dgr._shape_hint[skey][0] > 64
        for k in ref:
            assert torch.equal(ref[k], got[k]), k
        assert float(got["gt_color"].abs().sum()) > 0
    finally:
        dgr.set_sync_mode("exact")


@pytest.mark.parametrize("variant", ["gated", "colors_precomp", "sh_degree_1", "tile_mask"])
def test_pooled_contexts_on_the_ops_other_inputs(torch_cuda, variant):
    """The pooled path runs the per-Gaussian preprocess at the head of the binning kernel and its late part beside the sorts
    (DqoRastCtx.frame_prezeroed + buckets): every input form of the op goes through those too — the object gate, precomputed colours,
    a lower SH degree, a tile mask — and must give the bits of a context per call."""
    import diff_gaussian_rasterization_depth as dgr
    torch = torch_cuda
    cam, sc = scenes.make_config(3, P=16000)
    dL = _dL(cam, 11)
    kw = {}
    if variant == "gated":
        res, _ = U.run_hip(cam, sc)
        hit = res["hit_depth"][0]
        go = np.asarray(sc["obj_id"], np.int32)
        kw["object_gate"] = (go, np.where(hit >= 0, go[np.clip(hit, 0, None)], -1).astype(np.int32))
    elif variant == "colors_precomp":
        kw["colors_precomp"] = np.random.default_rng(2).uniform(0, 1, (16000, 3)).astype(np.float32)
    elif variant == "sh_degree_1":
        sc = dict(sc)
        sc["shs"] = np.ascontiguousarray(sc["shs"][:, :4, :])
        kw["sh_degree"] = 1
    else:
        tm = np.ones(((cam.H + 15) // 16, (cam.W + 15) // 16), np.int32)
        tm[::3, :] = 0
        kw["tile_mask"] = tm
    ref = [U.run_hip(cam, _moved(sc, i), dL=dL, **kw) for i in range(4)]
    try:
        dgr.set_sync_mode("deferred")
        for i in range(4):
            h, g = U.run_hip(cam, _moved(sc, i), dL=dL, **kw)
            dgr.verify_pending()
            _same(ref[i][0], h), _same(ref[i][1], g)
        key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, 16000, cam.W, cam.H)
        assert dgr._pool[key][0].order_valid and dgr._pool[key][0].clean
    finally:
        dgr.set_sync_mode("exact")
