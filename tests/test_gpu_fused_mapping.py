"""GPU parity of the fused mapping-step helpers (row f2): each kernel against the numpy oracle, and the whole fused
iteration against the autograd + torch.optim.Adam path built on the drop-in operator."""
import numpy as np
import pytest

from dqo_harness import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import _dqo_native
    _dqo_native.lib()
    return torch


def _problem(torch, P=6000, cfg=1):
    from dqo_harness import mapping
    cam, scene = scenes.make_config(cfg, P=P)
    dev = torch.device("cuda")
    settings = mapping.make_settings(cam, dev)
    rng = np.random.default_rng(3)
    pert = dict(scene)
    pert["xyz"] = (scene["xyz"] + rng.normal(0, 0.004, scene["xyz"].shape)).astype(np.float32)
    pert["shs"] = scene["shs"].copy()
    pert["shs"][:, 0, :] += rng.normal(0, 0.15, (P, 3)).astype(np.float32)
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(pert, dev).activated())
    gt_color, gt_depth = tgt["render"].clone(), tgt["depth"].clone()
    mask = torch.tensor(rng.uniform(size=(cam.H, cam.W)) < 0.8, device=dev) & (tgt["depth_index_map"][0] >= 0)
    return cam, scene, settings, gt_color, gt_depth, mask, dev


def test_loss_kernel_vs_oracle(env):
    torch = env
    import ctypes
    import _dqo_native as N
    from oracle import map_oracle as mo
    rng = np.random.default_rng(0)
    H, W = 123, 211
    color, gt_color = rng.uniform(0, 1, (3, H, W)).astype(np.float32), rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    depth, gt_depth = rng.uniform(0.5, 3, (1, H, W)).astype(np.float32), rng.uniform(0.4, 3, (1, H, W)).astype(np.float32)
    gt_depth[0, :7] = 0
    color[1, 9, 9] = gt_color[1, 9, 9]
    idx = rng.integers(-1, 50, (1, H, W)).astype(np.int32)
    for mask in (rng.uniform(size=(H, W)) < 0.6, None):
        t = lambda a: torch.tensor(a, device="cuda")
        lib = N.lib()
        loss = torch.zeros(8, device="cuda")
        dC, dD = torch.empty((3, H, W), device="cuda"), torch.empty((1, H, W), device="cuda")
        ws = torch.empty(lib.dqo_map_loss_workspace_bytes(), dtype=torch.uint8, device="cuda")
        tc, td, ti, tgc, tgd = t(color), t(depth), t(idx), t(gt_color), t(gt_depth)
        tm = None if mask is None else t(mask.astype(np.uint8))
        N.check(lib.dqo_map_loss_fwd_bwd(W, H, N.ptr(tc), N.ptr(td), N.ptr(ti), N.ptr(tgc), N.ptr(tgd), N.ptr(tm), 0.8, 1.0, 0.1,
                                         N.ptr(loss), N.ptr(dC), N.ptr(dD), N.ptr(ws), ws.numel(), N.current_stream()))
        tot, cl, dl, oC, oD = mo.masked_loss(color, depth, idx, gt_color, gt_depth, mask)
        np.testing.assert_allclose(loss.cpu().numpy()[:3], [tot, cl, dl], rtol=2e-6)
        ls = loss.cpu().numpy()
        m_ = np.ones((H, W), bool) if mask is None else mask
        np.testing.assert_allclose(ls[4] / (3 * ls[5]), cl, rtol=2e-6)  # the unnormalised sums (what object shards all-reduce)
        assert int(ls[5]) == int(m_.sum()) and abs(ls[6] / max(ls[7], 1) - dl) <= 2e-6 * dl
        np.testing.assert_allclose(dC.cpu().numpy(), oC, rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(dD.cpu().numpy(), oD, rtol=2e-6, atol=1e-12)


def test_activate_and_adam_kernels_vs_oracle(env):
    torch = env
    import ctypes
    import _dqo_native as N
    from oracle import map_oracle as mo
    rng = np.random.default_rng(2)
    P, M = 3001, 16
    f = np.float32
    raw = dict(xyz=rng.normal(size=(P, 3)).astype(f), shs=rng.normal(size=(P, M, 3)).astype(f), op=rng.normal(size=(P, 1)).astype(f),
               sc=rng.normal(-4, 0.5, (P, 3)).astype(f), rot=rng.normal(size=(P, 4)).astype(f))
    t = {k: torch.tensor(v, device="cuda") for k, v in raw.items()}
    lib = N.lib()
    opac, scal, rots = torch.empty((P, 1), device="cuda"), torch.empty((P, 3), device="cuda"), torch.empty((P, 4), device="cuda")
    N.check(lib.dqo_map_activate(P, N.ptr(t["op"]), N.ptr(t["sc"]), N.ptr(t["rot"]), N.ptr(opac), N.ptr(scal), N.ptr(rots), N.current_stream()))
    a = mo.activate(raw["op"], raw["sc"], raw["rot"])
    for x, y in zip((opac, scal, rots), a):
        np.testing.assert_allclose(x.cpu().numpy(), y, rtol=3e-6)
    lrs = dict(xyz=0.001, f_dc=0.0005, f_rest=0.0005 / 20, opacity=0.0, scaling=0.004, rotation=0.001)
    m = {k: torch.zeros_like(v) for k, v in t.items()}
    v = {k: torch.zeros_like(v) for k, v in t.items()}
    om = {k: np.zeros(r.shape) for k, r in raw.items()}
    ov = {k: np.zeros(r.shape) for k, r in raw.items()}
    op_ = {k: r.astype(np.float64) for k, r in raw.items()}
    for step in range(1, 4):
        g = {k: rng.normal(size=r.shape).astype(f) for k, r in raw.items()}
        tg = {k: torch.tensor(x, device="cuda") for k, x in g.items()}
        st = N.DqoAdamStep(P=P, M=M, step=step, beta1=0.9, beta2=0.999, eps=1e-15, lr_xyz=lrs["xyz"], lr_f_dc=lrs["f_dc"],
                           lr_f_rest=lrs["f_rest"], lr_opacity=lrs["opacity"], lr_scaling=lrs["scaling"], lr_rotation=lrs["rotation"],
                           xyz=N.ptr(t["xyz"]), shs=N.ptr(t["shs"]), opacity_raw=N.ptr(t["op"]), scaling_raw=N.ptr(t["sc"]),
                           rotation_raw=N.ptr(t["rot"]), g_means3D=N.ptr(tg["xyz"]), g_sh=N.ptr(tg["shs"]), g_opacity=N.ptr(tg["op"]),
                           g_scales=N.ptr(tg["sc"]), g_rotations=N.ptr(tg["rot"]), m_xyz=N.ptr(m["xyz"]), m_shs=N.ptr(m["shs"]),
                           m_opacity=N.ptr(m["op"]), m_scaling=N.ptr(m["sc"]), m_rotation=N.ptr(m["rot"]), v_xyz=N.ptr(v["xyz"]),
                           v_shs=N.ptr(v["shs"]), v_opacity=N.ptr(v["op"]), v_scaling=N.ptr(v["sc"]), v_rotation=N.ptr(v["rot"]),
                           act_opacity=N.ptr(opac), act_scales=N.ptr(scal), act_rotations=N.ptr(rots))
        N.check(lib.dqo_map_adam_step(ctypes.byref(st), N.current_stream()))
        # the activations written by the Adam kernel are bit-identical to a separate activate launch on the updated values
        o2, s2, r2 = torch.empty_like(opac), torch.empty_like(scal), torch.empty_like(rots)
        N.check(lib.dqo_map_activate(P, N.ptr(t["op"]), N.ptr(t["sc"]), N.ptr(t["rot"]), N.ptr(o2), N.ptr(s2), N.ptr(r2), N.current_stream()))
        assert torch.equal(o2, opac) and torch.equal(s2, scal) and torch.equal(r2, rots)
        rg_op, rg_sc, rg_rot = mo.raw_grads(op_["op"], op_["sc"], op_["rot"], g["op"], g["sc"], g["rot"])
        rawg = dict(xyz=g["xyz"].astype(np.float64), shs=g["shs"].astype(np.float64), op=rg_op, sc=rg_sc, rot=rg_rot)
        for k in raw:
            if k == "shs":
                lr = np.full((1, M, 1), lrs["f_rest"])
                lr[0, 0, 0] = lrs["f_dc"]
            else:
                lr = lrs[dict(xyz="xyz", op="opacity", sc="scaling", rot="rotation")[k]]
            op_[k], om[k], ov[k] = mo.adam_step(op_[k], rawg[k], om[k], ov[k], lr, step)
    for k in raw:
        np.testing.assert_allclose(t[k].cpu().numpy(), op_[k], rtol=2e-5, atol=2e-6), k
        np.testing.assert_allclose(m[k].cpu().numpy(), om[k], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(v[k].cpu().numpy(), ov[k], rtol=2e-5, atol=1e-9)
    assert np.array_equal(t["op"].cpu().numpy(), raw["op"])  # opacity lr = 0: parameter untouched, moments still updated


def test_fused_iteration_matches_autograd_path(env):
    torch = env
    from dqo_harness import mapping
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    params = mapping.GaussianParams(scene, dev)
    opt = mapping.make_optimizer(params)
    init_stat = params.init_stat()  # mapper.py:533-545
    fm = FusedMapper(scene, settings, dev)
    assert 0 < fm.attach_count < fm.P  # the synthetic scenes give 10 % of the Gaussians opacity 0.1: the attach loss is live
    ref_losses, fused_losses, ref_attach, fused_attach = [], [], [], []
    for it in range(4):
        out = mapping.render(settings, params.activated())
        loss, parts = mapping.mapping_loss(out, gt_color, gt_depth, render_mask=mask)
        attach = mapping.attach_loss(params, init_stat)
        (loss + attach).backward()  # mapper.py:905
        opt.step()
        opt.zero_grad(set_to_none=True)
        ref_losses.append([parts[k].item() for k in ("total_loss", "color_loss", "depth_loss")])
        ref_attach.append(attach.item())
        fm.step(gt_color, gt_depth, mask)
        fused_losses.append(fm.loss.cpu().numpy()[:3].tolist())
        fused_attach.append(fm.attach_loss().item())
    np.testing.assert_allclose(fused_losses[:2], ref_losses[:2], rtol=2e-5)
    np.testing.assert_allclose(fused_losses, ref_losses, rtol=3e-4)  # (Adam's sign-like first steps amplify last-bit gradient differences)
    assert ref_attach[0] == 0.0 and fused_attach[0] == 0.0 and ref_attach[-1] > 0.0  # nothing has moved in the first iteration
    np.testing.assert_allclose(fused_attach, ref_attach, rtol=2e-2, atol=1e-9)  # (Adam's sign-like first steps: see below)
    ref = dict(xyz=params._xyz, shs=torch.cat([params._features_dc, params._features_rest], 1), opacity=params._opacity,
               scaling=params._scaling, rotation=params._rotation)
    got = fm._params()
    for k in ref:
        a, b = got[k].detach().cpu().numpy().reshape(-1), ref[k].detach().cpu().numpy().reshape(-1)
        # Adam's first steps move every touched parameter by ~lr regardless of gradient scale: compare against that step size
        lr = dict(xyz=0.001, shs=0.0005, opacity=1.0, scaling=0.004, rotation=0.001)[k]
        bad = np.abs(a - b) > 0.02 * 3 * lr + 1e-7
        assert bad.mean() < 2e-3, (k, bad.mean(), np.abs(a - b).max())


def test_unmasked_fused_iteration_carries_the_ssim_term(env):
    """No render mask: Mapping.loss_update adds 0.2 * (1 - ssim) (mapper.py:839-845).  Eager fused steps and the captured graph against
    the autograd path."""
    torch = env
    from dqo_harness import mapping
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    params = mapping.GaussianParams(scene, dev)
    opt = mapping.make_optimizer(params)
    init_stat = params.init_stat()
    fm = FusedMapper(scene, settings, dev)
    fg = FusedMapper(scene, settings, dev)
    fg.capture(gt_color, gt_depth, None)  # (its eager warm-up iteration is step 1)
    assert fg._g.tap is None  # the SSIM gradient is an image: loss kernels, not the tap
    ref_losses, fused_losses, graph_losses = [], [], []
    for it in range(3):
        out = mapping.render(settings, params.activated())
        loss, parts = mapping.mapping_loss(out, gt_color, gt_depth, render_mask=None)
        (loss + mapping.attach_loss(params, init_stat)).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        ref_losses.append([parts[k].item() for k in ("total_loss", "color_loss", "depth_loss", "ssim_loss")])
        fm.step(gt_color, gt_depth, None)
        fused_losses.append(fm.loss.cpu().numpy()[:4].tolist())
        if it > 0:
            fg.replay()
        graph_losses.append(fg.loss.cpu().numpy()[:4].tolist())
    assert ref_losses[0][3] > 0.01
    np.testing.assert_allclose(fused_losses[:2], ref_losses[:2], rtol=2e-5)
    np.testing.assert_allclose(fused_losses, ref_losses, rtol=3e-4)
    np.testing.assert_allclose(graph_losses, fused_losses, rtol=1e-5)
    ref = dict(xyz=params._xyz, scaling=params._scaling)
    got = fm._params()
    for k in ref:
        a, b = got[k].detach().cpu().numpy().reshape(-1), ref[k].detach().cpu().numpy().reshape(-1)
        lr = dict(xyz=0.001, scaling=0.004)[k]
        bad = np.abs(a - b) > 0.02 * 3 * lr + 1e-7
        assert bad.mean() < 2e-3, (k, bad.mean(), np.abs(a - b).max())


def test_graph_replay_matches_eager_steps(env):
    """capture() runs one iteration eagerly over the persistent buffers and records the next ones into a hipGraph (Adam step
    count on the device): capture + 3 replays must leave the same parameters / moments as 4 eager step() calls."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, settings, dev)
    for _ in range(4):
        a.step(gt_color, gt_depth, mask)
    b.capture(gt_color, gt_depth, mask)
    for _ in range(3):
        b.replay()
    torch.cuda.synchronize()
    assert not b.graph_overflowed()
    assert a.step_count == b.step_count == 4
    np.testing.assert_allclose(b.loss.cpu().numpy()[:3], a.loss.cpu().numpy()[:3], rtol=1e-5)
    for k, pa in a._params().items():
        pb = b._params()[k]
        lr = dict(xyz=0.001, shs=0.0005, opacity=1.0, scaling=0.004, rotation=0.001)[k]
        d = (pa - pb).abs()
        # identical arithmetic except the bias corrections (host pow vs device pow): differences stay far below one Adam step
        assert (d > 0.01 * lr + 1e-7).float().mean().item() < 1e-3, (k, d.max().item())
        for i in (0, 1):
            np.testing.assert_allclose(b.state[k][i].cpu().numpy(), a.state[k][i].cpu().numpy(), rtol=1e-3, atol=1e-9)
    # an eager step between replays is allowed: the device-side step count is resynchronised
    b.step(gt_color, gt_depth, mask)
    b.replay()
    torch.cuda.synchronize()
    assert b.step_count == 6 and int(b._g.step_dev.item()) == 7


def test_recapture_after_eager_steps_keeps_the_step_count(env):
    """capture -> replays -> eager step() calls -> capture(): the eager steps advance only the host-side step count (they never touch
    the graph's device counter), so a re-capture must not roll the count back to the device value — every later Adam step would use
    the bias corrections of an earlier step.  The sequence must end where the same number of plain eager steps ends."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, settings, dev)
    b.capture(gt_color, gt_depth, mask)          # iteration 1 (the capture's eager one)
    b.replay(), b.replay()                       # 2, 3
    b.step(gt_color, gt_depth, mask)             # 4, 5: eager, host count only
    b.step(gt_color, gt_depth, mask)
    assert b.step_count == 5
    b.capture(gt_color, gt_depth, mask)          # 6: must see step_count 5, not the device counter's 3
    assert b.step_count == 6 and int(b._g.step_dev.item()) == 7
    b.replay()                                   # 7
    torch.cuda.synchronize()
    assert b.step_count == 7 and int(b._g.step_dev.item()) == 8
    for _ in range(7):
        a.step(gt_color, gt_depth, mask)
    for k, pa in a._params().items():
        lr = dict(xyz=0.001, shs=0.0005, opacity=1.0, scaling=0.004, rotation=0.001)[k]
        d = (pa - b._params()[k]).abs()
        # (a rolled-back count changes the bias corrections of every later step by tens of percent of a step)
        assert (d > 0.01 * lr + 1e-7).float().mean().item() < 1e-3, (k, d.max().item())
        for i in (0, 1):
            np.testing.assert_allclose(b.state[k][i].cpu().numpy(), a.state[k][i].cpu().numpy(), rtol=1e-3, atol=1e-9)


@pytest.mark.parametrize("cfg,P,sparse,attach", [(1, 6000, True, True), (1, 6000, False, False), (3, 60000, True, True), (2, 30011, True, False)])
def test_fused_tail_is_bitwise_the_three_kernels(env, cfg, P, sparse, attach):
    """dqo_rast_backward_adam (record sum -> per-Gaussian chain -> Adam in ONE kernel, gradient rows in LDS) against
    dqo_rast_backward + dqo_map_adam_step (three kernels, gradient rows and summed records through HBM): the same statements on the
    same operands, so parameters, moments, activations, live bytes and the loss must be BIT-identical after several iterations — with
    and without the exact sparse mode and the attach loss, on maps whose blocks span several record chunks (cfg 3 layout) and whose
    size is not a multiple of the block (30011).  Only the reported attach loss is a differently grouped sum."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch, P=P, cfg=cfg)
    a = FusedMapper(scene, settings, dev, sparse_moments=sparse, attach=attach)
    b = FusedMapper(scene, settings, dev, sparse_moments=sparse, attach=attach)
    a.capture(gt_color, gt_depth, mask, fused_tail=True)
    b.capture(gt_color, gt_depth, mask, fused_tail=False)
    assert a._g.fused_tail and not b._g.fused_tail
    for it in range(5):
        a.replay(), b.replay()
        torch.cuda.synchronize()
        for k, pa in a._params().items():
            assert torch.equal(pa, b._params()[k]), (it, k, (pa - b._params()[k]).abs().max().item())
            assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), (it, k)
        assert torch.equal(a.opacity, b.opacity) and torch.equal(a.scales, b.scales) and torch.equal(a.rotations, b.rotations)
        if sparse:
            assert torch.equal(a.moment_live, b.moment_live)
        assert torch.equal(a.loss, b.loss)
        if attach:
            np.testing.assert_allclose(a.attach_loss().item(), b.attach_loss().item(), rtol=1e-5, atol=1e-12)
    assert not a.graph_overflowed() and a.step_count == b.step_count == 6
    assert int(a._g.step_dev.item()) == int(b._g.step_dev.item()) == 7
    if attach:
        assert a.attach_loss().item() > 0


def test_fused_tail_is_a_noop_on_an_overflowed_frame(env):
    """The fused tail keeps adam_kernel's rule: a frame whose header says overflow trains nothing (parameters, moments, live bytes
    and the device step count stay bit for bit)."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    fm = FusedMapper(scene, settings, dev)
    for _ in range(2):
        fm.step(gt_color, gt_depth, mask)
    before = {k: v.clone() for k, v in fm._params().items()}
    state_before = {k: (m.clone(), v.clone()) for k, (m, v) in fm.state.items()}
    live_before = fm.moment_live.clone()
    fm.capture(gt_color, gt_depth, mask, capacity_margin=0.05, fused_tail=True)
    assert fm._g.fused_tail and fm.step_count == 2
    step_before = int(fm._g.step_dev.item())
    for _ in range(3):
        fm.replay()
    torch.cuda.synchronize()
    assert fm.graph_overflowed() and int(fm._g.step_dev.item()) == step_before
    for k, v in fm._params().items():
        assert torch.equal(v, before[k]), k
        assert torch.equal(fm.state[k][0], state_before[k][0]) and torch.equal(fm.state[k][1], state_before[k][1]), k
    assert torch.equal(fm.moment_live, live_before)


def test_loss_tap_equals_the_loss_kernels(env):
    """DqoRastCtx.loss_tap: the masked loss summed inside the forward's blend kernel and its gradient formed inside the backward's
    must train exactly like the two loss kernels between them — same counts, same gradient scale, hence bit-identical parameters and
    moments — and report the same loss (fixed-point sums against double partial sums: 1e-6).  Also with a tile mask that leaves
    masked-in pixels in empty tiles, without a render mask, and on the eagerly issued iteration."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    for cfg, P, use_mask, use_tiles in ((1, 6000, True, False), (3, 20000, True, True), (3, 20000, False, False)):
        cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch, P=P, cfg=cfg)
        tile_mask = None
        if use_tiles:  # every other tile column switched off; the render mask keeps pixels there (they see the 0 / bg fills)
            tile_mask = torch.ones(((cam.H + 15) // 16, (cam.W + 15) // 16), dtype=torch.int32, device=dev)
            tile_mask[:, ::2] = 0
        m = mask if use_mask else None
        a = FusedMapper(scene, settings, dev)
        b = FusedMapper(scene, settings, dev)
        a.ssim_weight = b.ssim_weight = 0.0  # (without a mask the SSIM term would switch the tap off: this is the L1 pair alone)
        a.capture(gt_color, gt_depth, m, tile_mask=tile_mask, loss_tap=True)
        b.capture(gt_color, gt_depth, m, tile_mask=tile_mask, loss_tap=False)
        assert a._g.tap is not None and b._g.tap is None
        for it in range(4):
            if it == 2:
                a.step_static(), b.step_static()
            else:
                a.replay(), b.replay()
            torch.cuda.synchronize()
            la, lb = a.loss.cpu().numpy(), b.loss.cpu().numpy()
            assert la[5] == lb[5] and la[7] == lb[7] and la[5] > 0, (la, lb)  # pixel counts: exact
            np.testing.assert_allclose(la[[0, 1, 2, 4, 6]], lb[[0, 1, 2, 4, 6]], rtol=1e-6, atol=1e-9)
            gc, gd = 0.8 / (3.0 * max(la[5], 1.0)), 1.0 / max(la[7], 1.0)
            np.testing.assert_allclose(a._g.grad_scale.cpu().numpy(), [gc, gd], rtol=1e-6)
        assert not a.graph_overflowed() and not b.graph_overflowed()
        for k, pa in a._params().items():
            assert torch.equal(pa, b._params()[k]), (cfg, k)
            assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), (cfg, k)
        for x, y in zip(a._g.out, b._g.out):
            assert torch.equal(x, y)


def test_loss_tap_on_an_empty_map(env):
    """P = 0 through the C ABI with a loss tap: the forward writes the background, the backward has no blend kernel to run and still
    reports the loss of that frame (tap_report_kernel): mean |bg - gt| over the mask, depth term 0 (no depth hit anywhere)."""
    torch = env
    import ctypes
    import _dqo_native as N
    import diff_gaussian_rasterization_depth as dgr
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    lib = N.lib()
    H, W = cam.H, cam.W
    f, i32, u8 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev), dict(dtype=torch.uint8, device=dev)
    st = settings._replace(bg=torch.tensor([0.2, 0.4, 0.6], **f))
    out = [torch.empty((3, H, W), **f), torch.empty((1, H, W), **f), torch.empty((1, H, W), **i32), torch.empty((1, H, W), **i32),
           torch.empty((1, H, W), **f), torch.empty((1, H, W), **f), torch.empty((1, H, W), **f), torch.empty((0,), **i32), torch.empty((0,), **i32)]
    geom = torch.empty((lib.dqo_rast_geom_bytes(0, W, H),), **u8)
    img = torch.empty((lib.dqo_rast_image_bytes(W, H),), **u8)
    binning = torch.empty((lib.dqo_rast_binning_bytes(1),), **u8)
    e = torch.empty((0,), **f)
    params = dgr._params(st, 0, 16)
    inputs = dgr._inputs(st, e, e, e, e, e, e, e, None)
    outputs = N.DqoRastOutputs(out_color=out[0].data_ptr(), out_depth=out[1].data_ptr(), out_hit_color=out[2].data_ptr(),
                               out_hit_depth=out[3].data_ptr(), out_hit_color_weight=out[4].data_ptr(), out_hit_depth_weight=out[5].data_ptr(),
                               out_T=out[6].data_ptr(), n_touched=None, radii=None)
    loss, scale = torch.full((8,), -1.0, **f), torch.full((2,), -1.0, **f)
    m8 = mask.to(torch.uint8).contiguous()
    tap = N.DqoLossTap(gt_color=N.ptr(gt_color), gt_depth=N.ptr(gt_depth), render_mask=N.ptr(m8), out_color=out[0].data_ptr(),
                       out_depth=out[1].data_ptr(), color_weight=0.8, depth_weight=1.0, add_depth_thres=0.1, loss_out=N.ptr(loss),
                       grad_scale=N.ptr(scale))
    cctx = N.DqoRastCtx(geom=geom.data_ptr(), geom_bytes=geom.numel(), binning=binning.data_ptr(), binning_bytes=binning.numel(),
                        image=img.data_ptr(), image_bytes=img.numel(), inst_capacity=1, loss_tap=ctypes.addressof(tap))
    stream = N.current_stream()
    N.check(lib.dqo_rast_forward(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(outputs), ctypes.byref(cctx), stream))
    grads = N.DqoRastGrads()
    N.check(lib.dqo_rast_backward(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(cctx), None, None, out[3].data_ptr(),
                                  ctypes.byref(grads), None, 0, stream))
    torch.cuda.synchronize()
    n = float(mask.sum().item())
    # an empty map has no active tile: the op leaves the reference's initial fills (colour 0, ids 0), not the background
    want_c = float((gt_color.abs().sum(0)[mask]).sum().item()) / (3.0 * n)
    l = loss.cpu().numpy()
    assert l[5] == n
    np.testing.assert_allclose(l[1], want_c, rtol=1e-5)
    np.testing.assert_allclose(scale.cpu().numpy()[0], 0.8 / (3.0 * n), rtol=1e-6)
    assert (out[0] == 0).all()


def test_sparse_moments_are_bitwise_dense_adam(env):
    """DqoAdamStep.moment_live: Gaussians whose moments are still all zero and that get no gradient are skipped — parameters,
    moments and activations must come out bit for bit as from the dense update, also when the view (visible set) changes."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch, P=20000, cfg=3)  # cfg 3: 40 % of the map is in view
    a = FusedMapper(scene, settings, dev, sparse_moments=True)
    b = FusedMapper(scene, settings, dev, sparse_moments=False)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    left = torch.zeros((gy, gx), dtype=torch.int32, device=dev)
    left[:, : gx // 2] = 1  # tile mask: first only the left half is rendered (fewer Gaussians get gradients), then everything
    for tm in (left, left, None, None):
        a.step(gt_color, gt_depth, mask, tile_mask=tm)
        b.step(gt_color, gt_depth, mask, tile_mask=tm)
    torch.cuda.synchronize()
    live = a.moment_live.cpu().numpy()
    assert 0 < live.sum() < live.size  # some Gaussians were never in view: their rows were never touched
    for k, pa in a._params().items():
        assert torch.equal(pa, b._params()[k]), k
        assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), k
    assert torch.equal(a.opacity, b.opacity) and torch.equal(a.scales, b.scales) and torch.equal(a.rotations, b.rotations)
    dormant = torch.from_numpy(live == 0).to(dev)
    assert (a.state["xyz"][0][dormant] == 0).all() and (a.state["shs"][1][dormant] == 0).all()


def test_tile_buckets_match_packed_lists(env):
    """DqoRastCtx.tile_bucket_capacity: per-tile buckets instead of scanned list positions — same lists, same order, so the
    captured iteration must leave bit-identical parameters and moments; a bucket that is too small raises the overflow flag."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch, P=20000, cfg=3)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, settings, dev)
    a.capture(gt_color, gt_depth, mask, tile_buckets=True)
    b.capture(gt_color, gt_depth, mask, tile_buckets=False)
    assert a._g.bucket >= 256 and b._g.bucket == 0
    for _ in range(3):
        a.replay()
        b.replay()
    torch.cuda.synchronize()
    assert not a.graph_overflowed() and not b.graph_overflowed()
    for k, pa in a._params().items():
        assert torch.equal(pa, b._params()[k]), k
        assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), k
    for x, y in zip(a._g.out, b._g.out):
        assert torch.equal(x, y)
    # DqoRastCtx.keep_tile_order: the replays of `a` ran without tile_scan_kernel (launch order kept from the capture's eager
    # iteration, ranges and header produced by the sort kernels) — the header must still be the one the scan kernel writes
    assert a._g.cctx.keep_tile_order == 1 and b._g.cctx.keep_tile_order == 0
    assert a.header() == b.header() and a.header()["num_rendered"] > 0 and a.header()["num_tiles"] > 0
    d = FusedMapper(scene, settings, dev)
    d.capture(gt_color, gt_depth, mask, tile_buckets=True, keep_tile_order=False)
    for _ in range(3):
        d.replay()
    torch.cuda.synchronize()
    assert d._g.cctx.keep_tile_order == 0 and d.header() == a.header()
    for k, pa in a._params().items():
        assert torch.equal(pa, d._params()[k]), k
    # a bucket smaller than the longest list: flagged, never silent
    c = FusedMapper(scene, settings, dev)
    c.capture(gt_color, gt_depth, mask, tile_buckets=True)
    c._g.cctx.tile_bucket_capacity = 8  # (the buffers are sized for the larger bucket: still in bounds)
    for keep in (1, 0):
        c._g.cctx.keep_tile_order = keep
        c._static_iteration()
        torch.cuda.synchronize()
        assert c.graph_overflowed() and c.header()["max_tile_count"] > 8


@pytest.mark.parametrize("n", [700, 1500, 3000, 5000])
def test_long_lists_in_bucket_mode_match_packed_lists(env, n):
    """The captured iteration on a map whose Gaussians all fall into one or two tiles: lists of 700 (two-wave sort) to 5 000 entries
    (queued block sort, two segments) in BUCKET mode with the kept tile order and the loss tap, against the packed lists with the scan
    and loss kernels — bit-identical parameters, moments, outputs and header after three replays."""
    torch = env
    from dqo_harness import mapping
    from dqo_harness.fused_mapping import FusedMapper
    cam = scenes.Camera(96, 64, 80.0, 80.0, 47.5, 31.5)
    rng = np.random.default_rng(n)
    scene = scenes.frustum_cloud(5, n, cam, zmin=1.0, zmax=4.0)
    z = rng.uniform(1.0, 4.0, n)
    u, v = rng.uniform(34.0, 44.0, n), rng.uniform(18.0, 28.0, n)
    pc = np.stack([(u - cam.cx) / cam.fx * z, (v - cam.cy) / cam.fy * z, z], 1)
    scene["xyz"] = ((pc - cam.t) @ cam.Rw2c).astype(np.float32)
    scene["scales"] = (rng.uniform(0.004, 0.02, (n, 3)) * z[:, None]).astype(np.float32)
    scene["opacity"] = rng.uniform(0.02, 0.08, (n, 1)).astype(np.float32)
    dev = torch.device("cuda")
    settings = mapping.make_settings(cam, dev)
    pert = dict(scene)
    pert["shs"] = scene["shs"].copy()
    pert["shs"][:, 0, :] += rng.normal(0, 0.2, (n, 3)).astype(np.float32)
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(pert, dev).activated())
    gt_color, gt_depth = tgt["render"].clone(), tgt["depth"].clone()
    mask = torch.ones((cam.H, cam.W), dtype=torch.bool, device=dev)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, settings, dev)
    a.capture(gt_color, gt_depth, mask)
    b.capture(gt_color, gt_depth, mask, tile_buckets=False, loss_tap=False)
    assert a._g.bucket >= 1024 and a._g.cctx.keep_tile_order == 1 and a._g.tap is not None and b._g.bucket == 0
    for _ in range(3):
        a.replay()
        b.replay()
    torch.cuda.synchronize()
    ha, hb = a.header(), b.header()
    assert ha == hb and not ha["overflow"] and ha["max_tile_count"] > n // 6, (ha, hb)
    for k, pa in a._params().items():
        assert torch.equal(pa, b._params()[k]), k
        assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), k
    for x, y in zip(a._g.out, b._g.out):
        assert torch.equal(x, y)
    np.testing.assert_allclose(a.loss.cpu().numpy(), b.loss.cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_graph_capacity_overflow_is_flagged(env):
    """A captured graph has a fixed instance capacity; when the scene needs more, nothing is written out of bounds, the device
    header says so and graph_overflowed() reports it (the frame's outputs are the initial fills)."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    fm = FusedMapper(scene, settings, dev)
    for _ in range(2):  # two valid eager steps first: the moments are non-zero, so a step on a zero gradient WOULD move things
        fm.step(gt_color, gt_depth, mask)
    before = {k: v.clone() for k, v in fm._params().items()}
    state_before = {k: (m.clone(), v.clone()) for k, (m, v) in fm.state.items()}
    live_before = fm.moment_live.clone()
    fm.capture(gt_color, gt_depth, mask, capacity_margin=0.05)  # room for ~5 % of the candidate pairs only
    step_before = int(fm._g.step_dev.item())
    assert fm.step_count == 2  # the capture's own eager iteration overflowed: not counted
    for _ in range(3):
        fm.replay()
    torch.cuda.synchronize()
    assert fm.graph_overflowed()
    # a flagged frame is a no-op for the optimiser (DqoAdamStep.frame_header): parameters, moments, live bytes and the device step
    # count are exactly what they were, so the caller can re-capture and carry on from a clean state
    assert int(fm._g.step_dev.item()) == step_before
    for k, v in fm._params().items():
        assert torch.equal(v, before[k]), k
        assert torch.equal(fm.state[k][0], state_before[k][0]) and torch.equal(fm.state[k][1], state_before[k][1]), k
    assert torch.equal(fm.moment_live, live_before)
    # re-capture with room: continues as if the invalid replays had never happened
    fm.capture(gt_color, gt_depth, mask)
    assert fm.step_count == 3
    fm.replay()
    torch.cuda.synchronize()
    assert not fm.graph_overflowed() and int(fm._g.step_dev.item()) == 5
    twin = FusedMapper(scene, settings, dev)
    for _ in range(4):
        twin.step(gt_color, gt_depth, mask)
    for k, v in fm._params().items():
        lr = dict(xyz=0.001, shs=0.0005, opacity=1.0, scaling=0.004, rotation=0.001)[k]
        assert ((v - twin._params()[k]).abs() > 0.01 * lr + 1e-7).float().mean().item() < 1e-3, k


def test_step_count_survives_invalid_replays_followed_by_eager_steps(env):
    """Overflowed replays (no-ops for the optimiser, not counted by the device) -> an eager step() -> replay() -> capture(): the host
    count must drop the invalid replays BEFORE the eager step builds on it and before the resynchronising replay overwrites the device
    count — otherwise every later Adam step runs with the bias corrections of a later step than it is."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    fm = FusedMapper(scene, settings, dev)
    for _ in range(2):
        fm.step(gt_color, gt_depth, mask)                        # valid steps 1, 2
    fm.capture(gt_color, gt_depth, mask, capacity_margin=0.05)   # overflows: not counted
    for _ in range(3):
        fm.replay()                                              # three invalid replays (the host count runs ahead: 5)
    assert fm.step_count == 5
    fm.step(gt_color, gt_depth, mask)                            # eager (exact capacity): valid step 3
    assert fm.step_count == 3
    fm.replay()                                                  # invalid again (same small capacity); device count rewritten to 4 first
    torch.cuda.synchronize()
    assert fm.graph_overflowed() and int(fm._g.step_dev.item()) == 4
    fm.capture(gt_color, gt_depth, mask)                         # valid step 4
    assert fm.step_count == 4 and int(fm._g.step_dev.item()) == 5
    fm.replay()                                                  # valid step 5
    torch.cuda.synchronize()
    assert not fm.graph_overflowed() and fm.step_count == 5 and int(fm._g.step_dev.item()) == 6
    twin = FusedMapper(scene, settings, dev)
    for _ in range(5):
        twin.step(gt_color, gt_depth, mask)
    for k, v in fm._params().items():
        lr = dict(xyz=0.001, shs=0.0005, opacity=1.0, scaling=0.004, rotation=0.001)[k]
        assert ((v - twin._params()[k]).abs() > 0.01 * lr + 1e-7).float().mean().item() < 1e-3, k


def test_run_recaptures_after_an_overflow_and_delivers_valid_iterations(env):
    """FusedMapper.run(n): n VALID iterations whatever happens to the capacities — here the graph is captured with room for 5 % of the
    candidate pairs, so every replay of the first batch is an invalid frame (a no-op for the optimiser); run() notices from the
    device-side step count, captures again with room, and ends exactly where a mapper with enough room from the start ends."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, settings, dev)
    a.capture(gt_color, gt_depth, mask, capacity_margin=0.05)
    assert a.step_count == 0 and a.graph_overflowed()  # (the capture's own eager iteration overflowed: not counted)
    n_recap = a.run(7, check_every=3)
    assert n_recap == 1 and a.step_count == 7 and int(a._g.step_dev.item()) == 8 and not a.graph_overflowed()
    b.capture(gt_color, gt_depth, mask)
    assert b.run(6, check_every=4) == 0 and b.step_count == 7  # (capture's eager iteration + 6)
    torch.cuda.synchronize()
    for k, pa in a._params().items():
        assert torch.equal(pa, b._params()[k]), k
        assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), k
    assert torch.equal(a.loss, b.loss)


def test_read_header_after_the_fused_tail_and_the_prezeroed_guard(env):
    """(1) dqo_rast_read_header reports the device header stage 2 wrote — also in bucket mode, whose slot allocators are regional
    (counters[0] stays 0), and after dqo_rast_backward_adam, whose tail clears the counters the header was formed from.
    (2) DqoRastCtx.frame_prezeroed is a promise (the previous frame on the ctx ended in the fused tail): a forward-only render in
    between breaks it, and the next frame must SAY so (header.overflow, a no-op for the optimiser) instead of binning with dirty slot
    allocators; that frame's tail clears the scalars again, so the frame after it is valid."""
    torch = env
    import ctypes
    import _dqo_native as N
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    fm = FusedMapper(scene, settings, dev)
    fm.capture(gt_color, gt_depth, mask, fused_tail=True)
    g = fm._g
    assert g.cctx.frame_prezeroed == 1 and g.cctx.tile_bucket_capacity > 0
    fm.replay()
    torch.cuda.synchronize()
    lib = N.lib()
    hdr = N.DqoRastHeader()
    N.check(lib.dqo_rast_read_header(ctypes.byref(g.cctx), ctypes.byref(hdr), N.current_stream()))
    dev_hdr = g.geom[:32].view(torch.int32).cpu().numpy()
    assert hdr.stage == 2 and hdr.overflow == 0 and hdr.num_rendered > 0 and hdr.num_visible > 0 and hdr.num_tiles > 0
    assert [hdr.num_rendered, hdr.num_tiles, hdr.overflow, hdr.max_tile_count, hdr.num_visible, hdr.num_candidates] == list(dev_hdr[:6])
    assert hdr.num_candidates >= hdr.num_rendered
    steps = int(g.step_dev.item())
    # a forward-only render on the graph's ctx: its own frame is fine (it started from clean scalars) ...
    N.check(lib.dqo_rast_forward(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.outputs), ctypes.byref(g.cctx), N.current_stream()))
    torch.cuda.synchronize()
    assert not fm.graph_overflowed()
    # ... but it leaves them dirty: the next prezeroed frame is flagged and trains nothing
    fm.replay()
    torch.cuda.synchronize()
    assert fm.graph_overflowed() and int(g.step_dev.item()) == steps
    fm.replay()
    torch.cuda.synchronize()
    assert not fm.graph_overflowed() and int(g.step_dev.item()) == steps + 1


def test_async_header_of_a_frame_without_the_long_sort_launch_and_the_stage_marker(env):
    """(1) In a frame that skips the long-list sort launch (buckets <= 1024 entries, kept tile order, no list split) the header is formed
    by an extra block of the BLEND launch: dqo_rast_forward_async must hand over THIS frame's header (copy and event behind the blend
    launch), not the previous frame's.  (2) A K1-fused frame launches nothing in its first stage: between dqo_rast_forward_prepare and
    dqo_rast_forward_render dqo_rast_read_header reports zeros at stage 1 (include/dqo_raster.h), not the previous frame's stage 2."""
    torch = env
    import ctypes
    import _dqo_native as N
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    fm = FusedMapper(scene, settings, dev)
    fm.capture(gt_color, gt_depth, mask, fused_tail=True)
    g = fm._g
    assert g.cctx.frame_prezeroed == 1 and 0 < g.cctx.tile_bucket_capacity <= 1024 and g.cctx.keep_tile_order == 1 and g.cctx.list_split == 0
    fm.replay()
    torch.cuda.synchronize()
    lib = N.lib()
    prev = g.geom[:32].view(torch.int32).cpu().numpy().copy()
    # move the map so that this frame's header differs from the previous one's, then one frame through the async entry point
    with torch.no_grad():
        fm.xyz[: fm.P // 2] += 0.05
    host = torch.zeros(8, dtype=torch.int32).pin_memory()
    ev = torch.cuda.Event()
    ev.record()
    N.check(lib.dqo_rast_forward_async(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.outputs), ctypes.byref(g.cctx),
                                       ctypes.c_void_p(host.data_ptr()), ctypes.c_void_p(ev.cuda_event), N.current_stream()))
    ev.synchronize()
    got = host.numpy().copy()
    torch.cuda.synchronize()
    now = g.geom[:32].view(torch.int32).cpu().numpy()
    assert list(got[:7]) == list(now[:7]) and got[6] == 2, (got, now)
    assert list(now[:6]) != list(prev[:6])
    # the forward-only frame left dirty scalars: run the tail once (flagged frame, trains nothing, clears them), then a clean one
    fm.replay()
    fm.replay()
    torch.cuda.synchronize()
    assert not fm.graph_overflowed()
    hdr = N.DqoRastHeader()
    N.check(lib.dqo_rast_forward_prepare(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.outputs), ctypes.byref(g.cctx), N.current_stream()))
    N.check(lib.dqo_rast_read_header(ctypes.byref(g.cctx), ctypes.byref(hdr), N.current_stream()))
    assert hdr.stage == 1 and hdr.num_rendered == 0 and hdr.overflow == 0 and hdr.num_tiles == 0
    N.check(lib.dqo_rast_forward_render(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.outputs), ctypes.byref(g.cctx), N.current_stream()))
    N.check(lib.dqo_rast_read_header(ctypes.byref(g.cctx), ctypes.byref(hdr), N.current_stream()))
    assert hdr.stage == 2 and hdr.num_rendered > 0 and hdr.overflow == 0


def test_capture_takes_the_reference_cameras_noncontiguous_matrices(env):
    """scene/cameras.py:137-139 builds world_view_transform as torch.tensor(...).transpose(0, 1).cuda(): a non-contiguous view.  The
    op's forward makes it contiguous per call; the captured path must do the same once (its raw pointers would otherwise read the
    un-transposed matrix)."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch)
    view_t = settings.viewmatrix.t().contiguous().t()  # same values, transposed strides
    proj_t = settings.projmatrix.t().contiguous().t()
    assert not view_t.is_contiguous() and torch.equal(view_t, settings.viewmatrix)
    st2 = settings._replace(viewmatrix=view_t, projmatrix=proj_t)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, st2, dev)
    a.capture(gt_color, gt_depth, mask)
    b.capture(gt_color, gt_depth, mask)
    for _ in range(2):
        a.replay()
        b.replay()
    torch.cuda.synchronize()
    for x, y in zip(a._g.out, b._g.out):
        assert torch.equal(x, y)
    for k, v in a._params().items():
        assert torch.equal(v, b._params()[k]), k
    with pytest.raises(RuntimeError):
        FusedMapper(scene, settings, dev).capture(gt_color, gt_depth, mask, tile_mask=torch.ones((3, 3), dtype=torch.int64, device=dev))


def test_adam_attach_term_vs_oracle(env):
    """DqoAdamStep.attach_*: gradient of mapper.py:812-829's attach loss added on the raw parameters, reported loss value; checked
    against the numpy oracle (itself pinned by the reference's l2_loss fixture, tests/test_harness_loss.py)."""
    torch = env
    import ctypes
    import _dqo_native as N
    from oracle import map_oracle as mo
    rng = np.random.default_rng(5)
    P, M = 1500, 16
    f = np.float32
    raw0 = dict(xyz=rng.normal(size=(P, 3)).astype(f), shs=rng.normal(size=(P, M, 3)).astype(f), op=rng.normal(2.0, 2.5, (P, 1)).astype(f),
                sc=rng.normal(-4, 0.5, (P, 3)).astype(f), rot=rng.normal(size=(P, 4)).astype(f))
    raw = {k: (v + rng.normal(0, 0.01, v.shape)).astype(f) for k, v in raw0.items()}
    raw["op"] = raw0["op"].copy()
    t = {k: torch.tensor(v, device="cuda") for k, v in raw.items()}
    t0 = {k: torch.tensor(v, device="cuda") for k, v in raw0.items()}
    amask = (1.0 / (1.0 + np.exp(-raw0["op"].astype(np.float64).reshape(-1)))) < 0.9
    tmask = torch.tensor(amask.astype(np.uint8), device="cuda")
    g = {k: rng.normal(size=r.shape).astype(f) for k, r in raw.items()}
    tg = {k: torch.tensor(x, device="cuda") for k, x in g.items()}
    m = {k: torch.zeros_like(v) for k, v in t.items()}
    v = {k: torch.zeros_like(v) for k, v in t.items()}
    partial = torch.zeros(((P + 255) // 256,), device="cuda")
    lrs = dict(xyz=0.001, f_dc=0.0005, f_rest=0.0005 / 20, opacity=0.0, scaling=0.004, rotation=0.001)
    st = N.DqoAdamStep(P=P, M=M, step=1, beta1=0.9, beta2=0.999, eps=1e-15, lr_xyz=lrs["xyz"], lr_f_dc=lrs["f_dc"], lr_f_rest=lrs["f_rest"],
                       lr_opacity=lrs["opacity"], lr_scaling=lrs["scaling"], lr_rotation=lrs["rotation"], xyz=N.ptr(t["xyz"]),
                       shs=N.ptr(t["shs"]), opacity_raw=N.ptr(t["op"]), scaling_raw=N.ptr(t["sc"]), rotation_raw=N.ptr(t["rot"]),
                       g_means3D=N.ptr(tg["xyz"]), g_sh=N.ptr(tg["shs"]), g_opacity=N.ptr(tg["op"]), g_scales=N.ptr(tg["sc"]),
                       g_rotations=N.ptr(tg["rot"]), m_xyz=N.ptr(m["xyz"]), m_shs=N.ptr(m["shs"]), m_opacity=N.ptr(m["op"]),
                       m_scaling=N.ptr(m["sc"]), m_rotation=N.ptr(m["rot"]), v_xyz=N.ptr(v["xyz"]), v_shs=N.ptr(v["shs"]),
                       v_opacity=N.ptr(v["op"]), v_scaling=N.ptr(v["sc"]), v_rotation=N.ptr(v["rot"]), attach_mask=N.ptr(tmask),
                       init_xyz=N.ptr(t0["xyz"]), init_scaling_raw=N.ptr(t0["sc"]), init_rotation_raw=N.ptr(t0["rot"]),
                       attach_count=int(amask.sum()), attach_partial=N.ptr(partial))
    N.check(N.lib().dqo_map_adam_step(ctypes.byref(st), N.current_stream()))
    loss, a_sc, a_xyz, a_rot = mo.attach_loss(raw["sc"], raw["xyz"], raw["rot"], raw0["sc"], raw0["xyz"], raw0["rot"], raw0["op"])
    np.testing.assert_allclose(partial.sum().item(), loss, rtol=1e-5)
    _, rg_sc, rg_rot = mo.raw_grads(raw["op"], raw["sc"], raw["rot"], g["op"], g["sc"], g["rot"])
    # after ONE step from zero moments m = (1 - beta1) * gradient: read the total raw gradient back from the first moment
    for k, want in (("xyz", g["xyz"].astype(np.float64) + a_xyz), ("sc", rg_sc + a_sc), ("rot", rg_rot + a_rot)):
        np.testing.assert_allclose(m[k].cpu().numpy() / 0.1, want, rtol=2e-5, atol=2e-6), k
    np.testing.assert_allclose(m["shs"].cpu().numpy() / 0.1, g["shs"], rtol=2e-5, atol=2e-6)


def test_early_part_at_the_head_of_the_binning_kernel_gives_identical_bits(env):
    """A replayed iteration runs the early part of the per-Gaussian forward at the head of bin_count_kernel (csrc/dqo_k1_early.h; the
    previous frame's tail has cleared the tile histogram and left its stamp) instead of launching preprocess_kernel: the same statements
    — parameters, moments, loss, outputs, radii and the header's statistics after a capture + 6 replays must agree bit for bit with the
    two-launch sequence (DQO_K1_FUSE=0).  The switch is read once per process: each setting runs in a child process and reports a digest."""
    import os, subprocess, sys
    code = r'''
import hashlib, sys, os
import numpy as np, torch
root = os.environ["DQO_TEST_ROOT"]
sys.path[:0] = [root, root + "/dqo-map_amd", root + "/tests"]
import test_gpu_fused_mapping as T
from dqo_harness.fused_mapping import FusedMapper
cam, scene, settings, gt_color, gt_depth, mask, dev = T._problem(torch, P=9000)
h = hashlib.sha256()
for gated in (False, True):
    fm = FusedMapper(scene, settings, dev)
    if gated:
        go = (np.arange(9000) % 5).astype(np.int32)
        po = (np.add.outer(np.arange(cam.H) // 40, np.arange(cam.W) // 50) % 6 - 1).astype(np.int32)
        fm.set_object_gate(go, po)
    fm.capture(gt_color, gt_depth, mask, fused_tail=True)
    assert fm._g.cctx.frame_prezeroed == 1 and fm._g.cctx.tile_bucket_capacity > 0
    for _ in range(6):
        fm.replay()
    torch.cuda.synchronize()
    assert not fm.graph_overflowed()
    hdr = fm._g.geom[:32].view(torch.int32).cpu().numpy()
    assert hdr[4] > 0 and hdr[5] >= hdr[0] > 0
    h.update(hdr[:6].tobytes())
    for k, v in sorted(fm._params().items()):
        h.update(v.cpu().numpy().tobytes())
        for m in fm.state[k]:
            h.update(m.cpu().numpy().tobytes())
    h.update(fm.loss.cpu().numpy().tobytes())
    for o in fm._g.out:
        h.update(o.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for fuse in ("0", "1"):
        e = dict(os.environ, DQO_K1_FUSE=fuse, DQO_TEST_ROOT=root)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert digests[0] == digests[1]


def test_capture_placed_is_capture_on_other_buffers(env):
    """FusedMapper.capture_placed captures on several placements of the context buffers and keeps the fastest; the trials' iterations are
    undone: the mapper is exactly where capture() leaves it, and replays from there produce the same bits."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, scene, settings, gt_color, gt_depth, mask, dev = _problem(torch, P=8000)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, settings, dev)
    a.capture(gt_color, gt_depth, mask, unroll=2)
    b.capture_placed(gt_color, gt_depth, mask, trials=3, probe_replays=4, unroll=2)
    assert len(b.placement_trials_ms) == 3 and a.step_count == b.step_count == 1
    torch.cuda.synchronize()
    assert torch.equal(a.loss, b.loss)
    for _ in range(3):
        a.replay(), b.replay()
    torch.cuda.synchronize()
    assert a.step_count == b.step_count == 7 and not a.graph_overflowed() and not b.graph_overflowed()
    assert int(a._g.step_dev.item()) == int(b._g.step_dev.item()) == 8
    for k, pa in a._params().items():
        assert torch.equal(pa, b._params()[k]), k
        for i in (0, 1):
            assert torch.equal(a.state[k][i], b.state[k][i]), k
    assert torch.equal(a.loss, b.loss)
    for x, y in zip(a._g.out, b._g.out):
        assert torch.equal(x, y)
