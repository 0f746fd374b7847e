"""GPU: the reference's optimise LOOP on the fused path (round 6) — a frame set with the per-iteration frame choice
(SLAM/multiprocess/mapper.py:549-576, 1159-1199), training one of the two clouds while rendering both (:533, 578, 1119, 1199-1204), the
per-iteration confidence counter (:908-910), per-call learning rates (:1120-1131) and history_merge (:607-650) — against the CPU oracle
iteration (oracle rasteriser forward / backward + oracle/map_oracle.py loss, activation Jacobians, attach loss, Adam), teacher-forced:
every iteration starts both sides from the GPU's state, so a last-bit difference cannot grow into a different trajectory."""
import os
import random

import numpy as np
import pytest

from dqo_harness import scenes
import util_rast as U

pytestmark = pytest.mark.gpu

B1, B2, EPS = 0.9, 0.999, 1e-15


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import _dqo_native
    _dqo_native.lib()
    from oracle import oracle_lib as ol
    return torch, ol


def _cameras(n):
    return [scenes.replica_camera(yaw=12.0 + 3.5 * k, pitch=4.0 - 1.5 * k, pos=(0.3 + 0.15 * k, 0.1 - 0.03 * k, -1.85 + 0.12 * k)) for k in range(n)]


def _window_problem(torch, P, n_frames, seed=0):
    from dqo_harness import mapping
    _, sc = scenes.make_config(3, P=P)
    dev = torch.device("cuda")
    cams, frames = _cameras(n_frames), []
    for k, cam in enumerate(cams):
        st = mapping.make_settings(cam, dev)
        tgt = mapping.perturbed_target(sc, st, dev, 100 + 7 * k + seed)
        rng = np.random.default_rng(50 + k)
        mask = torch.tensor(rng.uniform(size=(cam.H, cam.W)) < 0.8, device=dev) & (tgt["pix_obj"] >= 0)
        frames.append(dict(settings=st, gt_color=tgt["gt_color"].contiguous(), gt_depth=tgt["gt_depth"].contiguous(),
                           render_mask=mask.to(torch.uint8).contiguous()))
    return sc, cams, frames, dev


def _state(fm):
    c = lambda t: t.detach().cpu().numpy().copy()
    return dict(xyz=c(fm.xyz), shs=c(fm.shs), opacity=c(fm.opacity_raw), scaling=c(fm.scaling_raw), rotation=c(fm.rotation_raw),
                m={k: c(v[0]) for k, v in fm.state.items()}, v={k: c(v[1]) for k, v in fm.state.items()}, conf=c(fm.confidence),
                live=None if fm.moment_live is None else c(fm.moment_live))


GROUPS = (("xyz", "xyz"), ("shs", "shs"), ("opacity", "opacity"), ("scaling", "scaling"), ("rotation", "rotation"))


def _oracle_iteration(torch, ol, fm, cam, sc_rows, rows, s0, mask, gtc, gtd, init, lrs):
    """One CPU iteration on the rows `rows` (the rendered ones) from the GPU's state s0: returns (loss triple, raw-parameter gradients of
    the fp32 and the fp64 oracle as dicts over the Adam groups, [len(rows), ...]); the rasteriser sees the GPU's own activations."""
    from oracle import map_oracle as mo
    act = [a.detach().cpu().numpy() for a in fm.activate()]
    sca = dict(xyz=s0["xyz"][rows], shs=s0["shs"][rows], opacity=act[0][rows], scales=act[1][rows], rotations=act[2][rows])
    st = U.oracle_settings(ol, cam)
    out = {}
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        o = ol.OracleRasterizer(dt, omp=True)
        r = o.forward(st, sca["xyz"], sca["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, shs=sca["shs"],
                      scales=sca["scales"], rotations=sca["rotations"])
        out[name] = (o, r)
    return sca, out


def _finish_oracle(ol, out, s0, rows, mask, gtc, gtd, init, attach_rows):
    from oracle import map_oracle as mo
    grads, loss = {}, None
    for name in ("f32", "f64"):
        o, r = out[name]
        tot, col, dep, dC, dD = mo.masked_loss(r.color, r.depth, r.hit_depth, gtc, gtd, mask)
        if name == "f32":
            loss, dL = (tot, col, dep), (dC.astype(np.float32), dD.astype(np.float32))
        g = o.backward(*dL)
        g_op, g_sc, g_rot = mo.raw_grads(s0["opacity"][rows], s0["scaling"][rows], s0["rotation"][rows], g.opacity, g.scales, g.rotations)
        gx = np.asarray(g.means3D, np.float64)
        # the attach loss (mapper.py:812-829) over the trained cloud's attach set: rows `attach_rows` (bool over ALL rows)
        a = attach_rows[rows]
        n = int(attach_rows.sum())
        if n > 0:
            gx = gx + np.where(a[:, None], 2000.0 * (s0["xyz"][rows].astype(np.float64) - init["xyz"][rows]) / (3 * n), 0.0)
            g_sc = g_sc + np.where(a[:, None], 2000.0 * (s0["scaling"][rows].astype(np.float64) - init["scaling"][rows]) / (3 * n), 0.0)
            g_rot = g_rot + np.where(a[:, None], 2000.0 * (s0["rotation"][rows].astype(np.float64) - init["rotation"][rows]) / (4 * n), 0.0)
        grads[name] = dict(xyz=gx, shs=np.asarray(g.sh, np.float64), opacity=g_op, scaling=g_sc, rotation=g_rot)
    return loss, grads


def _check_iteration(s0, s1, t, trained, grads, lrs, sub_of_row, report):
    """GPU state s0 -> s1 by iteration t (1-based Adam step) against the oracle gradients `grads` (over the rendered rows; sub_of_row maps a
    map row to its index there, -1 = not rendered)."""
    P = s0["xyz"].shape[0]
    frozen = ~trained
    # ---- frozen rows: bit for bit untouched ----
    for k in ("xyz", "shs", "opacity", "scaling", "rotation", "conf"):
        assert np.array_equal(s0[k][frozen], s1[k][frozen]), f"frozen rows changed: {k}"
    for k in s0["m"]:
        assert np.array_equal(s0["m"][k][frozen], s1["m"][k][frozen]) and np.array_equal(s0["v"][k][frozen], s1["v"][k][frozen]), k
    if s0["live"] is not None:
        assert np.array_equal(s0["live"][frozen], s1["live"][frozen])
    # ---- the gradient the tail consumed = what its first moment moved by: g = m0 + (m1 - m0) / (1 - beta1) ----
    tr = np.nonzero(trained)[0]
    sub = sub_of_row[tr]
    assert (sub >= 0).all()
    hg, og, og64 = {}, {}, {}
    for gk, sk in GROUPS:
        m0, m1 = s0["m"][sk][tr].astype(np.float64), s1["m"][sk][tr].astype(np.float64)
        hg[gk] = m0 + (m1 - m0) / np.float64(np.float32(1.0 - B1))
        og[gk], og64[gk] = grads["f32"][gk][sub], grads["f64"][gk][sub]
    gs = U.compare_grads(hg, og, og64)
    # ---- Adam on the GPU's own moments: p1 = p0 - lr / (1 - b1^t) * m1 / (sqrt(v1) / sqrt(1 - b2^t) + eps) ----
    bc1, bc2 = 1.0 - B1 ** t, 1.0 - B2 ** t
    for gk, sk in GROUPS:
        p0, p1 = s0[sk][tr].astype(np.float64), s1[sk][tr].astype(np.float64)
        m1, v1 = s1["m"][sk][tr].astype(np.float64), s1["v"][sk][tr].astype(np.float64)
        if sk == "shs":
            lr = np.full(p0.shape, lrs["f_rest"])
            lr[:, 0, :] = lrs["f_dc"]
        else:
            lr = lrs[sk]
        want = p0 - (lr / bc1) * (m1 / (np.sqrt(v1) / np.sqrt(bc2) + EPS))
        touched = (m1 != 0).reshape(len(tr), -1).any(1) | (s1["m"][sk][tr] != s0["m"][sk][tr]).reshape(len(tr), -1).any(1)
        err = np.abs(p1 - want).reshape(len(tr), -1).max(1)
        tol = 4e-7 * np.abs(p0).reshape(len(tr), -1).max(1) + 1e-9
        assert (err[touched] <= tol[touched]).all(), (sk, t, float(err[touched].max()))
        untouched = ~touched
        assert np.array_equal(s0[sk][tr][untouched], s1[sk][tr][untouched]), sk
    # ---- confidence: += 1 where the f_dc gradient has a non-zero element (mapper.py:908-910) ----
    want_inc = (og["shs"][:, 0, :] != 0).any(-1)
    got_inc = (s1["conf"][tr] - s0["conf"][tr])
    assert set(np.unique(got_inc)) <= {0.0, 1.0}
    mism = int((got_inc.astype(bool) != want_inc).sum())
    assert mism <= 2, f"confidence increments differ on {mism} rows"
    report.append(dict(step=t, rows_beyond={k: v["beyond_bar"] for k, v in gs.items() if isinstance(v, dict) and "rows" in v},
                       confidence_mismatch=mism, confidence_gained=int(want_inc.sum())))
    return gs


def test_window_schedule_against_the_oracle_loop(env):
    """local_optimize on the fused path: 3 frames, the reference's schedule (random frame in the first half, the newest afterwards,
    mapper.py:570-576) for 8 iterations, 40 % of the rows trained (the unstable cloud), all rendered."""
    torch, ol = env
    from dqo_harness import mapping
    from dqo_harness.fused_mapping import FusedMapper
    P = 30000
    sc, cams, frames, dev = _window_problem(torch, P, 3)
    fm = FusedMapper(sc, frames[0]["settings"], dev)
    rng = np.random.default_rng(9)
    trained = rng.uniform(size=P) < 0.4
    fm.set_training_rows(trainable=torch.tensor(trained, device=dev))
    conf_init = rng.integers(0, 40, P).astype(np.float32)  # (confidence earned in earlier calls: history_merge weighs with it)
    fm.confidence.copy_(torch.tensor(conf_init, device=dev))
    fm.begin_mapping_call(reset_optimizer=True, history=True)
    init = dict(xyz=fm.init_xyz.cpu().numpy().astype(np.float64), scaling=fm.init_scaling.cpu().numpy().astype(np.float64),
                rotation=fm.init_rotation.cpu().numpy().astype(np.float64))
    attach_rows = fm.attach_mask.cpu().numpy().astype(bool)
    assert attach_rows.any() and not attach_rows[~trained].any()
    base_masks = [f["render_mask"].clone() for f in frames]
    before = _state(fm)
    fm.capture_window(frames, loss_tap=True, fused_tail=True)
    after = _state(fm)
    for k in ("xyz", "shs", "scaling", "rotation", "conf"):  # the captures' own eager iterations were undone
        assert np.array_equal(before[k], after[k]), k
    assert fm.step_count == 0 and len(fm._frames) == 3 and all(g.cctx.frame_prezeroed == 1 for g in fm._frames)
    sched = FusedMapper.window_schedule(8, 3, random.Random(5))
    assert sched[5:] == [2, 2, 2] and len(set(sched[:5])) > 1, sched
    rows = np.arange(P)
    sub_of_row = np.arange(P)
    report = []
    for it, k in enumerate(sched):
        cam, fr = cams[k], frames[k]
        s0 = _state(fm)
        sca, out = _oracle_iteration(torch, ol, fm, cam, None, rows, s0, None, None, None, init, fm.lrs)
        # flipped pixels of THIS state (found with the eager op on the same activations) leave the render mask on both sides
        hr = U.HipRun(cam, sca, grad=False)
        names = U.HipRun.names
        bad = U.flipped_pixels(hr.res, {n: getattr(out["f32"][1], n) for n in names}, {n: getattr(out["f64"][1], n) for n in names})
        assert bad.mean() <= 1e-3
        mask = base_masks[k].cpu().numpy().astype(bool) & ~bad
        fm.set_frame(k, render_mask=torch.tensor(mask, device=dev))
        fm.replay(frame=k)
        torch.cuda.synchronize()
        assert not fm.graph_overflowed()
        s1 = _state(fm)
        loss, grads = _finish_oracle(ol, out, s0, rows, mask, fr["gt_color"].cpu().numpy(), fr["gt_depth"].cpu().numpy(), init, attach_rows)
        np.testing.assert_allclose(fm.loss[:3].double().cpu().numpy(), loss, rtol=2e-5)
        _check_iteration(s0, s1, it + 1, trained, grads, fm.lrs, sub_of_row, report)
    assert fm.step_count == 8 and int(fm._step_dev.item()) == 9
    gained = fm.confidence.cpu().numpy() - conf_init
    assert np.abs(gained[~trained]).max() == 0 and gained[trained].max() >= 4
    print("window schedule", sched, report)
    # ---- history_merge closes the call (mapper.py:605): against the numpy restatement on the trained cloud ----
    from oracle import map_oracle as mo
    s1 = _state(fm)
    tr = np.nonzero(trained)[0]
    rot0 = torch.nn.functional.normalize(fm.init_rotation).cpu().numpy()
    hist = dict(confidence=fm.init_confidence.cpu().numpy()[tr, None], xyz=fm.init_xyz.cpu().numpy()[tr], features_dc=fm.init_shs.cpu().numpy()[tr, :1],
                features_rest=fm.init_shs.cpu().numpy()[tr, 1:], scaling=fm.init_scaling.cpu().numpy()[tr], rotation=rot0[tr])
    cur = dict(confidence=s1["conf"][tr, None], xyz=s1["xyz"][tr], features_dc=s1["shs"][tr, :1], features_rest=s1["shs"][tr, 1:],
               scaling=s1["scaling"][tr], rotation_raw=s1["rotation"][tr])
    want = mo.history_merge(hist, cur, 0.5)
    fm.history_merge(0.5)
    s2 = _state(fm)
    np.testing.assert_array_equal(s2["xyz"][tr], want["xyz"])
    np.testing.assert_array_equal(s2["shs"][tr, :1], want["features_dc"])
    np.testing.assert_array_equal(s2["shs"][tr, 1:], want["features_rest"])
    np.testing.assert_array_equal(s2["scaling"][tr], want["scaling"])
    np.testing.assert_allclose(s2["rotation"][tr], want["rotation"], atol=2e-6, rtol=0)
    for k in ("xyz", "shs", "scaling", "rotation"):
        assert np.array_equal(s1[k][~trained], s2[k][~trained]), k


def test_history_merge_kernel_against_the_reference_fixtures(env):
    torch, _ = env
    import _dqo_native as N
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "history_merge_golden.npz"))
    lib = N.lib()
    for ci in range(int(G["n_cases"])):
        p = f"c{ci}_"
        t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device="cuda")
        P = G[p + "conf"].shape[0]
        M = 1 + G[p + "hist_rest"].shape[1]
        shs0 = t(np.concatenate([G[p + "hist_dc"], G[p + "hist_rest"]], 1))
        shs = t(np.concatenate([G[p + "cur_dc"], G[p + "cur_rest"]], 1))
        xyz, sc, rot = t(G[p + "cur_xyz"]), t(G[p + "cur_scaling"]), t(G[p + "rot_raw"])
        args = [t(G[p + "conf0"]).reshape(-1), t(G[p + "conf"]).reshape(-1), t(G[p + "hist_xyz"]), shs0, t(G[p + "hist_scaling"]), t(G[p + "rot0"])]
        N.check(lib.dqo_map_history_merge(P, M, float(G[p + "max_weight"]), 0, None, *[N.ptr(a) for a in args], N.ptr(xyz), N.ptr(shs), N.ptr(sc),
                                          N.ptr(rot), N.current_stream()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(xyz.cpu().numpy(), G[p + "out_xyz"])
        np.testing.assert_array_equal(shs.cpu().numpy()[:, :1], G[p + "out_dc"])
        np.testing.assert_array_equal(shs.cpu().numpy()[:, 1:], G[p + "out_rest"])
        np.testing.assert_array_equal(sc.cpu().numpy(), G[p + "out_scaling"])
        np.testing.assert_allclose(rot.cpu().numpy(), G[p + "out_rotation"], atol=2e-6, rtol=0)
    # frozen rows stay, and `first_row` names the row whose weight serves the broadcast quirk
    p = "c0_"
    P = G[p + "conf"].shape[0]
    flags = np.zeros(P, np.uint8)
    flags[:3] = 1
    flags[10::2] = 1
    shs_in = np.concatenate([G[p + "cur_dc"], G[p + "cur_rest"]], 1)
    shs0 = t(np.concatenate([G[p + "hist_dc"], G[p + "hist_rest"]], 1))
    shs, xyz, sc, rot = t(shs_in), t(G[p + "cur_xyz"]), t(G[p + "cur_scaling"]), t(G[p + "rot_raw"])
    args = [t(G[p + "conf0"]).reshape(-1), t(G[p + "conf"]).reshape(-1), t(G[p + "hist_xyz"]), shs0, t(G[p + "hist_scaling"]), t(G[p + "rot0"])]
    fl = torch.tensor(flags, device="cuda")
    N.check(lib.dqo_map_history_merge(P, shs_in.shape[1], 0.5, 3, N.ptr(fl), *[N.ptr(a) for a in args], N.ptr(xyz), N.ptr(shs), N.ptr(sc), N.ptr(rot),
                                      N.current_stream()))
    from oracle import map_oracle as mo
    tr = np.nonzero(flags == 0)[0]
    assert tr[0] == 3
    hist = dict(confidence=G[p + "conf0"][tr], xyz=G[p + "hist_xyz"][tr], features_dc=G[p + "hist_dc"][tr], features_rest=G[p + "hist_rest"][tr],
                scaling=G[p + "hist_scaling"][tr], rotation=G[p + "rot0"][tr])
    cur = dict(confidence=G[p + "conf"][tr], xyz=G[p + "cur_xyz"][tr], features_dc=G[p + "cur_dc"][tr], features_rest=G[p + "cur_rest"][tr],
               scaling=G[p + "cur_scaling"][tr], rotation_raw=G[p + "rot_raw"][tr])
    want = mo.history_merge(hist, cur, 0.5)
    np.testing.assert_array_equal(xyz.cpu().numpy()[tr], want["xyz"])
    np.testing.assert_array_equal(shs.cpu().numpy()[tr, :1], want["features_dc"])
    np.testing.assert_array_equal(sc.cpu().numpy()[tr], want["scaling"])
    fz = flags != 0
    assert np.array_equal(xyz.cpu().numpy()[fz], G[p + "cur_xyz"][fz]) and np.array_equal(shs.cpu().numpy()[fz], shs_in[fz])
    assert np.array_equal(rot.cpu().numpy()[fz], G[p + "rot_raw"][fz]) and np.array_equal(sc.cpu().numpy()[fz], G[p + "cur_scaling"][fz])


def test_global_optimisation_trains_and_renders_the_stable_cloud_alone(env):
    """global_optimization on the fused path (mapper.py:1105-1228): the stable rows are the only ones rendered (stable_params) and
    trained, with xyz lr 0 and the other groups x 0.1 (:1120-1123) — against the oracle iteration on the stable rows ALONE."""
    torch, ol = env
    from dqo_harness.fused_mapping import FusedMapper
    P = 20000
    sc, cams, frames, dev = _window_problem(torch, P, 2, seed=3)
    fm = FusedMapper(sc, frames[0]["settings"], dev)
    rng = np.random.default_rng(4)
    stable = rng.uniform(size=P) < 0.7
    st_t = torch.tensor(stable, device=dev)
    fm.set_training_rows(trainable=st_t, rendered=st_t)
    fm.set_lrs(dict(xyz=0.0), f_dc=0.1, f_rest=0.1, opacity=0.1, scaling=0.1, rotation=0.1)
    fm.begin_mapping_call(reset_optimizer=True)
    init = dict(xyz=fm.init_xyz.cpu().numpy().astype(np.float64), scaling=fm.init_scaling.cpu().numpy().astype(np.float64),
                rotation=fm.init_rotation.cpu().numpy().astype(np.float64))
    attach_rows = fm.attach_mask.cpu().numpy().astype(bool)
    base_masks = [f["render_mask"].clone() for f in frames]
    fm.capture_window(frames, loss_tap=True, fused_tail=True)
    rows = np.nonzero(stable)[0]
    sub_of_row = np.full(P, -1)
    sub_of_row[rows] = np.arange(len(rows))
    report = []
    for it, k in enumerate([0, 1, 1]):
        cam, fr = cams[k], frames[k]
        s0 = _state(fm)
        sca, out = _oracle_iteration(torch, ol, fm, cam, None, rows, s0, None, None, None, init, fm.lrs)
        hr = U.HipRun(cam, sca, grad=False)  # (the eager op on the stable rows alone: what the hidden rows must leave)
        names = U.HipRun.names
        bad = U.flipped_pixels(hr.res, {n: getattr(out["f32"][1], n) for n in names}, {n: getattr(out["f64"][1], n) for n in names})
        mask = base_masks[k].cpu().numpy().astype(bool) & ~bad
        fm.set_frame(k, render_mask=torch.tensor(mask, device=dev))
        fm.replay(frame=k)
        torch.cuda.synchronize()
        assert not fm.graph_overflowed()
        # the graph's forward is the render of the stable cloud alone: radii of hidden rows are 0, the image equals the eager op's
        o = fm._g.out
        assert int(o[8][~st_t].abs().max().item()) == 0
        assert np.array_equal(o[0].cpu().numpy(), hr.res["color"]) and np.array_equal(o[3].cpu().numpy() >= 0, hr.res["hit_depth"] >= 0)
        s1 = _state(fm)
        loss, grads = _finish_oracle(ol, out, s0, rows, mask, fr["gt_color"].cpu().numpy(), fr["gt_depth"].cpu().numpy(), init, attach_rows)
        np.testing.assert_allclose(fm.loss[:3].double().cpu().numpy(), loss, rtol=2e-5)
        _check_iteration(s0, s1, it + 1, stable, grads, fm.lrs, sub_of_row, report)
        assert np.array_equal(s0["xyz"], s1["xyz"])  # lr 0 (mapper.py:1120)
    print("global optimisation", report)
    # the reference's second half (mapper.py:1188-1204): a RANDOM keyframe's camera and target under the LAST keyframe's masks
    sched = FusedMapper.window_schedule(12, 2, random.Random(3), global_opt=True)
    assert all(not isinstance(k, tuple) for k in sched[:7]) and (0, 1) in sched[7:] and all(k in (1, (0, 1)) for k in sched[7:]), sched
    fm.set_frame(0, render_mask=base_masks[0]), fm.set_frame(1, render_mask=base_masks[1])
    twin = FusedMapper(sc, frames[0]["settings"], dev)
    twin.set_training_rows(trainable=st_t, rendered=st_t)
    twin.set_lrs(dict(fm.lrs))
    for k in twin._params():
        twin._params()[k].copy_(fm._params()[k])
    twin.begin_mapping_call(reset_optimizer=True), fm.begin_mapping_call(reset_optimizer=True)
    mixed = dict(frames[0], render_mask=frames[1]["render_mask"], tile_mask=frames[1].get("tile_mask"))
    twin.capture_window([mixed], loss_tap=True, fused_tail=True)      # frame 0's camera and images, frame 1's masks — captured directly
    fm.replay(frame=(0, 1)), twin.replay(frame=0)
    torch.cuda.synchronize()
    assert (0, 1) in fm._mixed and not fm.graph_overflowed() and torch.equal(fm.loss, twin.loss)
    for k in fm._params():
        assert torch.equal(fm._params()[k], twin._params()[k]), k


def test_set_frame_rewrites_a_captured_frame_in_place(env):
    """The next mapping call's window reuses the captured graphs: new camera, images and masks are written into the frame's buffers
    (FusedMapper.set_frame) — the replay then equals a mapper captured on the new frame from scratch, bit for bit."""
    torch, _ = env
    from dqo_harness.fused_mapping import FusedMapper
    P = 12000
    sc, cams, frames, dev = _window_problem(torch, P, 2, seed=1)
    a = FusedMapper(sc, frames[0]["settings"], dev)
    a.capture_window([frames[0]], loss_tap=True, fused_tail=True, capacity_margin=2.0)
    a.set_frame(0, gt_color=frames[1]["gt_color"], gt_depth=frames[1]["gt_depth"], render_mask=frames[1]["render_mask"], settings=frames[1]["settings"])
    b = FusedMapper(sc, frames[1]["settings"], dev)
    b.capture_window([frames[1]], loss_tap=True, fused_tail=True)
    for _ in range(3):
        a.replay(frame=0), b.replay(frame=0)
    torch.cuda.synchronize()
    assert not a.graph_overflowed() and not b.graph_overflowed() and a.step_count == b.step_count == 3
    for k in a._params():
        assert torch.equal(a._params()[k], b._params()[k]), k
    assert torch.equal(a.confidence, b.confidence) and torch.equal(a.loss, b.loss)
    for x, y in zip(a._g.out, b._g.out):
        assert torch.equal(x, y)


def test_run_window_recaptures_a_frame_that_outgrew_its_capacities(env):
    """run_window(): one frame of the window is captured with room for 5 % of its candidate pairs — its replays are invalid frames, no-ops
    for the optimiser (parameters, moments, confidence, step count untouched); run_window notices from the device step count, captures
    THAT frame again with room and replays the lost iterations on it: the call delivers len(schedule) valid iterations, and the
    frames that were fine keep their graphs."""
    torch, _ = env
    from dqo_harness.fused_mapping import FusedMapper
    P = 12000
    sc, cams, frames, dev = _window_problem(torch, P, 3, seed=5)
    fm = FusedMapper(sc, frames[0]["settings"], dev)
    fm.begin_mapping_call(reset_optimizer=True)
    fm.capture_window(frames, loss_tap=True, fused_tail=True)
    good = [fm._frames[0], fm._frames[2]]
    snap = fm._snapshot_state()
    f1 = frames[1]
    fm.capture(f1["gt_color"], f1["gt_depth"], f1["render_mask"], settings=f1["settings"], frame=1, capacity_margin=0.05)  # too small
    fm._restore_state(snap)
    assert fm.step_count == 0
    sched = [0, 1, 2, 1, 1, 0, 2, 1, 0, 2]
    n_recap = fm.run_window(sched, check_every=4)
    torch.cuda.synchronize()
    assert n_recap >= 1 and fm.step_count == len(sched) and int(fm._step_dev.item()) == len(sched) + 1
    assert fm._frames[0] is good[0] and fm._frames[2] is good[1]
    assert not any(fm.graph_overflowed(g) for g in fm._frames)
    conf = fm.confidence.cpu().numpy()
    assert conf.max() <= len(sched) and conf.max() >= 3


def test_in_place_growth_keeps_the_captured_window(env):
    """A growth step between two mapping calls (FusedMapper.grow with spare rows: in place) under a captured WINDOW: every frame's graph
    stays valid, the new Gaussians are rendered and trained by the next call (row flags 0, confidence 0), deleted rows become spare rows
    (hidden + frozen for every camera of the window, whatever side of it they are parked on)."""
    torch, _ = env
    import _dqo_native as N
    from dqo_harness.fused_mapping import FusedMapper
    P = 12000
    sc, cams, frames, dev = _window_problem(torch, P, 2, seed=6)
    fm = FusedMapper(sc, frames[0]["settings"], dev)
    fm.reserve(2000)
    assert int((fm.row_flags[P:] == (N.ROW_HIDDEN | N.ROW_FROZEN)).all())
    fm.begin_mapping_call(reset_optimizer=True)
    fm.capture_window(frames, loss_tap=True, fused_tail=True, capacity_margin=1.5)
    graphs = list(fm._frames)
    for k in (0, 1, 1, 0):
        fm.replay(frame=k)
    rng = np.random.default_rng(8)
    src = rng.choice(P, 700, replace=False)
    new = dict(xyz=(sc["xyz"][src] + rng.normal(0, 0.02, (700, 3))).astype(np.float32), scales=sc["scales"][src], rotations=sc["rotations"][src],
               opacity=sc["opacity"][src], shs=sc["shs"][src])
    delete = torch.zeros(fm.P, dtype=torch.bool, device=dev)
    delete[torch.tensor(rng.choice(P, 300, replace=False), device=dev)] = True
    st = fm.grow(new, delete_mask=delete, new_mapping_call=True)
    assert st["in_place"] and st["deleted"] == 300 and st["added"] > 0
    assert all(a is b for a, b in zip(fm._frames, graphs)) and not any(g.stale for g in fm._frames)
    rows = st["rows"]
    assert int(fm.row_flags[rows].max()) == 0 and float(fm.confidence[rows].abs().max()) == 0
    gone = delete.clone()
    gone[rows] = False  # (spare rows are handed out lowest index first: the new Gaussians took some of the rows just freed)
    assert int(gone.sum()) > 0 or st["added"] >= 300
    assert int((fm.row_flags[gone] == (N.ROW_HIDDEN | N.ROW_FROZEN)).all()) and float(fm.confidence[delete].abs().max()) == 0
    before = fm.xyz[rows].clone()
    for k in (1, 0, 1):
        fm.replay(frame=k)
    torch.cuda.synchronize()
    assert not any(fm.graph_overflowed(g) for g in fm._frames) and fm.step_count == 3
    moved = (fm.xyz[rows] != before).any(1)
    seen = fm._frames[1].out[8][rows] > 0
    assert bool(seen.any()) and bool(moved[seen].float().mean() > 0.9)          # the new Gaussians train
    spare = fm.alive == 0
    assert int(fm._frames[1].out[8][spare].abs().max()) == 0 and int(fm._frames[0].out[8][spare].abs().max()) == 0   # spare rows are not rendered
    assert float(fm.confidence[rows].max()) >= 1


def test_run_graphs_of_several_iterations_are_the_single_launches_bit_for_bit(env):
    """capture_window(run_unroll=k): the stretches of a schedule that stay on one frame (the second half of local_optimize: the newest
    frame only, mapper.py:574-576) go as launches of k iterations each — the same kernels in the same order as one launch per iteration:
    parameters, moments, confidence, step count and losses must be the same bits."""
    import random
    torch, _ = env
    from dqo_harness.fused_mapping import FusedMapper
    P = 12000
    sc, cams, frames, dev = _window_problem(torch, P, 3, seed=7)
    sched = FusedMapper.window_schedule(21, 3, random.Random(2))
    assert sched[-9:] == [2] * 9
    res = []
    for k in (1, 4):
        fm = FusedMapper(sc, frames[0]["settings"], dev)
        fm.begin_mapping_call(reset_optimizer=True)
        fm.capture_window(frames, loss_tap=True, fused_tail=True, run_unroll=k)
        assert (fm._frames[2].run_graph is not None) == (k > 1)
        assert fm.run_window(sched, check_every=8) == 0
        torch.cuda.synchronize()
        assert fm.step_count == len(sched) and int(fm._step_dev.item()) == len(sched) + 1
        res.append(([v.clone() for _, v in sorted(fm._params().items())], [m.clone() for _, (m, v) in sorted(fm.state.items())],
                    [v.clone() for _, (m, v) in sorted(fm.state.items())], fm.confidence.clone(), fm.loss.clone()))
    for a, b in zip(res[0], res[1]):
        if isinstance(a, list):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
        else:
            assert torch.equal(a, b)
