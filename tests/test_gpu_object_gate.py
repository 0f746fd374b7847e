"""GPU: the object gate (DqoObjectGate, rasterize_gaussians_gated) and the per-object loss (DqoLossTap.per_object) — the pieces that
make the sharded mapping job ONE function for every number of shards: HIP against the gated oracle, shards against the
unsharded map, the loss tap against the eager torch statement of the per-object loss."""
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
    return torch


def _scene(P=30000, cfg=3):
    cam, sc = scenes.make_config(cfg, P=P)
    assert len(np.unique(sc["obj_id"])) >= 4
    # owner of every pixel: the object of the Gaussian that fixes its depth in the (ungated) render, -1 where nothing does
    res, _ = U.run_hip(cam, sc)
    hit = res["hit_depth"][0]
    go = np.asarray(sc["obj_id"], np.int32)
    po = np.where(hit >= 0, go[np.clip(hit, 0, None)], -1).astype(np.int32)
    return cam, sc, go, po


def test_gated_render_matches_the_gated_oracle(env, oracle):
    """Forward within 1e-4, flipped pixels within budget and masked out of dL on both sides, gradients within 1e-3 (north_star's bar,
    util_rast) — with the gate on in both the HIP op and the oracle."""
    cam, sc, go, po = _scene()
    rng = np.random.default_rng(5)
    dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))
    fs, gs = U.parity_case(oracle, cam, sc, dL, fp64=True, object_gate=(go, po))
    # the gate really gates: pixels without an owner render nothing, owned pixels differ from the ungated render somewhere
    hr = U.HipRun(cam, sc, grad=False, object_gate=(go, po))
    assert (hr.res["color"][:, po < 0] == 0).all() and (hr.res["T_map"][0][po < 0] == 1).all() and (hr.res["hit_depth"][0][po < 0] == -1).all()
    ungated, _ = U.run_hip(cam, sc)
    assert np.abs(hr.res["color"] - ungated["color"])[:, po >= 0].max() > 1e-3


def test_a_shard_renders_exactly_its_pixels_of_the_unsharded_map(env):
    """The property the sharded job rests on: with the gate, a call that holds only SOME objects' Gaussians produces, on the pixels those
    objects own, bit for bit what the call holding all of them produces — and the same gradients for its Gaussians."""
    cam, sc, go, po = _scene()
    rng = np.random.default_rng(6)
    dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))
    mine = np.isin(go, [0, 2, 5])
    own_px = np.isin(po, [0, 2, 5])
    dLm = (dL[0] * own_px, dL[1] * own_px)
    full = U.HipRun(cam, sc, object_gate=(go, po))
    gfull = full.backward(dLm, retain=False)
    sub = {k: (v[mine] if hasattr(v, "shape") and v.shape[:1] == (len(go),) else v) for k, v in sc.items()}
    part = U.HipRun(cam, sub, object_gate=(go[mine], po))
    gpart = part.backward(dLm, retain=False)
    for k in ("color", "depth", "hit_color_weight", "hit_depth_weight", "T_map"):
        assert np.array_equal(full.res[k][..., own_px], part.res[k][..., own_px]), k
    idx = np.nonzero(mine)[0]
    assert np.array_equal(idx[np.clip(part.res["hit_depth"][0][own_px], 0, None)] * (part.res["hit_depth"][0][own_px] >= 0),
                          full.res["hit_depth"][0][own_px] * (full.res["hit_depth"][0][own_px] >= 0))
    for k in gfull:
        a, b = gfull[k][mine], gpart[k]
        assert np.array_equal(a, b), (k, np.abs(a - b).max())
        if k != "colors":
            assert np.abs(gfull[k][~mine]).max() == 0  # the other objects get nothing from these pixels


def test_tile_object_sets_only_drop_entries_that_act_on_no_pixel(env):
    """DqoObjectGate.tile_objects (instances whose object owns no pixel of the tile are dropped at binning time): fewer list entries,
    and every output and the trained parameters bit for bit what the gate alone gives."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper, tile_object_sets
    cam, sc, go, po, settings, gt_color, gt_depth, dev = _mapping_problem(torch)
    ts = tile_object_sets(torch.tensor(po, device=dev))
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    want = np.zeros(gy * gx, np.uint64)
    for ty in range(gy):
        for tx in range(gx):
            for k in np.unique(po[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]):
                if k >= 0:
                    want[ty * gx + tx] |= np.uint64(1) << np.uint64(k)
    assert np.array_equal(ts.cpu().numpy().view(np.uint64), want)
    mask = torch.tensor(po >= 0, device=dev)
    a = FusedMapper(sc, settings, dev).set_object_gate(go, po)
    b = FusedMapper(sc, settings, dev).set_object_gate(go, po)
    b.tile_objects = None  # the gate alone
    a.capture(gt_color, gt_depth, mask), b.capture(gt_color, gt_depth, mask)
    for _ in range(3):
        a.replay(), b.replay()
    torch.cuda.synchronize()
    assert 0 < a.header()["num_rendered"] < b.header()["num_rendered"]
    for i in range(7):
        assert torch.equal(a._g.out[i], b._g.out[i]), i
    assert torch.equal(a.loss, b.loss)
    for k, pa in a._params().items():
        assert torch.equal(pa, b._params()[k]), k


def _mapping_problem(torch, P=30000):
    from dqo_harness import mapping
    cam, sc, go, po = _scene(P=P)
    dev = torch.device("cuda")
    settings = mapping.make_settings(cam, dev)
    rng = np.random.default_rng(3)
    pert = dict(sc)
    pert["xyz"] = (sc["xyz"] + rng.normal(0, 0.004, sc["xyz"].shape)).astype(np.float32)
    pert["shs"] = sc["shs"].copy()
    pert["shs"][:, 0, :] += rng.normal(0, 0.15, (P, 3)).astype(np.float32)
    gate = (torch.tensor(go, device=dev), torch.tensor(po, device=dev))
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(pert, dev).activated(), object_gate=gate)
    return cam, sc, go, po, settings, tgt["render"].clone(), tgt["depth"].clone(), dev


def test_per_object_loss_tap_matches_the_eager_statement(env):
    """DqoLossTap.per_object (sums and gradient scales per object id inside the blend kernels) against mapping.per_object_loss (eager
    torch + autograd) through the same gated op: the same loss to 1e-5 and parameters that stay within a fraction of an Adam step."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, sc, go, po, settings, gt_color, gt_depth, dev = _mapping_problem(torch)
    mask = torch.tensor(po >= 0, device=dev)
    a = FusedMapper(sc, settings, dev).set_object_gate(go, po)
    b = FusedMapper(sc, settings, dev).set_object_gate(go, po)
    a.capture(gt_color, gt_depth, mask)
    losses = [a.loss[:3].tolist()]
    b.step(gt_color, gt_depth, mask)
    np.testing.assert_allclose(losses[0], b.loss[:3].tolist(), rtol=1e-5)
    for _ in range(3):
        a.replay(), b.step(gt_color, gt_depth, mask)
    torch.cuda.synchronize()
    np.testing.assert_allclose(a.loss[:3].tolist(), b.loss[:3].tolist(), rtol=3e-4)
    for k, pa in a._params().items():
        lr = dict(xyz=0.001, shs=0.0005, opacity=1.0, scaling=0.004, rotation=0.001)[k]
        assert ((pa - b._params()[k]).abs() > 0.02 * 3 * lr + 1e-7).float().mean().item() < 2e-3, k


def test_shards_of_one_map_add_up_to_the_unsharded_job(env):
    """ONE map trained as a whole and as two shards (objects {0, 2, 5, 7} / the rest), each shard on its own mask: the shards' losses
    add up to the whole map's loss at every iteration, and every Gaussian ends bit for bit where it ends in the whole map — the
    sharded job computes the N = 1 function."""
    torch = env
    from dqo_harness.fused_mapping import FusedMapper
    cam, sc, go, po, settings, gt_color, gt_depth, dev = _mapping_problem(torch)
    objs_a = [0, 2, 5, 7]
    in_a = np.isin(go, objs_a)
    P = len(go)
    sub = lambda m: {k: (v[m] if hasattr(v, "shape") and v.shape[:1] == (P,) else v) for k, v in sc.items()}
    whole = FusedMapper(sc, settings, dev).set_object_gate(go, po)
    # (the attach loss is a mean over the attach set of the WHOLE map: the shards divide by the whole map's count)
    n_attach = whole.attach_count
    assert 0 < n_attach < P
    sa = FusedMapper(sub(in_a), settings, dev, attach_count_reducer=lambda n: n_attach).set_object_gate(go[in_a], po)
    sb = FusedMapper(sub(~in_a), settings, dev, attach_count_reducer=lambda n: n_attach).set_object_gate(go[~in_a], po)
    owned = lambda ids: torch.tensor(np.isin(po, ids), device=dev)
    all_ids = sorted(set(np.unique(go).tolist()))
    whole.capture(gt_color, gt_depth, owned(all_ids))
    sa.capture(gt_color, gt_depth, owned(objs_a))
    sb.capture(gt_color, gt_depth, owned([k for k in all_ids if k not in objs_a]))
    for it in range(4):
        if it:
            whole.replay(), sa.replay(), sb.replay()
        torch.cuda.synchronize()
        lw, la, lb = whole.loss[:3].double(), sa.loss[:3].double(), sb.loss[:3].double()
        np.testing.assert_allclose((la + lb).tolist(), lw.tolist(), rtol=2e-6)
    ia, ib = torch.tensor(np.nonzero(in_a)[0], device=dev), torch.tensor(np.nonzero(~in_a)[0], device=dev)
    for k, pw in whole._params().items():
        assert torch.equal(pw[ia], sa._params()[k]), k
        assert torch.equal(pw[ib], sb._params()[k]), k
    np.testing.assert_allclose((sa.attach_loss() + sb.attach_loss()).item(), whole.attach_loss().item(), rtol=1e-5)


def test_shards_of_one_map_add_up_under_the_window_schedule(env):
    """The N-invariance of the per-object job holds for the reference's LOOP as well (round 6): three frames — own camera, target and
    pixel -> owner map each — replayed in the reference's schedule (mapper.py:570-576), a third of the rows frozen (the stable cloud),
    the confidence counter on.  The whole map against its two object shards: losses add up at every iteration, every Gaussian's
    parameters AND confidence end bit for bit where they end in the unsharded job."""
    import random
    torch = env
    from dqo_harness import mapping, scenes
    from dqo_harness.fused_mapping import FusedMapper
    cam0, sc, go, _ = _scene(P=24000)
    dev = torch.device("cuda")
    P = len(go)
    rng = np.random.default_rng(3)
    pert = dict(sc)
    pert["xyz"] = (sc["xyz"] + rng.normal(0, 0.004, sc["xyz"].shape)).astype(np.float32)
    pert["shs"] = sc["shs"].copy()
    pert["shs"][:, 0, :] += rng.normal(0, 0.15, (P, 3)).astype(np.float32)
    cams = [scenes.replica_camera(yaw=12.0 - 3.0 * k, pitch=4.0 + 0.7 * k, pos=(0.3 - 0.08 * k, 0.1, -1.85 + 0.05 * k)) for k in (2, 1, 0)]
    go_t = torch.tensor(go, device=dev)
    frames = []
    for cam in cams:
        st = mapping.make_settings(cam, dev)
        with torch.no_grad():  # the owner map of the frame: the object that fixes the pixel's depth in an ungated render of the target
            t0 = mapping.render(st, mapping.GaussianParams(pert, dev).activated())
            hit = t0["depth_index_map"][0]
            po = torch.where(hit >= 0, go_t[hit.long().clamp(min=0)], torch.full_like(hit, -1)).to(torch.int32).contiguous()
            tgt = mapping.render(st, mapping.GaussianParams(pert, dev).activated(), object_gate=(go_t, po))
        frames.append(dict(settings=st, gt_color=tgt["render"].clone(), gt_depth=tgt["depth"].clone(), pixel_object=po))
    objs_a = [0, 2, 5, 7]
    in_a = np.isin(go, objs_a)
    all_ids = sorted(set(np.unique(go).tolist()))
    trainable = rng.uniform(size=P) < 0.67
    sub = lambda m: {k: (v[m] if hasattr(v, "shape") and v.shape[:1] == (P,) else v) for k, v in sc.items()}

    def build(rows, ids, n_attach=None):
        fm = FusedMapper(sub(rows), frames[0]["settings"], dev, attach_count_reducer=(None if n_attach is None else (lambda n: n_attach)))
        fm.set_object_gate(go[rows], frames[0]["pixel_object"])
        fm.set_training_rows(trainable=torch.tensor(trainable[rows], device=dev))
        fm.begin_mapping_call(reset_optimizer=True)
        fr = [dict(f, render_mask=torch.isin(f["pixel_object"], torch.tensor(ids, device=dev, dtype=torch.int32)).to(torch.uint8).contiguous())
              for f in frames]
        return fm, fr

    everything = np.ones(P, bool)
    whole, fw = build(everything, all_ids)
    n_attach = whole.attach_count
    assert 0 < n_attach < int(trainable.sum())
    sa, fa = build(in_a, objs_a, n_attach)
    sb, fb = build(~in_a, [k for k in all_ids if k not in objs_a], n_attach)
    for fm, fr in ((whole, fw), (sa, fa), (sb, fb)):
        fm.capture_window(fr, loss_tap=True, fused_tail=True)
    sched = FusedMapper.window_schedule(8, 3, random.Random(2))
    for k in sched:
        whole.replay(frame=k), sa.replay(frame=k), sb.replay(frame=k)
        torch.cuda.synchronize()
        assert not (whole.graph_overflowed() or sa.graph_overflowed() or sb.graph_overflowed())
        np.testing.assert_allclose((sa.loss[:3].double() + sb.loss[:3].double()).tolist(), whole.loss[:3].double().tolist(), rtol=2e-6)
    ia, ib = torch.tensor(np.nonzero(in_a)[0], device=dev), torch.tensor(np.nonzero(~in_a)[0], device=dev)
    for k, pw in whole._params().items():
        assert torch.equal(pw[ia], sa._params()[k]) and torch.equal(pw[ib], sb._params()[k]), k
    assert torch.equal(whole.confidence[ia], sa.confidence) and torch.equal(whole.confidence[ib], sb.confidence)
    assert float(whole.confidence.max()) >= 4 and float(whole.confidence[torch.tensor(~trainable, device=dev)].max()) == 0
