"""GPU: the drop-in operator inside a caller's torch.cuda.graph (set_sync_mode('graph')): one whole mapping iteration of the reference's
loop — activations, the op, the loss, autograd, the optimiser — captured once and replayed, against the same iterations issued eagerly."""
import numpy as np
import pytest

from dqo_harness import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture()
def env():
    import torch
    assert torch.cuda.is_available()
    import _dqo_native
    _dqo_native.lib()
    import diff_gaussian_rasterization_depth as dgr
    yield torch, dgr
    dgr.set_sync_mode("exact")


def _problem(torch, P=6000):
    from dqo_harness import mapping
    cam, scene = scenes.make_config(1, P=P)
    scene = {k: v for k, v in scene.items() if k != "normals"}  # (the reference's normal gather indexes with a boolean mask: a host sync)
    dev = torch.device("cuda")
    settings = mapping.make_settings(cam, dev)
    rng = np.random.default_rng(3)
    pert = dict(scene)
    pert["xyz"] = (scene["xyz"] + rng.normal(0, 0.004, scene["xyz"].shape)).astype(np.float32)
    pert["shs"] = scene["shs"].copy()
    pert["shs"][:, 0, :] += rng.normal(0, 0.15, (P, 3)).astype(np.float32)
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(pert, dev).activated())
    mask = torch.tensor(rng.uniform(size=(cam.H, cam.W)) < 0.8, device=dev) & (tgt["depth_index_map"][0] >= 0)
    return scene, settings, tgt["render"].clone(), tgt["depth"].clone(), mask, dev


def _loop(torch, dgr, scene, settings, gt_color, gt_depth, mask, dev, optin, n_iters, graph):
    from dqo_harness import mapping, fused_ops
    params = mapping.GaussianParams(scene, dev)
    if optin:
        opt = fused_ops.DqoAdam(params.param_groups(), lr=0.0, eps=1e-15, capturable=True)
        aset = fused_ops.AttachSet(params.init_stat())
    else:
        opt = torch.optim.Adam(params.param_groups(), lr=0.0, eps=1e-15, capturable=True)
    losses = torch.zeros(n_iters, device=dev)
    cell = torch.zeros((), device=dev)

    def iteration():
        out = mapping.render(settings, params.activated())
        if optin:
            loss, _ = fused_ops.masked_mapping_loss(out, gt_color, gt_depth, mask)
            total = loss + fused_ops.fused_attach_loss(params._scaling, params._xyz, params._rotation, aset)
        else:
            loss, _ = mapping.mapping_loss(out, gt_color, gt_depth, render_mask=mask)
            total = loss
        total.backward()
        opt.step()
        cell.copy_(loss.detach())

    dgr.set_sync_mode("lazy")
    if not graph:
        for it in range(n_iters):
            opt.zero_grad(set_to_none=True)
            iteration()
            losses[it] = cell
        dgr.verify_pending()
    else:
        if optin:  # the helper: warm-up iteration(s) on a side stream in 'lazy' mode, capture in 'graph' mode, mode restored
            cap = fused_ops.CapturedIteration(iteration, opt, warmup=1)
            assert dgr._sync_mode == "lazy"
            losses[0] = cell
            for it in range(1, n_iters):
                cap.replay()
                losses[it] = cell
            hdr = cap.check()
        else:  # ... and by hand
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # the warm-up iteration (also measures the op's capacity): iteration 0
                opt.zero_grad(set_to_none=True)
                iteration()
                losses[0] = cell
            torch.cuda.current_stream().wait_stream(side)
            dgr.verify_pending()
            dgr.set_sync_mode("graph")
            g = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):  # capturing runs nothing: iteration 1 is the first replay
                iteration()
            for it in range(1, n_iters):
                g.replay()
                losses[it] = cell
            hdr = dgr.last_header()
        assert hdr["overflow"] == 0 and hdr["num_rendered"] > 0
    torch.cuda.synchronize()
    return losses.cpu().numpy(), [p.detach().cpu().numpy() for p in (params._xyz, params._features_dc, params._features_rest,
                                                                     params._scaling, params._rotation)], opt


@pytest.mark.parametrize("optin", [False, True])
def test_a_captured_iteration_replays_like_the_eager_loop(env, optin):
    torch, dgr = env
    prob = _problem(torch)
    n = 6
    le, pe, _ = _loop(torch, dgr, *prob, optin, n, graph=False)
    lg, pg, opt = _loop(torch, dgr, *prob, optin, n, graph=True)
    assert le[0] > le[-1] > 0  # the loop trains
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    lrs = (0.001, 0.0005, 0.0005 / 20, 0.004, 0.001)
    for a, b, lr in zip(pg, pe, lrs):
        # (torch's capturable Adam forms its bias corrections in float32 on the device, eager and captured alike; the kernels are the
        # same launches either way — what may differ is the order of autograd's accumulations: compare against one Adam step)
        assert np.abs(a - b).max() <= 0.05 * lr + 1e-7, (np.abs(a - b).max(), lr)
    if optin:
        assert int(opt._step_dev.item()) == n


def test_a_capture_after_three_warm_up_iterations_runs_on_the_warm_ups_pooled_context(env):
    """The op's context pool in 'graph' mode: the captured forward takes the context the warm-up iterations ran on — out of the pool for
    good, owned by the CapturedIteration — with its tile order kept and its counters cleared by the (captured) backward: the short
    launch sequence at every replay.  The replays train like the eager loop."""
    torch, dgr = env
    from dqo_harness import mapping, fused_ops
    scene, settings, gt_color, gt_depth, mask, dev = _problem(torch, P=5000)
    n = 7

    def run(graph):
        params = mapping.GaussianParams(scene, dev)
        opt = fused_ops.DqoAdam(params.param_groups(), lr=0.0, eps=1e-15, capturable=True)
        cell = torch.zeros((), device=dev)

        def iteration():
            out = mapping.render(settings, params.activated())
            loss, _ = fused_ops.masked_mapping_loss(out, gt_color, gt_depth, mask)
            loss.backward()
            opt.step()
            cell.copy_(loss.detach())

        dgr.set_sync_mode("lazy")
        losses = []
        if graph:
            cap = fused_ops.CapturedIteration(iteration, opt, warmup=3)
            assert len(cap._contexts) == 1 and cap._contexts[0].captured == (1, 1) and cap._contexts[0].leased
            assert all(cap._contexts[0] is not cs for sets in dgr._pool.values() for cs in sets)
            for _ in range(n - 3):
                cap.replay()
                losses.append(float(cell))
            assert cap.check()["overflow"] == 0
        else:
            for it in range(n):
                opt.zero_grad(set_to_none=True)
                iteration()
                if it >= 3:
                    losses.append(float(cell))
            dgr.verify_pending()
        torch.cuda.synchronize()
        return np.array(losses), params._xyz.detach().cpu().numpy()

    le, xe = run(False)
    lg, xg = run(True)
    np.testing.assert_allclose(lg, le, rtol=2e-5)
    assert np.abs(xg - xe).max() <= 0.05 * 0.001 + 1e-7


def test_graph_mode_needs_a_known_capacity(env):
    torch, dgr = env
    from dqo_harness import mapping
    scene, settings, *_ = _problem(torch, P=777)  # a shape no earlier test has rendered in lazy mode
    params = mapping.GaussianParams(scene, torch.device("cuda"))
    dgr.set_sync_mode("graph")
    with pytest.raises(RuntimeError, match="no instance capacity known"):
        mapping.render(settings, params.activated())
    dgr.set_capacity(777, settings.image_width, settings.image_height, 200000)
    out = mapping.render(settings, params.activated())
    assert dgr.last_header()["overflow"] == 0 and float(out["render"].detach().sum()) > 0


def test_dqo_adam_capturable_is_dqo_adam(env):
    """The device-side step count: same element arithmetic, bias corrections formed by the kernel with the host path's expressions."""
    torch, _ = env
    from dqo_harness import fused_ops
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(1000, 3), (1000, 15, 3), (1000, 1), (7,)]
    mk = lambda: [torch.randn(s, device="cuda", generator=torch.Generator(device="cuda").manual_seed(i)).requires_grad_(True)
                  for i, s in enumerate(shapes)]
    pa, pb = mk(), mk()
    groups = lambda ps: [dict(params=[p], lr=lr) for p, lr in zip(ps, (0.001, 0.0005, 0.01, 0.1))]
    oa = fused_ops.DqoAdam(groups(pa), lr=0.0, eps=1e-15)
    ob = fused_ops.DqoAdam(groups(pb), lr=0.0, eps=1e-15, capturable=True)
    for it in range(12):
        grads = [torch.randn(s, device="cuda", generator=g) for s in shapes]
        for ps, o in ((pa, oa), (pb, ob)):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
            o.step()
    for a, b in zip(pa, pb):
        # (host pow() and the device's may differ in the last bit of a bias correction: one ulp of the step, not more)
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=1e-6, atol=1e-9)
    assert int(ob._step_dev.item()) == 12
