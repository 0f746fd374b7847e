"""GPU: the opt-in fused Functions for callers on the reference's API (dqo_harness.fused_ops) against the eager torch statements of
Mapping.loss_update they replace (dqo_harness.mapping.mapping_loss / attach_loss): values and gradients."""
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


def test_masked_mapping_loss_function_matches_the_eager_loss(env):
    torch = env
    from dqo_harness import mapping, fused_ops
    g = torch.Generator(device="cuda").manual_seed(0)
    H, W = 123, 211
    render = torch.rand((3, H, W), device="cuda", generator=g, requires_grad=True)
    depth = (torch.rand((1, H, W), device="cuda", generator=g) * 2.5 + 0.5).requires_grad_(True)
    gt_color, gt_depth = torch.rand((3, H, W), device="cuda", generator=g), torch.rand((1, H, W), device="cuda", generator=g) * 2.6 + 0.4
    gt_depth[0, :7] = 0
    idx = torch.randint(-1, 50, (1, H, W), device="cuda", generator=g, dtype=torch.int32)
    mask = torch.rand((H, W), device="cuda", generator=g) < 0.6
    out = dict(render=render, depth=depth, depth_index_map=idx)
    ref, rparts = mapping.mapping_loss(out, gt_color, gt_depth, render_mask=mask)
    gr = torch.autograd.grad(2.5 * ref, [render, depth])
    got, gparts = fused_ops.masked_mapping_loss(out, gt_color, gt_depth, mask)
    gg = torch.autograd.grad(2.5 * got, [render, depth])
    np.testing.assert_allclose(got.item(), ref.item(), rtol=2e-6)
    for k in ("total_loss", "color_loss", "depth_loss"):
        np.testing.assert_allclose(gparts[k].item(), rparts[k].item(), rtol=2e-6)
    for a, b in zip(gg, gr):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-6, atol=1e-12)


def test_fused_attach_loss_function_matches_the_eager_loss(env):
    torch = env
    from dqo_harness import mapping, fused_ops
    cam, scene = scenes.make_config(1, P=5003)
    dev = torch.device("cuda")
    params = mapping.GaussianParams(scene, dev)
    init_stat = params.init_stat()
    with torch.no_grad():  # the parameters have moved since the start of the mapping call
        g = torch.Generator(device="cuda").manual_seed(1)
        params._xyz += 0.01 * torch.randn(params._xyz.shape, device=dev, generator=g)
        params._scaling += 0.02 * torch.randn(params._scaling.shape, device=dev, generator=g)
        params._rotation += 0.03 * torch.randn(params._rotation.shape, device=dev, generator=g)
    ref = mapping.attach_loss(params, init_stat)
    gr = torch.autograd.grad(ref, [params._scaling, params._xyz, params._rotation])
    aset = fused_ops.AttachSet(init_stat)
    assert 0 < aset.count < 5003
    got = fused_ops.fused_attach_loss(params._scaling, params._xyz, params._rotation, aset)
    gg = torch.autograd.grad(got, [params._scaling, params._xyz, params._rotation])
    np.testing.assert_allclose(got.item(), ref.item(), rtol=1e-5)
    for a, b in zip(gg, gr):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-9)
    # an empty attach set: zero loss, zero gradients (mapper.py:813)
    init_stat["opacity"].fill_(10.0)
    e = fused_ops.AttachSet(init_stat)
    assert e.count == 0
    z = fused_ops.fused_attach_loss(params._scaling, params._xyz, params._rotation, e)
    gz = torch.autograd.grad(z, [params._scaling, params._xyz, params._rotation])
    assert z.item() == 0.0 and all(float(t.abs().max()) == 0.0 for t in gz)


def test_dqo_adam_is_torch_adam(env):
    """DqoAdam against torch.optim.Adam on the reference's six parameter groups (gaussian_pointcloud.py:331-378: per-group lr, eps =
    1e-15), ten steps with fresh random gradients, one group without a gradient in some steps (torch skips it and its step count):
    parameters and both moments within float rounding of torch's (whose foreach / fused paths do not agree with each other bit for
    bit either), and the state keys are torch's."""
    torch = env
    from dqo_harness.fused_ops import DqoAdam
    dev = torch.device("cuda")
    gen = torch.Generator(device="cuda").manual_seed(3)
    P = 20011
    shapes = dict(xyz=(P, 3), f_dc=(P, 1, 3), f_rest=(P, 15, 3), opacity=(P, 1), scaling=(P, 3), rotation=(P, 4))
    lrs = dict(xyz=1e-3, f_dc=2.5e-3, f_rest=1.25e-4, opacity=5e-2, scaling=5e-3, rotation=1e-3)

    def build(cls):
        ps = {k: torch.nn.Parameter(torch.randn(s, device=dev, generator=torch.Generator(device="cuda").manual_seed(hash(k) % 1000))) for k, s in shapes.items()}
        opt = cls([dict(params=[ps[k]], lr=lrs[k], name=k) for k in shapes], lr=0.0, eps=1e-15)
        return ps, opt

    pa, oa = build(torch.optim.Adam)
    pb, ob = build(DqoAdam)
    for k in shapes:
        assert torch.equal(pa[k], pb[k])
    for it in range(10):
        for k in shapes:
            g = torch.randn(shapes[k], device=dev, generator=gen) * (10.0 ** (it % 4 - 2))
            if k == "opacity" and it in (3, 4):
                pa[k].grad = pb[k].grad = None  # no gradient this step: skipped by both, its step count does not advance
                continue
            pa[k].grad, pb[k].grad = g.clone(), g.clone()
        oa.step(), ob.step()
    for k in shapes:
        assert torch.allclose(pa[k], pb[k], rtol=2e-6, atol=1e-7), (k, (pa[k] - pb[k]).abs().max().item())
        sa, sb = oa.state[pa[k]], ob.state[pb[k]]
        assert int(sa["step"]) == int(sb["step"]) == (8 if k == "opacity" else 10)
        # (an exp_avg element near zero is a cancelled sum: an absolute bar of a few ulps of the gradients' size)
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=1e-6) and torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-12)
    # a non-contiguous gradient, an odd size (scalar tail of the float4 path) and a learning-rate change between steps
    q = torch.nn.Parameter(torch.randn(1001, 3, device=dev))
    r = torch.nn.Parameter(q.detach().clone())
    o1, o2 = torch.optim.Adam([q], lr=1e-2), DqoAdam([r], lr=1e-2)
    for it in range(3):
        g = torch.randn(3, 1001, device=dev).t()
        q.grad, r.grad = g, g
        for o in (o1, o2):
            o.param_groups[0]["lr"] = 1e-2 / (it + 1)
        o1.step(), o2.step()
    assert torch.allclose(q, r, rtol=2e-6, atol=1e-7)


def test_fused_ssim_against_the_reference_fixture(env):
    """tests/golden/loss_golden.npz: values and gradients of the reference's own utils/loss_utils.py::ssim (make_loss_golden.py)."""
    torch = env
    import os
    from dqo_harness import fused_ops
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_golden.npz"))
    for k in range(3):
        a = torch.tensor(d[f"ssim{k}_img1"], device="cuda", requires_grad=True)
        b = torch.tensor(d[f"ssim{k}_img2"], device="cuda")
        s = fused_ops.fused_ssim(a, b)
        (g,) = torch.autograd.grad(1 - s, [a])  # the fixture holds the gradient of `1 - ssim`, the loss term
        np.testing.assert_allclose(s.item(), float(d[f"ssim{k}_value"]), rtol=0, atol=2e-6)
        ref = d[f"ssim{k}_grad"]
        # fp32 windows summed in another order than conv2d's: compare against the gradient's own scale; pair 2 (identical images) has
        # a true gradient of zero: the fixture holds 3e-8 of rounding noise there, the kernel 5e-8 (tools/diag_ssim.py: both against float64)
        assert np.abs(g.cpu().numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-7, (k, np.abs(g.cpu().numpy() - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("H,W", [(48, 64), (123, 211), (16, 16), (7, 5), (480, 640)])
def test_fused_ssim_matches_eager_ssim_on_ragged_sizes(env, H, W):
    """Sizes that are not multiples of the 16 x 16 tile and smaller than the window: value and gradient against the eager statement
    (dqo_harness.mapping.ssim, itself pinned by the fixture) evaluated in float64."""
    torch = env
    from dqo_harness import mapping, fused_ops
    g = torch.Generator(device="cuda").manual_seed(H * 1000 + W)
    gt = torch.rand((3, H, W), device="cuda", generator=g)
    img = (gt + 0.15 * torch.randn((3, H, W), device="cuda", generator=g)).clamp(0, 1).requires_grad_(True)
    i64 = img.detach().double().requires_grad_(True)
    w64 = mapping._gaussian_window(11, 1.5, 3, "cuda").double()
    F = torch.nn.functional
    c = lambda x: F.conv2d(x[None], w64, padding=5, groups=3)
    mu1, mu2 = c(i64), c(gt.double())
    s1, s2, s12 = c(i64 * i64) - mu1 * mu1, c(gt.double() ** 2) - mu2 * mu2, c(i64 * gt.double()) - mu1 * mu2
    ref = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    (gr,) = torch.autograd.grad(ref, [i64])
    s = fused_ops.fused_ssim(img, gt)
    (gg,) = torch.autograd.grad(3.0 * s, [img])
    np.testing.assert_allclose(s.item(), ref.item(), rtol=0, atol=3e-6)
    err = (gg.double() / 3.0 - gr).abs().max().item()
    assert err <= 2e-4 * gr.abs().max().item() + 1e-7, (err, gr.abs().max().item())
    # and the fp32 eager statement is no closer to float64 than the kernel by more than rounding
    e32 = mapping.ssim(img, gt)
    assert abs(s.item() - ref.item()) <= abs(e32.item() - ref.item()) + 2e-6


def test_unmasked_mapping_loss_function_carries_the_ssim_term(env):
    torch = env
    from dqo_harness import mapping, fused_ops
    g = torch.Generator(device="cuda").manual_seed(3)
    H, W = 90, 130
    gt_color = torch.rand((3, H, W), device="cuda", generator=g)
    render = (gt_color + 0.1 * torch.randn((3, H, W), device="cuda", generator=g)).clamp(0, 1).requires_grad_(True)
    depth = (torch.rand((1, H, W), device="cuda", generator=g) * 2.5 + 0.5).requires_grad_(True)
    gt_depth = torch.rand((1, H, W), device="cuda", generator=g) * 2.6 + 0.4
    idx = torch.randint(-1, 50, (1, H, W), device="cuda", generator=g, dtype=torch.int32)
    out = dict(render=render, depth=depth, depth_index_map=idx)
    ref, rparts = mapping.mapping_loss(out, gt_color, gt_depth, render_mask=None)
    gr = torch.autograd.grad(ref, [render, depth])
    got, gparts = fused_ops.masked_mapping_loss(out, gt_color, gt_depth, None)
    gg = torch.autograd.grad(got, [render, depth])
    assert rparts["ssim_loss"].item() > 0.01
    for k in ("total_loss", "color_loss", "depth_loss", "ssim_loss"):
        np.testing.assert_allclose(gparts[k].item(), rparts[k].item(), rtol=1e-5)
    np.testing.assert_allclose(gg[1].cpu().numpy(), gr[1].cpu().numpy(), rtol=2e-6, atol=1e-12)
    a, b = gg[0].cpu().numpy(), gr[0].cpu().numpy()
    assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max()
