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
