"""Pins the numpy oracle of the fused mapping helpers against torch (CPU): the reference implements these pieces WITH torch
eager ops and torch.optim.Adam, so torch itself is the reference here."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import map_oracle as mo


def _data(seed=0, H=40, W=56):
    rng = np.random.default_rng(seed)
    color, gt_color = rng.uniform(0, 1, (3, H, W)), rng.uniform(0, 1, (3, H, W))
    depth, gt_depth = rng.uniform(0.5, 3, (1, H, W)), rng.uniform(0.4, 3, (1, H, W))
    gt_depth[0, :3] = 0  # invalid gt depth
    idx = rng.integers(-1, 50, (1, H, W)).astype(np.int32)
    mask = rng.uniform(size=(H, W)) < 0.7
    color[0, 5, 5] = gt_color[0, 5, 5]  # exact zero residual: sign(0) = 0
    return color, depth, idx, gt_color, gt_depth, mask


def test_masked_loss_matches_torch_autograd():
    color, depth, idx, gt_color, gt_depth, mask = _data()
    tc = torch.tensor(color, dtype=torch.float64, requires_grad=True)
    td = torch.tensor(depth, dtype=torch.float64, requires_grad=True)
    m = torch.tensor(mask)
    gc, gd, ti = torch.tensor(gt_color), torch.tensor(gt_depth), torch.tensor(idx)
    # literal restatement of mapper.py:836-875 for the masked case
    image, dep, dindex = tc.permute(1, 2, 0), td.permute(1, 2, 0), ti.permute(1, 2, 0)
    color_loss = torch.abs(image[m] - gc.permute(1, 2, 0)[m]).mean()
    depth_error = dep - gd.permute(1, 2, 0)
    valid = (dindex != -1).squeeze() & (gd.permute(1, 2, 0) > 0).squeeze() & (depth_error < 0.1).squeeze() & m
    depth_loss = torch.abs(depth_error[valid]).mean()
    total = 1.0 * depth_loss + 0.8 * color_loss
    total.backward()
    t, cl, dl, dC, dD = mo.masked_loss(color, depth, idx, gt_color, gt_depth, mask)
    np.testing.assert_allclose([t, cl, dl], [total.item(), color_loss.item(), depth_loss.item()], rtol=1e-12)
    np.testing.assert_allclose(dC, tc.grad.numpy(), atol=1e-15)
    np.testing.assert_allclose(dD, td.grad.numpy(), atol=1e-15)


def test_activation_jacobians_and_adam_match_torch():
    rng = np.random.default_rng(1)
    P = 64
    raw = dict(op=rng.normal(size=(P, 1)), sc=rng.normal(-4, 0.5, (P, 3)), rot=rng.normal(size=(P, 4)))
    g = dict(op=rng.normal(size=(P, 1)), sc=rng.normal(size=(P, 3)), rot=rng.normal(size=(P, 4)))
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in raw.items()}
    act = (torch.sigmoid(t["op"]), torch.exp(t["sc"]), F.normalize(t["rot"]))
    (act[0] * torch.tensor(g["op"])).sum().backward()
    (act[1] * torch.tensor(g["sc"])).sum().backward()
    (act[2] * torch.tensor(g["rot"])).sum().backward()
    a = mo.activate(raw["op"], raw["sc"], raw["rot"])
    for x, y in zip(a, act):
        np.testing.assert_allclose(x, y.detach().numpy(), rtol=1e-12)
    rg = mo.raw_grads(raw["op"], raw["sc"], raw["rot"], g["op"], g["sc"], g["rot"])
    for x, k in zip(rg, ("op", "sc", "rot")):
        np.testing.assert_allclose(x, t[k].grad.numpy(), rtol=1e-10, atol=1e-14)
    # three Adam steps vs torch.optim.Adam(eps=1e-15) with the reference's group lrs
    p = torch.tensor(raw["sc"], dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([{"params": [p], "lr": 0.004}], lr=0.0, eps=1e-15)
    pn, m, v = raw["sc"].copy(), np.zeros_like(raw["sc"]), np.zeros_like(raw["sc"])
    for step in range(1, 4):
        gr = rng.normal(size=(P, 3))
        p.grad = torch.tensor(gr)
        opt.step()
        pn, m, v = mo.adam_step(pn, gr, m, v, 0.004, step)
        np.testing.assert_allclose(pn, p.detach().numpy(), rtol=1e-12, atol=1e-15)


def test_accumulate_error_oracle_small_hand_case():
    """map_process.cu:33-110 on a 2x3 image worked by hand."""
    ce = np.array([[0.5, 0.1, -1.0], [0.3, 0.9, 0.2]], np.float32)
    de = np.array([[0.05, 0.2, 0.0], [0.15, 0.01, 0.3]], np.float32)
    ne = np.zeros((2, 3), np.float32)
    ci = np.array([[0, 0, 1], [-1, 2, 2]], np.int32)
    di = np.array([[1, 1, 1], [5, 0, -1]], np.int32)  # 5 is out of range for P = 3
    gc, gd, gn, rs = mo.accumulate_gaussian_error(2, 3, 3, ce, de, ne, ci, di, 0.25, 0.1, 0.3, True)
    np.testing.assert_allclose(gc[:, 0], [0.5, 0.0, 0.9])   # Gaussian 1 only saw a negative error: stays at the 0 init
    np.testing.assert_allclose(gd[:, 0], [0.01, 0.2, 0.0])
    np.testing.assert_allclose(rs[:, 0], [1 + 0, 0 + 1, 1])  # colour > .25: g0 (0.5), g2 (0.9); depth > .1: g1 (0.2)
    gc, gd, gn, rs2 = mo.accumulate_gaussian_error(2, 3, 3, ce, de, ne, ci, di, 0.25, 0.1, 0.3, False)
    np.testing.assert_allclose(gc[:, 0], [0.3, -1.0, 0.55], rtol=1e-6)
    np.testing.assert_allclose(gd[:, 0], [0.01, (0.05 + 0.2 + 0.0) / 3, 0.0], rtol=1e-6)
    np.testing.assert_array_equal(rs, rs2)


def test_accumulate_confidence_oracle_small_hand_case():
    """map_process.cu:247-360 on a 2x3 image worked by hand: strict comparisons (NaN never wins), 0 / 0 / 0 for unseen Gaussians."""
    conf = np.array([[0.5, -0.25, 2.0], [np.nan, 0.75, -1.5]], np.float32)
    idx = np.array([[0, 0, 2], [2, 7, 0]], np.int32)  # 7 is out of range for P = 4; Gaussians 1 and 3 are never named
    gmax, gmin, mean = mo.accumulate_gaussian_confidence(2, 3, 4, idx, conf)
    np.testing.assert_array_equal(gmax[:, 0], np.array([0.5, 0.0, 2.0, 0.0], np.float32))
    np.testing.assert_array_equal(gmin[:, 0], np.array([-1.5, 0.0, 2.0, 0.0], np.float32))
    np.testing.assert_allclose(mean[[0, 1, 3], 0], [(0.5 - 0.25 - 1.5) / 3, 0.0, 0.0], rtol=1e-6)
    assert np.isnan(mean[2, 0])  # the NaN pixel enters the sum (atomicAdd) but neither the max nor the min


def test_zero_moment_rows_without_gradient_are_fixed_points_of_adam():
    """The property the exact sparse Adam (DqoAdamStep.moment_live) rests on, checked against torch.optim.Adam itself with the
    reference's eps = 1e-15 (mapper.py:548): a row whose moments are zero and whose gradient is zero does not move by a single
    bit, for any number of steps — so not touching it is the dense update."""
    rng = np.random.default_rng(5)
    p0 = rng.normal(size=(64, 3)).astype(np.float32)
    p = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([p], lr=1e-3, eps=1e-15)
    dormant = np.arange(64) % 3 != 0  # two thirds of the rows never get a gradient
    for step in range(1, 6):
        g = rng.normal(size=(64, 3)).astype(np.float32)
        g[dormant] = 0.0
        p.grad = torch.tensor(g)
        opt.step()
    st = opt.state[p]
    assert np.array_equal(p.detach().numpy()[dormant], p0[dormant])
    assert (st["exp_avg"].numpy()[dormant] == 0).all() and (st["exp_avg_sq"].numpy()[dormant] == 0).all()
    assert not np.array_equal(p.detach().numpy()[~dormant], p0[~dormant])
    # and the numpy restatement agrees: one dense step over a dormant row returns its inputs
    q, m, v = mo.adam_step(p0[dormant].astype(np.float64), 0.0, 0.0, 0.0, 1e-3, 7)
    assert np.array_equal(q, p0[dormant].astype(np.float64)) and np.all(m == 0) and np.all(v == 0)
