"""CPU: the decision of Mapping.temp_points_attach (mapper.py:1384-1430) — dqo_mapgrowth.temp_points_attach_indices (torch gathers,
device-agnostic) against the per-point restatement in oracle/map_oracle.py."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dqo-map_amd")]
from oracle import map_oracle as mo  # noqa: E402


def _case(seed, n, some_attached):
    rng = np.random.default_rng(seed)
    W, H, S = 64, 48, 300
    K = np.array([[50.0, 0, 31.5], [0, 50.0, 23.5], [0, 0, 1]], np.float32)
    a = 0.2
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3], w2c[:3, 3] = R, [0.1, -0.05, 0.3]
    stable_xyz = rng.uniform(-1, 1, (S, 3)).astype(np.float32) + np.array([0, 0, 2.5], np.float32)
    nrm = rng.normal(size=(S, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    index_map = rng.integers(0, S, (1, H // 4, W // 4)).repeat(4, 1).repeat(4, 2).astype(np.int32)
    index_map[0][rng.uniform(size=(H, W)) < 0.3] = -1
    # temp points: most of them on or near the plane of the stable Gaussian their pixel shows, some far off, some outside the image
    z = rng.uniform(1.0, 4.0, n)
    u, v = rng.uniform(-6, W + 6, n), rng.uniform(-6, H + 6, n)
    pc = np.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], 1)
    xyz = ((pc - w2c[:3, 3]) @ R).astype(np.float32)  # R^T (pc - t)
    for i in range(n):
        ui, vi = int(u[i]), int(v[i])
        if 0 <= ui < W and 0 <= vi < H and index_map[0, vi, ui] >= 0 and rng.uniform() < 0.6:
            s = index_map[0, vi, ui]
            off = nrm[s] * np.dot(stable_xyz[s] - xyz[i], nrm[s])  # move onto the plane, then a little off it
            xyz[i] = xyz[i] + off * np.float32(1.0 - rng.uniform(-0.02, 0.02))
    opacity = np.full((n, 1), 0.99, np.float32)
    if some_attached:
        opacity[rng.uniform(size=n) < 0.2] = 0.1  # exercises the reference's filtered / unfiltered index mix (quirk B15)
    return W, H, K, w2c, stable_xyz, nrm, index_map, xyz, opacity


def test_temp_points_attach_indices_vs_oracle():
    import dqo_mapgrowth
    for seed, n, some in ((0, 400, False), (1, 700, True), (2, 3, False)):
        W, H, K, w2c, sx, sn, im, xyz, op = _case(seed, n, some)
        t = torch.tensor
        got = dqo_mapgrowth.temp_points_attach_indices(t(xyz), t(op), t(w2c), t(K), W, H, t(im), t(sx), t(sn), 0.1).numpy()
        want = mo.temp_points_attach_indices(xyz, op, w2c, K, W, H, im, sx, sn, 0.1)
        # (a point whose projection lands within float rounding of a pixel border, or whose plane distance is within rounding of the
        # threshold, may differ between torch's batched matmul and the per-point loop: none in these seeded cases)
        np.testing.assert_array_equal(got, want)
        if n > 100:
            assert 0 < len(want) < n
    # no stable hit anywhere / every temp point already attached
    W, H, K, w2c, sx, sn, im, xyz, op = _case(3, 50, False)
    t = torch.tensor
    assert dqo_mapgrowth.temp_points_attach_indices(t(xyz), t(op), t(w2c), t(K), W, H, t(np.full_like(im, -1)), t(sx), t(sn), 0.1).numel() == 0
    assert dqo_mapgrowth.temp_points_attach_indices(t(xyz), t(np.full_like(op, 0.1)), t(w2c), t(K), W, H, t(im), t(sx), t(sn), 0.1).numel() == 0


def test_per_object_mask_form_equals_the_chain():
    """The per-object job's decision as ONE mask over the candidates (dqo_mapgrowth.temp_points_pixels +
    temp_points_attach_mask_per_object: the statements csrc/map_attach.hip follows) against the reference's boolean-index chain with
    temp_obj / stable_obj: the same candidates attach, except where the chain's matmul projection and the element-wise one put a candidate
    into different pixels (none expected at these sizes; allowed for)."""
    import dqo_mapgrowth
    t = torch.tensor
    for seed, n in ((5, 500), (6, 900)):
        W, H, K, w2c, sx, sn, im, xyz, op = _case(seed, n, False)
        rng = np.random.default_rng(seed + 100)
        op[rng.uniform(size=n) < 0.15] = 0.05      # below unstable_opacity_low: never attached
        sobj = rng.integers(0, 4, sx.shape[0]).astype(np.int32)
        tobj = rng.integers(0, 4, n).astype(np.int32)
        # most candidates carry the object of the Gaussian their pixel shows (otherwise nothing would attach)
        u = (t(xyz) @ t(w2c)[:3, :3].T + t(w2c)[:3, 3]) @ t(K).T
        uv_mm = (u[:, :2] / u[:, 2:]).long()
        ok = (uv_mm[:, 0] >= 0) & (uv_mm[:, 0] < W) & (uv_mm[:, 1] >= 0) & (uv_mm[:, 1] < H)
        for i in np.nonzero(ok.numpy())[0]:
            s = im[0, uv_mm[i, 1], uv_mm[i, 0]]
            if s >= 0 and rng.uniform() < 0.8:
                tobj[i] = sobj[s]
        weight = np.where(im >= 0, 0.5, 0.0).astype(np.float32)
        im0 = im.copy()
        im0[0, :4, :4] = 0      # a never-rendered tile: index 0 with weight 0 is the op's zero fill, no hit
        weight[0, :4, :4] = 0
        uv, inside = dqo_mapgrowth.temp_points_pixels(t(xyz), t(w2c), t(K), W, H)
        mask = dqo_mapgrowth.temp_points_attach_mask_per_object(t(xyz), t(op), t(tobj), uv, inside, W, H, t(im0), t(weight), t(sx), t(sn),
                                                                t(sobj), 0.1, 0.1)
        cim = np.where((im0 == 0) & (weight == 0), -1, im0).astype(np.int32)
        chain = dqo_mapgrowth.temp_points_attach_indices(t(xyz), t(op), t(w2c), t(K), W, H, t(cim), t(sx), t(sn), 0.1, 0.1,
                                                         temp_obj=t(tobj), stable_obj=t(sobj))
        c_mask = torch.zeros(n, dtype=torch.bool)
        c_mask[chain] = True
        moved = (uv_mm != uv).any(dim=1) & inside
        assert not bool(((c_mask != mask) & ~moved).any()) and int(moved.sum()) <= 2
        assert 0 < int(mask.sum()) < n and bool((mask & (t(op).reshape(-1) <= 0.1)).sum() == 0)
        assert not bool(inside.all()) and bool((uv[inside] >= 0).all())
