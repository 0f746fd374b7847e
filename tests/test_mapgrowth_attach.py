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
