"""Pins for the CPU oracle of the rasteriser (the reference ships no tests for it: 'parity unpinned').

(1) analytic known-answer cases, (2) fp64 finite differences of the oracle's own forward,
(3) cross-checks against restated pure-python reference helpers (eval_sh layout, cov3D = R S S^T R^T).
"""
import math

import numpy as np
import pytest

from dqo_harness import scenes

NT60 = math.cos(math.radians(60.0))


def settings(ol, cam, **kw):
    d = dict(normal_threshold=NT60)
    d.update(kw)
    return ol.RastSettings(cam.W, cam.H, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, **d)


def run(ol, cam, sc, dtype=np.float32, tile_mask=None, colors_precomp=None, **kw):
    o = ol.OracleRasterizer(dtype)
    st = settings(ol, cam, **kw)
    r = o.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                  shs=None if colors_precomp is not None else sc["shs"], colors_precomp=colors_precomp,
                  scales=sc["scales"], rotations=sc["rotations"], tile_mask=tile_mask)
    return o, r


def active_mask(o, cam):
    gx = (cam.W + 15) // 16
    act = np.zeros((cam.H, cam.W), bool)
    for t in o.ctx("tile_indices"):
        ty, tx = divmod(int(t), gx)
        act[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    return act


def one_gaussian(z=2.0, s=(0.05, 0.05, 0.005), rgb=(0.9, 0.5, 0.1), opac=0.99, xy=(0.0, 0.0)):
    sh = np.zeros((1, 16, 3), np.float32)
    sh[0, 0] = scenes.rgb_to_sh(np.array(rgb))
    return dict(xyz=np.array([[xy[0], xy[1], z]], np.float32), scales=np.array([s], np.float32),
                rotations=np.array([[1, 0, 0, 0]], np.float32), opacity=np.array([[opac]], np.float32), shs=sh)


def test_kat_single_gaussian_on_axis(oracle):
    """Isotropic-in-plane surfel on the optical axis facing the camera: closed-form alpha, radius, rect, depth."""
    W, H, f = 64, 48, 50.0
    cam = scenes.Camera(W, H, f, f, 31.5, 23.5)
    sc = one_gaussian(s=(0.2, 0.2, 0.005))
    o, r = run(oracle, cam, sc)
    # EWA: cov2D = (f*s/z)^2 + 0.3 on both axes, no off-diagonal
    var = (f * 0.2 / 2.0) ** 2 + 0.3
    radius = math.ceil(3.0 * math.sqrt(var))
    assert r.radii[0] == radius
    # pixel centre = ndc*S/2 + c with ndc = 0  ->  (cx, cy)
    np.testing.assert_allclose(o.ctx("means2D")[0], [31.5, 23.5], atol=1e-5)
    np.testing.assert_allclose(o.ctx("conic_opacity")[0], [1 / var, 0, 1 / var, 0.99], rtol=1e-5, atol=1e-7)
    # tile rect: int((31.5 - r)/16) .. int((31.5 + r + 15)/16)
    x0, x1 = max(0, int((31.5 - radius) / 16)), min(4, int((31.5 + radius + 15) / 16))
    y0, y1 = max(0, int((23.5 - radius) / 16)), min(3, int((23.5 + radius + 15) / 16))
    assert r.num_rendered == (x1 - x0) * (y1 - y0)
    # pixel (31, 23): d = (0.5, 0.5)
    alpha = min(0.99, 0.99 * math.exp(-0.5 * (0.25 + 0.25) / var))
    px = r.color[:, 23, 31]
    np.testing.assert_allclose(px, np.array([0.9, 0.5, 0.1]) * alpha, rtol=2e-5)
    assert r.hit_depth[0, 23, 31] == 0 and r.hit_color[0, 23, 31] == 0
    np.testing.assert_allclose(r.hit_color_weight[0, 23, 31], alpha, rtol=1e-5)
    np.testing.assert_allclose(r.T_map[0, 23, 31], 1 - alpha, rtol=1e-4)
    # surfel normal = local z = camera z, plane z = 2  ->  ray-plane depth = 2 exactly on every covered pixel
    hit = (r.hit_depth[0] >= 0) & active_mask(o, cam)
    assert hit.sum() > 10
    np.testing.assert_allclose(r.depth[0][hit], 2.0, rtol=1e-5)
    # far corner pixel is untouched by the Gaussian but its tile is active -> rendered, no hit: ids = -1
    assert r.hit_depth[0, 0, 0] == -1 and r.hit_color[0, 0, 0] == -1 and r.T_map[0, 0, 0] == 1.0
    # n_touched counts pairs with T' > 0.5  (alpha < 0.5)
    ys, xs = np.mgrid[0:H, 0:W]
    a_all = np.minimum(0.99, 0.99 * np.exp(-0.5 * ((31.5 - xs) ** 2 + (23.5 - ys) ** 2) / var))
    expect = int(((a_all >= 1 / 255) & ((1 - a_all) > 0.5) & active_mask(o, cam)).sum())
    assert abs(int(r.n_touched[0]) - expect) <= 2  # float32 exp at the 0.5 / 1/255 boundaries


def test_kat_two_opaque_surfels_nearer_wins(oracle):
    W, H, f = 48, 48, 40.0
    cam = scenes.Camera(W, H, f, f, 23.5, 23.5)
    a, b = one_gaussian(z=3.0, rgb=(1, 0, 0)), one_gaussian(z=1.5, rgb=(0, 1, 0))
    sc = {k: np.concatenate([a[k], b[k]]) for k in a}
    o, r = run(oracle, cam, sc)
    assert r.hit_depth[0, 23, 23] == 1  # id of the nearer surfel
    np.testing.assert_allclose(r.depth[0, 23, 23], 1.5, rtol=1e-5)
    # sorted order inside the centre tile: near first
    rng_ = o.ctx("ranges")[1 * 3 + 1]
    pl = o.ctx("point_list")[rng_[0]:rng_[1]]
    assert list(pl) == [1, 0]
    v1, v2 = (f * 0.05 / 1.5) ** 2 + 0.3, (f * 0.05 / 3.0) ** 2 + 0.3
    a1, a2 = 0.99 * math.exp(-0.25 / v1), 0.99 * math.exp(-0.25 / v2)
    np.testing.assert_allclose(r.color[:, 23, 23], [a2 * (1 - a1), a1, 0], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(r.T_map[0, 23, 23], (1 - a1) * (1 - a2), rtol=1e-4)
    np.testing.assert_allclose(r.hit_depth_weight[0, 23, 23], a1, rtol=1e-5)


def test_kat_tilted_surfel_rayplane_depth(oracle):
    """Surfel tilted about x: depth follows t*ray.z of the ray-plane hit, exact against a direct computation."""
    W, H, f = 64, 64, 60.0
    cam = scenes.Camera(W, H, f, f, 31.5, 31.5)
    ang = math.radians(30)
    q = np.array([[math.cos(ang / 2), math.sin(ang / 2), 0, 0]], np.float32)
    sc = one_gaussian(z=2.0, s=(0.2, 0.2, 0.01))
    sc["rotations"] = q
    o, r = run(oracle, cam, sc)
    n = np.array([0, -math.sin(ang), math.cos(ang)])  # R(q) column 2
    pc = np.array([0, 0, 2.0])
    for (u, v) in [(31, 31), (20, 40), (45, 25)]:
        if r.hit_depth[0, v, u] < 0:
            continue
        ray = np.array([(u - 31.5) / f, (v - 31.5) / f, 1.0])
        ray /= np.linalg.norm(ray)
        t = n @ pc / (n @ ray + 1e-8)
        assert abs(t * ray[2] - pc[2]) <= 0.2 * 1.0  # inside the depth window, so the ray-plane branch is taken
        np.testing.assert_allclose(r.depth[0, v, u], t * ray[2], rtol=2e-5)


def test_kat_opaque_branch_when_grazing(oracle):
    """|n.ray| < cos(60deg): depth falls back to the centre depth p_view.z."""
    W, H, f = 64, 64, 60.0
    cam = scenes.Camera(W, H, f, f, 31.5, 31.5)
    ang = math.radians(75)
    sc = one_gaussian(z=2.0, s=(0.2, 0.2, 0.01))
    sc["rotations"] = np.array([[math.cos(ang / 2), math.sin(ang / 2), 0, 0]], np.float32)
    o, r = run(oracle, cam, sc)
    hit = (r.hit_depth[0] >= 0) & active_mask(o, cam)
    assert hit.sum() > 0
    np.testing.assert_allclose(r.depth[0][hit], 2.0, rtol=1e-6)


def test_kat_transparent_scene_and_empty(oracle):
    W, H, f = 40, 24, 30.0
    cam = scenes.Camera(W, H, f, f, 19.5, 11.5)
    sc = one_gaussian(opac=0.001)  # alpha < 1/255 everywhere
    o, r = run(oracle, cam, sc, bg=(0.2, 0.3, 0.4))
    act = active_mask(o, cam)
    assert act.any()
    np.testing.assert_allclose(r.color[:, act], np.array([[0.2], [0.3], [0.4]]) * np.ones((1, act.sum())), rtol=1e-6)
    assert (r.hit_depth[0][act] == -1).all() and (r.T_map[0][act] == 1).all()
    # pixels of never-rendered tiles keep the initial fills: colour 0 (NOT bg), ids 0 (quirk B7), T 1
    assert (r.color[:, ~act] == 0).all() and (r.hit_depth[0][~act] == 0).all() and (r.T_map[0][~act] == 1).all()
    # P == 0 (B9)
    e = {k: v[:0] for k, v in sc.items()}
    o2, r2 = run(oracle, cam, e)
    assert r2.num_rendered == 0 and (r2.color == 0).all() and (r2.T_map == 1).all() and (r2.hit_color == 0).all()
    g = o2.backward(np.ones((3, H, W)), np.ones((1, H, W)))
    assert g.means3D.shape == (0, 3)


def test_kat_tile_mask(oracle):
    cam, sc = scenes.make_config(1, P=1500)
    o_full, r_full = run(oracle, cam, sc)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    mask = np.ones((gy, gx), np.int32)
    mask[:, gx // 2:] = 0
    o, r = run(oracle, cam, sc, tile_mask=mask)
    pm = np.repeat(np.repeat(mask, 16, 0), 16, 1)[:cam.H, :cam.W].astype(bool)
    # unmasked tiles render identically; masked tiles keep the initial fills
    np.testing.assert_array_equal(r.color[:, pm], r_full.color[:, pm])
    np.testing.assert_array_equal(r.hit_depth[0][pm], r_full.hit_depth[0][pm])
    assert (r.color[:, ~pm] == 0).all() and (r.depth[0][~pm] == 0).all() and (r.T_map[0][~pm] == 1).all()
    assert (r.hit_depth[0][~pm] == 0).all()
    assert r.num_rendered < r_full.num_rendered
    assert (mask.ravel()[o.ctx("point_tile")] == 1).all()


def test_binning_invariants(oracle):
    cam, sc = scenes.make_config(1, P=3000)
    o, r = run(oracle, cam, sc)
    pl, pt, rg = o.ctx("point_list"), o.ctx("point_tile"), o.ctx("ranges")
    d = o.ctx("depths")
    assert r.num_rendered == int(o.ctx("tiles_touched").sum()) == len(pl)
    assert (np.diff(pt.astype(np.int64)) >= 0).all()
    for t in o.ctx("tile_indices"):
        s, e = rg[t]
        assert e > s and (pt[s:e] == t).all()
        dd, ids = d[pl[s:e]], pl[s:e].astype(np.int64)
        key = dd.view(np.uint32).astype(np.int64) * (1 << 32) + ids
        assert (np.diff(key) > 0).all()  # sorted by (depth bits, id): stable sort semantics
    act = set(int(t) for t in o.ctx("tile_indices"))
    for t in range(rg.shape[0]):
        if t not in act:
            assert rg[t][0] == rg[t][1]


def test_sh_eval_matches_reference_formula(oracle):
    """computeColorFromSH vs a direct restatement of utils/sh_utils.py:57-121 eval_sh (layout [..., C, K] there,
    [P, K, 3] in the kernel)."""
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
    C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435]
    cam, sc = scenes.make_config(1, P=400)
    rng = np.random.default_rng(3)
    sc["shs"] = rng.normal(0, 0.3, sc["shs"].shape).astype(np.float32)
    o, r = run(oracle, cam, sc, dtype=np.float64)
    vis = r.radii > 0
    d = sc["xyz"].astype(np.float64) - cam.camera_center.astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    sh = sc["shs"].astype(np.float64)
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    res = (C0 * sh[:, 0] - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3] + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5]
           + C2[2] * (2 * zz - xx - yy) * sh[:, 6] + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8]
           + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
           + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
           + C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15]) + 0.5
    np.testing.assert_allclose(o.ctx("rgb")[vis], np.maximum(res, 0)[vis], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(o.ctx("clamped")[vis].astype(bool), (res < 0)[vis])


def test_cov3d_matches_build_covariance(oracle):
    """cov3D vs R diag(s)^2 R^T (utils/general_utils.py:108-150 build_covariance_from_scaling_rotation, restated)."""
    cam, sc = scenes.make_config(1, P=300)
    o, r = run(oracle, cam, sc, dtype=np.float64)
    q = sc["rotations"].astype(np.float64)
    rr, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - rr * z), 2 * (x * z + rr * y), 2 * (x * y + rr * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - rr * x), 2 * (x * z - rr * y), 2 * (y * z + rr * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    L = R * sc["scales"].astype(np.float64)[:, None, :]
    S = L @ L.transpose(0, 2, 1)
    vis = r.radii > 0
    exp = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
    np.testing.assert_allclose(o.ctx("cov3D")[vis], exp[vis], rtol=1e-10, atol=1e-14)


def test_mark_visible(oracle):
    cam, sc = scenes.make_config(2, P=5000)
    vis = oracle.mark_visible(sc["xyz"], cam.world_view_transform, cam.full_proj_transform)
    pc = sc["xyz"].astype(np.float64) @ cam.Rw2c.T + cam.t
    ndc_x = pc[:, 0] / pc[:, 2] / cam.tanfovx
    ndc_y = pc[:, 1] / pc[:, 2] / cam.tanfovy
    exp = (pc[:, 2] > 0.2) & (np.abs(ndc_x) <= 1.3) & (np.abs(ndc_y) <= 1.3)
    border = (np.abs(np.abs(ndc_x) - 1.3) < 1e-4) | (np.abs(np.abs(ndc_y) - 1.3) < 1e-4) | (np.abs(pc[:, 2] - 0.2) < 1e-4)
    assert (vis == exp)[~border].all()
    assert 0.2 < vis.mean() < 0.95


def _fd_scene(seed, P=40):
    """Tiny scene for finite differences, built so that no discrete decision sits near a threshold."""
    W, H, f = 48, 32, 40.0
    cam = scenes.Camera(W, H, f, f, 23.3, 15.6, scenes.rot_yx(5.0, -3.0), np.array([0.02, -0.01, 0.05]))
    sc = scenes.frustum_cloud(seed, P, cam, zmin=1.0, zmax=3.0)
    rng = np.random.default_rng(seed + 100)
    sc["scales"] = (sc["scales"] * 6).astype(np.float32)  # big soft splats: many overlapping semi-transparent layers
    sc["opacity"] = rng.uniform(0.15, 0.95, sc["opacity"].shape).astype(np.float32)
    sc["shs"][:, 1:, :] = rng.normal(0, 0.08, sc["shs"][:, 1:, :].shape)
    return cam, sc


@pytest.mark.parametrize("seed", [11, 12])
def test_fd_gradients_fp64(oracle, seed):
    """Analytic backward of the oracle (fp64 instantiation) vs central differences of its forward.

    The reference's backward deliberately omits some terms (no gradient through the 0.99 clamp being active, none
    through the hit-selection, B2's T/end_T asymmetry with bg != 0); the scene uses bg = 0 and the comparison
    is restricted to inputs whose perturbation flips no discrete decision (checked via the index maps)."""
    cam, sc = _fd_scene(seed)
    st_kw = dict(bg=(0, 0, 0), opaque_threshold=0.6)
    rng = np.random.default_rng(seed)
    wC = rng.normal(size=(3, cam.H, cam.W))
    wD = rng.normal(size=(1, cam.H, cam.W))
    base = {k: sc[k].astype(np.float64) for k in ("xyz", "scales", "rotations", "opacity", "shs")}

    def fwd(p):
        o = oracle.OracleRasterizer(np.float64)
        st = settings(oracle, cam, **st_kw)
        r = o.forward(st, p["xyz"], p["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                      shs=p["shs"], scales=p["scales"], rotations=p["rotations"])
        return o, r

    o, r = fwd(base)
    L0 = (r.color * wC).sum() + (r.depth * wD).sum()
    g = o.backward(wC, wD)
    ana = dict(xyz=g.means3D, scales=g.scales, rotations=g.rotations, opacity=g.opacity, shs=g.sh)
    sig = (r.hit_depth.copy(), r.hit_color.copy(), o.ctx("n_contrib").copy(), r.radii.copy())
    # the 0.99 alpha clamp must be inactive everywhere for FD to agree (reference drops that Jacobian)
    checked = 0
    vis = np.nonzero(r.radii > 0)[0]
    assert len(vis) >= 10
    for name, eps in (("xyz", 1e-6), ("scales", 1e-7), ("rotations", 1e-6), ("opacity", 1e-6), ("shs", 1e-6)):
        for gi in vis[:6]:
            flat = base[name][gi].reshape(-1)
            for comp in range(min(flat.size, 5)):
                Ls = []
                ok = True
                for sgn in (+1, -1):
                    p = {k: v.copy() for k, v in base.items()}
                    p[name][gi].reshape(-1)[comp] += sgn * eps
                    o2, r2 = fwd(p)
                    s2 = (r2.hit_depth, r2.hit_color, o2.ctx("n_contrib"), r2.radii)
                    ok &= all(np.array_equal(a, b) for a, b in zip(sig, s2))
                    Ls.append((r2.color * wC).sum() + (r2.depth * wD).sum())
                if not ok:
                    continue
                fd = (Ls[0] - Ls[1]) / (2 * eps)
                an = ana[name][gi].reshape(-1)[comp]
                scale = max(abs(fd), abs(an), 1e-3 * np.abs(ana[name]).max())
                assert abs(fd - an) <= 2e-4 * scale + 1e-7, (name, gi, comp, fd, an)
                checked += 1
    assert checked > 60


def test_openmp_build_equals_serial(oracle):
    """oracle/_build/libdqo_oracle_omp.so (full-size GPU tests, cpu_baseline) runs the same statements as the serial oracle with
    the Gaussian / tile loops shared between cores: forward outputs are bitwise equal; the backward's per-Gaussian sums are double
    accumulators in both builds, so their fp32 roundings agree too (to the last bit on this scene)."""
    import util_rast as U
    cam, sc = scenes.make_config(2, P=6000)
    rng = np.random.default_rng(4)
    dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))
    st = U.oracle_settings(oracle, cam)
    res = []
    for omp in (False, True):
        o = oracle.OracleRasterizer(np.float32, omp=omp)
        r = o.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                      shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
        res.append((r, o.backward(dL[0], dL[1])))
    assert oracle.num_threads(True) >= 1 and oracle.num_threads(False) == 1
    for k in ("color", "depth", "hit_color", "hit_depth", "hit_color_weight", "hit_depth_weight", "T_map", "n_touched", "radii"):
        assert np.array_equal(getattr(res[0][0], k), getattr(res[1][0], k)), k
    for k in ("means3D", "sh", "opacity", "scales", "rotations", "colors"):
        a, b = getattr(res[0][1], k), getattr(res[1][1], k)
        assert np.abs(a - b).max() <= 1e-6 * np.abs(a).max(), k


def test_object_gate_of_the_oracle(oracle):
    """The oracle's object gate (test infrastructure of the sharded job, off by default): with every Gaussian in ONE object, gating equals
    masking — owned pixels render exactly as without the gate, pixels without an owner render nothing, and the backward equals the
    ungated backward of the incoming gradient restricted to the owned pixels.  With two objects a pixel only sees its owner's Gaussians:
    the gated render of the whole map equals, on object k's pixels, the ungated render of object k's Gaussians alone."""
    from dqo_harness import scenes
    import util_rast as U
    cam, sc = scenes.make_config(1, P=1500)
    P = 1500
    o, r, _ = U.run_oracle(oracle, cam, sc)
    owned = r["hit_depth"][0] >= 0
    go, po = np.zeros(P, np.int32), np.where(owned, 0, -1).astype(np.int32)
    og, rg, _ = U.run_oracle(oracle, cam, sc, object_gate=(go, po))
    for k in ("color", "depth", "hit_depth", "hit_color", "T_map"):
        assert np.array_equal(rg[k][..., owned], r[k][..., owned]), k
    assert (rg["color"][:, ~owned] == 0).all() and (rg["T_map"][0][~owned] == 1).all() and (rg["hit_depth"][0][~owned] == -1).all()
    rng = np.random.default_rng(1)
    dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))
    gg, gu = U.oracle_backward(og, dL), U.oracle_backward(o, (dL[0] * owned, dL[1] * owned))
    for k in gg:
        assert np.array_equal(gg[k], gu[k]), k
    # two objects
    go2 = (np.arange(P) % 2).astype(np.int32)
    po2 = np.where(owned, go2[np.clip(r["hit_depth"][0], 0, None)], -1).astype(np.int32)
    _, r2, _ = U.run_oracle(oracle, cam, sc, object_gate=(go2, po2))
    for k_obj in (0, 1):
        m = go2 == k_obj
        sub = {k: (v[m] if hasattr(v, "shape") and v.shape[:1] == (P,) else v) for k, v in sc.items()}
        _, rk, _ = U.run_oracle(oracle, cam, sub)
        px = po2 == k_obj
        assert np.array_equal(r2["color"][:, px], rk["color"][:, px]) and np.array_equal(r2["depth"][:, px], rk["depth"][:, px])


def test_prescaled_conic_gives_the_reference_power_bit_for_bit():
    """The blend kernels keep an entry's conic in LDS with A and C multiplied by -0.5 and evaluate
    (A' dx) dx + (C' dy) dy - (B dx) dy   (dqo_power_pre, csrc/dqo_cull.h)
    instead of the reference's -0.5 (A dx dx + C dy dy) - B dx dy (forward.cu:758-760, backward.cu:937-939; dqo_power): scaling by a
    power of two commutes with every IEEE rounding, so the two are the same float — checked here on a million draws of the ranges the
    blend loops see, with numpy's float32 operations (one rounding per operation, no contraction: the kernels compile both forms with
    fp contract off)."""
    rng = np.random.default_rng(7)
    n = 1_000_000
    f = np.float32
    A = rng.uniform(1e-3, 40.0, n).astype(f)
    C = rng.uniform(1e-3, 40.0, n).astype(f)
    B = (rng.uniform(-0.99, 0.99, n) * np.sqrt(A.astype(np.float64) * C)).astype(f)
    dx = rng.uniform(-40.0, 40.0, n).astype(f)
    dy = rng.uniform(-40.0, 40.0, n).astype(f)
    half = f(-0.5)
    ref = half * (A * dx * dx + C * dy * dy) - B * dx * dy
    Ah, Ch = half * A, half * C
    pre = (Ah * dx * dx + Ch * dy * dy) - B * dx * dy
    assert ref.dtype == np.float32 and pre.dtype == np.float32
    assert np.array_equal(ref.view(np.uint32), pre.view(np.uint32))
    # ... and on splats seen far from their centre, where the three terms cancel to a result 1e4 times smaller than they are
    t = rng.uniform(-30.0, 30.0, n).astype(f)
    dx2, dy2 = t, (t * f(0.999) + rng.uniform(-0.05, 0.05, n).astype(f)).astype(f)
    A2 = np.full(n, 25.0, f)
    C2 = np.full(n, 25.0, f)
    B2 = np.full(n, -24.99, f)
    ref2 = half * (A2 * dx2 * dx2 + C2 * dy2 * dy2) - B2 * dx2 * dy2
    pre2 = ((half * A2) * dx2 * dx2 + (half * C2) * dy2 * dy2) - B2 * dx2 * dy2
    assert np.array_equal(ref2.view(np.uint32), pre2.view(np.uint32))
