"""GPU parity tests (run on the MI355X box): the HIP rasteriser, called through the drop-in Python operator and therefore
through the C ABI of libdqoraster.so, against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from dqo_harness import scenes
import util_rast as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import _dqo_native
    _dqo_native.lib()  # fails loudly if the HIP extension is missing
    return torch


def _check(oracle, cam, sc, dL, **kw):
    """HIP vs the fp32 oracle, north_star's numbers with no slack: forward 1e-4, gradients 1e-3 (util_rast.parity_case: pixels whose
    discrete decisions flipped are counted against the budget and get a zero incoming gradient on both sides)."""
    hr = U.HipRun(cam, sc, **kw)
    o, r, _ = U.run_oracle(oracle, cam, sc, **kw)
    bad = U.flipped_pixels(hr.res, r)
    st = U.compare_forward(hr.res, r)
    keep = (~bad).astype(np.float32)
    dLm = (dL[0] * keep[None], dL[1] * keep[None])
    precomp = kw.get("colors_precomp") is not None
    o64, _, _ = U.run_oracle(oracle, cam, sc, dtype=np.float64, **kw)  # (only consulted for rows beyond the bar: compare_grads)
    gs = U.compare_grads(hr.backward(dLm, retain=False), U.oracle_backward(o, dLm, precomp), U.oracle_backward(o64, dLm, precomp))
    print("fwd", st, "grads (rel-to-max, q99)", gs)
    return hr.res, r, o


def _dL(cam, seed=0):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32)


@pytest.mark.parametrize("P", [3000, 10000])
def test_forward_backward_cfg1(torch_cuda, oracle, P):
    cam, sc = scenes.make_config(1, P=P)
    _check(oracle, cam, sc, _dL(cam))


def test_forward_backward_room_sh_rest(torch_cuda, oracle):
    """Surfel room (cfg-2 layout at reduced P), non-zero higher-order SH, non-zero background."""
    cam, sc = scenes.make_config(2, P=20000)
    rng = np.random.default_rng(5)
    sc["shs"][:, 1:, :] = rng.normal(0, 0.05, sc["shs"][:, 1:, :].shape).astype(np.float32)
    _check(oracle, cam, sc, _dL(cam, 1), bg=(0.1, 0.2, 0.3))


def test_tile_mask_and_colors_precomp(torch_cuda, oracle):
    cam, sc = scenes.make_config(1, P=4000)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    mask = (np.random.default_rng(2).uniform(size=(gy, gx)) < 0.6).astype(np.int32)
    cp = np.random.default_rng(3).uniform(0, 1, (4000, 3)).astype(np.float32)
    h, r, o = _check(oracle, cam, sc, _dL(cam, 2), tile_mask=mask, colors_precomp=cp)
    pm = np.repeat(np.repeat(mask, 16, 0), 16, 1)[:cam.H, :cam.W].astype(bool)
    assert (h["color"][:, ~pm] == 0).all() and (h["T_map"][0][~pm] == 1).all() and (h["hit_depth"][0][~pm] == 0).all()


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_lower_sh_degree(torch_cuda, oracle, deg):
    cam, sc = scenes.make_config(1, P=2000)
    sc["shs"][:, 1:, :] = np.random.default_rng(7).normal(0, 0.1, sc["shs"][:, 1:, :].shape).astype(np.float32)
    _check(oracle, cam, sc, _dL(cam, 3), sh_degree=deg)


def test_backward_twice_on_one_forward(torch_cuda):
    """The backward is a function of its arguments only, like the reference's (which is stateless): a second backward over the same
    saved forward context (retain_graph=True; one autograd.grad call per loss term) must not see anything the first one left behind
    — in particular not the partial-gradient records of quadrants that now receive no gradient."""
    cam, sc = scenes.make_config(1, P=5000)
    dL = _dL(cam, 4)
    hr = U.HipRun(cam, sc)
    g_full = hr.backward(dL, retain=True)
    m = np.zeros((cam.H, cam.W), np.float32)
    m[:, : cam.W // 2] = 1  # the right half of the image sends no gradient in the second call
    dLm = (dL[0] * m, dL[1] * m)
    g2 = hr.backward(dLm, retain=True)
    fresh = U.HipRun(cam, sc).backward(dLm, retain=False)
    for k in g2:
        assert np.array_equal(g2[k], fresh[k]), k
    # ... and then the full gradient again: identical to the first call, bit for bit
    g3 = hr.backward(dL, retain=True)
    for k in g3:
        assert np.array_equal(g3[k], g_full[k]), k
    # one call per loss term (colour only, depth only) adds up to the joint call
    gc = hr.backward((dL[0], np.zeros_like(dL[1])), retain=True)
    gd = hr.backward((np.zeros_like(dL[0]), dL[1]), retain=False)
    for k in g_full:
        want = g_full[k].astype(np.float64)
        err = np.abs(gc[k].astype(np.float64) + gd[k] - want).max() / (np.abs(want).max() + 1e-30)
        assert err < 2e-4, (k, err)


def test_binning_exact(torch_cuda, oracle):
    """Integer / index work is bit-exact: radii, instance count, per-tile sorted id lists, ranges."""
    import torch
    import ctypes
    import _dqo_native as N
    from diff_gaussian_rasterization_depth import _RasterizeGaussians
    cam, sc = scenes.make_config(1, P=6000)
    o, r, _ = U.run_oracle(oracle, cam, sc)
    rs = U.raster_settings_torch(cam, "cuda")
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device="cuda")

    class Ctx:  # minimal stand-in for the autograd ctx to get at the saved buffers
        def save_for_backward(self, *a):
            self.saved = a

        def mark_non_differentiable(self, *a):
            pass

    c = Ctx()
    out = _RasterizeGaussians.forward(c, t(sc["xyz"]), t(sc["shs"]), torch.Tensor([]), t(sc["opacity"]), t(sc["scales"]),
                                      t(sc["rotations"]), torch.Tensor([]), None, rs)
    geom, binning, img = c.saved[8], c.saved[9], c.saved[10]
    Nn = c.num_rendered
    assert Nn == o.N  # exact mode sizes the buffers for the reference's num_rendered (every pair inside the tile rects)
    gx = (cam.W + 15) // 16
    T = gx * ((cam.H + 15) // 16)
    al = lambda n: (n + 255) // 256 * 256
    # binning layout: unsorted 16-byte list records [cap] | point_list u32[cap] | slot_list u32[cap], each 256-B aligned
    off_pl = al(16 * Nn)
    pl = binning[off_pl:off_pl + 4 * Nn].view(torch.int32).cpu().numpy().astype(np.uint32)
    # image layout: tile_count (one counter per 64 words) | tile_flag | tile_cursor (same stride) | ranges ...  (dqo_common.h)
    off_rg = 2 * al(4 * T * 64) + al(4 * T)
    rg = img[off_rg:off_rg + 8 * T].view(torch.int32).cpu().numpy().reshape(T, 2).astype(np.uint32)
    np.testing.assert_array_equal(out[8].cpu().numpy(), r["radii"])
    # Per tile the HIP list is the oracle's (= reference's) list with the dead entries removed, in the same order:
    # every instance the HIP binning dropped must be a provable no-op (alpha < 1/255 or power > 0 on every pixel of the
    # tile, evaluated with the blend loop's own fp32 arithmetic), so all outputs are unchanged (dqo_cull.h).
    opl, org = o.ctx("point_list"), o.ctx("ranges")
    m2d, con = o.ctx("means2D"), o.ctx("conic_opacity")
    dropped = kept = 0
    for t in range(T):
        mine = pl[rg[t, 0]:rg[t, 1]]
        ref = opl[org[t, 0]:org[t, 1]]
        it = iter(ref)
        assert all(any(x == y for y in it) for x in mine), f"tile {t}: HIP list is not an ordered sub-list of the reference list"
        gone = np.setdiff1d(ref, mine)
        kept += len(mine)
        dropped += len(gone)
        if len(gone):
            ty, tx = divmod(t, gx)
            ys, xs = np.mgrid[ty * 16:min(cam.H, ty * 16 + 16), tx * 16:min(cam.W, tx * 16 + 16)]
            xs, ys = xs.astype(np.float32).ravel(), ys.astype(np.float32).ravel()
            dx = m2d[gone, 0][:, None] - xs[None]
            dy = m2d[gone, 1][:, None] - ys[None]
            A, B, C, op = (con[gone, k][:, None] for k in range(4))
            power = np.float32(-0.5) * (A * dx * dx + C * dy * dy) - B * dx * dy
            alpha = np.minimum(np.float32(0.99), op * np.exp(power))
            live = (power <= 0) & (alpha >= np.float32(1.0 / 255.0))
            assert not live.any(), f"tile {t}: dropped a live instance"
    n_live = int(geom[:4].view(torch.int32).cpu()[0])  # device header: instances kept after the footprint test
    assert kept == n_live and kept + dropped == o.N
    print(f"binning: kept {kept} of {o.N} reference instances ({dropped / o.N:.1%} dead entries culled)")


def test_empty_and_tiny(torch_cuda, oracle):
    import torch
    cam, sc = scenes.make_config(1, P=50)
    e = {k: v[:0] for k, v in sc.items()}
    h, _ = U.run_hip(cam, e)
    assert (h["color"] == 0).all() and (h["T_map"] == 1).all() and (h["hit_depth"] == 0).all()
    _check(oracle, cam, sc, _dL(cam, 4))


def test_mark_visible(torch_cuda, oracle):
    import torch
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    cam, sc = scenes.make_config(2, P=20000)
    rast = GaussianRasterizer(U.raster_settings_torch(cam, "cuda"))
    vis = rast.markVisible(torch.tensor(sc["xyz"], device="cuda")).cpu().numpy()
    np.testing.assert_array_equal(vis, oracle.mark_visible(sc["xyz"], cam.world_view_transform, cam.full_proj_transform))


def test_errors(torch_cuda):
    import torch
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    cam, sc = scenes.make_config(1, P=10)
    rast = GaussianRasterizer(U.raster_settings_torch(cam, "cuda"))
    t = lambda a: torch.tensor(a, device="cuda")
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        rast(means3D=t(sc["xyz"]), opacities=t(sc["opacity"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        rast(means3D=t(sc["xyz"]), opacities=t(sc["opacity"]), shs=t(sc["shs"]))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        rast(means3D=t(sc["xyz"]).reshape(-1), opacities=t(sc["opacity"]), shs=t(sc["shs"]), scales=t(sc["scales"]),
             rotations=t(sc["rotations"]))
    with pytest.raises(RuntimeError, match="no CPU path"):
        rast(means3D=torch.tensor(sc["xyz"]), opacities=t(sc["opacity"]), shs=t(sc["shs"]), scales=t(sc["scales"]),
             rotations=t(sc["rotations"]))


def test_backward_deterministic(torch_cuda):
    """No global float atomics in the backward: two runs are bitwise identical (the reference's is order-dependent, B10)."""
    cam, sc = scenes.make_config(1, P=5000)
    dL = _dL(cam, 6)
    _, g1 = U.run_hip(cam, sc, dL=dL)
    _, g2 = U.run_hip(cam, sc, dL=dL)
    for k in g1:
        np.testing.assert_array_equal(g1[k], g2[k])


def test_late_part_placement_gives_identical_bits(torch_cuda):
    """The late part of the per-Gaussian forward (SH colour, its direction derivative, surfel normal) runs inside preprocess_kernel or in
    extra blocks of one of the two sort launches (csrc/dqo_k1_late.h, DQO_K1_WHERE): the same statements — outputs and gradients must agree bit for
    bit.  The switch is read once per process, so each placement renders in a child process and reports a digest."""
    import os, subprocess, sys
    code = r'''
import hashlib, sys, os
import numpy as np
sys.path[:0] = [os.environ["DQO_TEST_ROOT"], os.environ["DQO_TEST_ROOT"] + "/dqo-map_amd", os.environ["DQO_TEST_ROOT"] + "/tests"]
from dqo_harness import scenes
import util_rast as U
cam, sc = scenes.make_config(1, P=7000)
rng = np.random.default_rng(5)
dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))
res, grads = U.run_hip(cam, sc, dL=dL)
h = hashlib.sha256()
for k in sorted(res):
    h.update(np.ascontiguousarray(res[k]).tobytes())
for k in sorted(grads):
    h.update(np.ascontiguousarray(grads[k]).tobytes())
print("DIGEST", h.hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for where in ("0", "1", "2"):
        env = dict(os.environ, DQO_K1_WHERE=where, DQO_TEST_ROOT=root)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert digests[0] == digests[1] == digests[2]
