"""ctypes loader for the CPU oracle (oracle/_build/libdqo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (dqo-map_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libdqo_oracle.so")
_LIB_PATH_OMP = os.path.join(_HERE, "_build", "libdqo_oracle_omp.so")
_lib = None
_lib_omp = None


def build(force=False):
    if force or not (os.path.exists(_LIB_PATH) and os.path.exists(_LIB_PATH_OMP)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def _load(path):
    L = ctypes.CDLL(path)
    for suf in ("f32", "f64"):
        getattr(L, f"orc_rast_forward_{suf}").restype = ctypes.c_void_p
        getattr(L, f"orc_rast_forward_gated_{suf}").restype = ctypes.c_void_p
    return L


def lib(omp=False):
    """The serial oracle (parity reference of the small tests) or, omp=True, the OpenMP build of the same sources (full-size
    GPU tests, bench.py's cpu_baseline; threads = OMP_NUM_THREADS, default all host cores)."""
    global _lib, _lib_omp
    if omp:
        if _lib_omp is None:
            build()
            _lib_omp = _load(_LIB_PATH_OMP)
        return _lib_omp
    if _lib is None:
        build()
        _lib = _load(_LIB_PATH)
    return _lib


def num_threads(omp=True):
    return int(lib(omp).orc_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class RastSettings:
    """Scalar settings, same meaning as GaussianRasterizationSettings (__init__.py:288-307)."""

    def __init__(self, W, H, tanfovx, tanfovy, cx, cy, sh_degree=3, scale_modifier=1.0, color_sigma=3.0,
                 opaque_threshold=0.6, depth_threshold=1.0, normal_threshold=0.5, T_threshold=1e-4, bg=(0, 0, 0)):
        self.W, self.H = int(W), int(H)
        self.tanfovx, self.tanfovy, self.cx, self.cy = float(tanfovx), float(tanfovy), float(cx), float(cy)
        self.sh_degree = int(sh_degree)
        self.scale_modifier, self.color_sigma = float(scale_modifier), float(color_sigma)
        self.opaque_threshold, self.depth_threshold = float(opaque_threshold), float(depth_threshold)
        self.normal_threshold, self.T_threshold = float(normal_threshold), float(T_threshold)
        self.bg = tuple(float(b) for b in bg)


class RastResult:
    pass


class OracleRasterizer:
    """Forward + backward of the oracle rasteriser for one call. dtype is np.float32 (parity) or np.float64 (FD checks)."""

    def __init__(self, dtype=np.float32, omp=False):
        self.dt = np.dtype(dtype)
        self.suf = "f32" if self.dt == np.float32 else "f64"
        self.h = None
        self.lib = lib(omp)

    def __del__(self):
        try:
            self.free()
        except Exception:  # interpreter shutdown: ctypes may already be gone
            pass

    def free(self):
        if self.h is not None:
            getattr(self.lib, f"orc_rast_ctx_free_{self.suf}")(ctypes.c_void_p(self.h))
            self.h = None

    def forward(self, st, means3D, opacities, view, proj, campos, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, tile_mask=None, pair_masks=False, gaussian_object=None, pixel_object=None):
        """gaussian_object [P] / pixel_object [H, W] (int32, both or neither): the object gate of the sharded job's per-object
        render — an entry acts on a pixel only if the ids agree (negative pixel id: nothing acts); default off = the reference."""
        self.free()
        dt = self.dt
        c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dt)
        means3D, opacities, view, proj, campos = c(means3D), c(opacities), c(view), c(proj), c(campos)
        shs, colors_precomp, scales, rotations, cov3D_precomp = c(shs), c(colors_precomp), c(scales), c(rotations), c(cov3D_precomp)
        P = means3D.shape[0]
        M = 0 if shs is None else shs.shape[1]
        W, H = st.W, st.H
        gx, gy = (W + 15) // 16, (H + 15) // 16
        if tile_mask is None:
            tile_mask = np.ones((gy, gx), np.int32)
        tile_mask = np.ascontiguousarray(tile_mask, np.int32)
        assert tile_mask.size == gx * gy
        if scales is None or rotations is None:
            raise ValueError("the depth rasteriser dereferences scales/rotations in its blend kernel (forward.cu:780)")
        ip = np.array([P, st.sh_degree, M, W, H, 1 if pair_masks else 0], np.int32)
        fp = np.array([st.tanfovx, st.tanfovy, st.cx, st.cy, st.scale_modifier, st.color_sigma, st.opaque_threshold,
                       st.depth_threshold, st.normal_threshold, st.T_threshold], np.float64)
        bg = np.array(st.bg, dt)
        r = RastResult()
        r.color = np.empty((3, H, W), dt)
        r.depth = np.empty((1, H, W), dt)
        r.hit_color = np.empty((1, H, W), np.int32)
        r.hit_depth = np.empty((1, H, W), np.int32)
        r.hit_color_weight = np.empty((1, H, W), dt)
        r.hit_depth_weight = np.empty((1, H, W), dt)
        r.T_map = np.empty((1, H, W), dt)
        r.n_touched = np.empty((P,), np.int32)
        r.radii = np.empty((P,), np.int32)
        self._keep = (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, tile_mask)
        self.P, self.M, self.W, self.H, self.gx, self.gy = P, M, W, H, gx, gy
        args = (_p(ip), _p(fp), _p(bg), _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales), _p(rotations),
                _p(cov3D_precomp), _p(view), _p(proj), _p(campos), _p(tile_mask), _p(r.color), _p(r.depth), _p(r.hit_color),
                _p(r.hit_depth), _p(r.hit_color_weight), _p(r.hit_depth_weight), _p(r.T_map), _p(r.n_touched), _p(r.radii))
        if (gaussian_object is None) != (pixel_object is None):
            raise ValueError("object gate: gaussian_object and pixel_object go together")
        if gaussian_object is None:
            self.h = getattr(self.lib, f"orc_rast_forward_{self.suf}")(*args)
        else:
            go = np.ascontiguousarray(gaussian_object, np.int32).reshape(-1)
            po = np.ascontiguousarray(pixel_object, np.int32).reshape(-1)
            assert go.size == P and po.size == W * H
            self._keep = self._keep + (go, po)
            self.h = getattr(self.lib, f"orc_rast_forward_gated_{self.suf}")(*args, _p(go), _p(po))
        info = np.zeros(4, np.int32)
        getattr(self.lib, f"orc_rast_ctx_info_{self.suf}")(ctypes.c_void_p(self.h), _p(info))
        r.num_rendered, r.num_tiles = int(info[0]), int(info[1])
        self.N, self.num_tiles = r.num_rendered, r.num_tiles
        return r

    _CTX = {"point_list": (0, np.uint32), "ranges": (1, np.uint32), "tile_indices": (2, np.int32), "means2D": (3, None),
            "depths": (4, None), "conic_opacity": (5, None), "rgb": (6, None), "cov3D": (7, None),
            "tiles_touched": (8, np.uint32), "final_T": (9, None), "n_contrib": (10, np.uint32), "hit_normal_c": (11, None),
            "hit_point_c": (12, None), "clamped": (13, np.uint8), "point_tile": (14, np.uint32), "weight_sum": (15, None),
            "n_blend": (16, np.uint32), "pair_mask": (17, np.uint64)}

    def ctx(self, name):
        which, dt = self._CTX[name]
        dt = self.dt if dt is None else dt
        P, HW, T = self.P, self.W * self.H, self.gx * self.gy
        shape = {"point_list": (self.N,), "ranges": (T, 2), "tile_indices": (self.num_tiles,), "means2D": (P, 2),
                 "depths": (P,), "conic_opacity": (P, 4), "rgb": (P, 3), "cov3D": (P, 6), "tiles_touched": (P,),
                 "final_T": (self.H, self.W), "n_contrib": (self.H, self.W), "hit_normal_c": (self.H, self.W, 3),
                 "hit_point_c": (self.H, self.W, 3), "clamped": (P, 3), "point_tile": (self.N,),
                 "weight_sum": (self.H, self.W), "n_blend": (self.H, self.W), "pair_mask": (self.N, 4)}[name]
        out = np.zeros(shape, dt)
        getattr(self.lib, f"orc_rast_ctx_copy_{self.suf}")(ctypes.c_void_p(self.h), which, _p(out))
        return out

    def backward(self, dL_dcolor, dL_ddepth):
        dt = self.dt
        P, M = self.P, self.M
        dL_dcolor = np.ascontiguousarray(dL_dcolor, dt).reshape(3, self.H, self.W)
        dL_ddepth = np.ascontiguousarray(dL_ddepth, dt).reshape(self.H, self.W)
        g = RastResult()
        g.means3D = np.empty((P, 3), dt)
        g.sh = np.empty((P, M, 3), dt)
        g.colors = np.empty((P, 3), dt)
        g.opacity = np.empty((P, 1), dt)
        g.scales = np.empty((P, 3), dt)
        g.rotations = np.empty((P, 4), dt)
        g.cov3D = np.empty((P, 6), dt)
        g.means2D = np.empty((P, 2), dt)
        g.conic = np.empty((P, 3), dt)
        getattr(self.lib, f"orc_rast_backward_{self.suf}")(
            ctypes.c_void_p(self.h), _p(dL_dcolor), _p(dL_ddepth), _p(g.means3D), _p(g.sh) if M > 0 else None, _p(g.colors),
            _p(g.opacity), _p(g.scales), _p(g.rotations), _p(g.cov3D), _p(g.means2D), _p(g.conic))
        return g


def mark_visible(means3D, view, proj):
    means3D = np.ascontiguousarray(means3D, np.float32)
    view = np.ascontiguousarray(view, np.float32)
    proj = np.ascontiguousarray(proj, np.float32)
    out = np.zeros(means3D.shape[0], np.uint8)
    lib().orc_mark_visible_f32(means3D.shape[0], _p(means3D), _p(view), _p(proj), _p(out))
    return out.astype(bool)


def knn3(xyz, return_morton=False):
    xyz = np.ascontiguousarray(xyz, np.float32)
    P = xyz.shape[0]
    mean_d2 = np.zeros(P, np.float32)
    idx3 = np.zeros((P, 3), np.int32)
    morton = np.zeros(P, np.uint32)
    order = np.zeros(P, np.uint32)
    lib().orc_knn3(P, _p(xyz), _p(mean_d2), _p(idx3), _p(morton), _p(order))
    if return_morton:
        return mean_d2, idx3, morton, order
    return mean_d2, idx3


def quadric_iou_fwd_bwd(axes, R, center, P34, obs, dtype=np.float32):
    dt = np.dtype(dtype)
    suf = "f32" if dt == np.float32 else "f64"
    c = lambda a, s: np.ascontiguousarray(np.asarray(a, dt).reshape(s))
    axes = c(axes, (-1, 3))
    B = axes.shape[0]
    R, center, P34, obs = c(R, (B, 3, 3)), c(center, (B, 3)), c(P34, (B, 3, 4)), c(obs, (B, 4))
    bbox, loss, valid = np.zeros((B, 4), dt), np.zeros(B, dt), np.zeros(B, np.int32)
    g_axes, g_R, g_center = np.zeros((B, 3), dt), np.zeros((B, 3, 3), dt), np.zeros((B, 3), dt)
    getattr(lib(), f"orc_quadric_iou_fwd_bwd_{suf}")(B, _p(axes), _p(R), _p(center), _p(P34), _p(obs), _p(bbox), _p(loss),
                                                    _p(valid), _p(g_axes), _p(g_R), _p(g_center))
    return dict(bbox=bbox, loss=loss, valid=valid, g_axes=g_axes, g_R=g_R, g_center=g_center)


def quadric_adam(axes, R, center, P34_views, obs_views, view_schedule):
    f = np.float32
    axes = np.array(axes, f).reshape(3).copy()
    R = np.array(R, f).reshape(3, 3).copy()
    center = np.array(center, f).reshape(3).copy()
    P34_views = np.ascontiguousarray(np.asarray(P34_views, f).reshape(-1, 3, 4))
    obs_views = np.ascontiguousarray(np.asarray(obs_views, f).reshape(-1, 4))
    sched = np.ascontiguousarray(view_schedule, np.int32)
    hist = np.zeros(len(sched), f)
    lib().orc_quadric_adam_f32(P34_views.shape[0], _p(P34_views), _p(obs_views), len(sched), _p(sched), _p(axes), _p(R),
                               _p(center), _p(hist))
    return axes, R, center, hist
