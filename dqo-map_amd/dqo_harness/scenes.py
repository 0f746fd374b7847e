"""Seeded synthetic scenes + camera matrices for tests and bench.py (numpy only, no torch, no GPU).

The reference datasets are not in the tree; these generators follow SURVEY.md §8(d):
attribute ranges from /root/reference/configs/base.yaml:30-36 (min_radius 0.001, max_radius 0.05, xyz_factor
[1,1,0.1]), opacity 0.99 (SLAM/multiprocess/mapper.py:1384), surfel orientation = local z along the surface
normal (SLAM/utils.py:246 compute_rot).  Camera matrices restate scene/cameras.py:138-155 and
utils/graphics_utils.py:52-86 (row-vector convention: the 4x4s handed to the op are the transposes).
"""
import math

import numpy as np

SH_C0 = 0.28209479177387814


def rgb_to_sh(rgb):
    # utils/sh_utils.py:123-124
    return (rgb - 0.5) / SH_C0


class Camera:
    """Pinhole camera; `Rw2c`, `t` map world -> camera (x_c = Rw2c x_w + t)."""

    def __init__(self, W, H, fx, fy, cx, cy, Rw2c=None, t=None, znear=0.01, zfar=100.0):
        self.W, self.H, self.fx, self.fy, self.cx, self.cy = int(W), int(H), float(fx), float(fy), float(cx), float(cy)
        Rw2c = np.eye(3) if Rw2c is None else np.asarray(Rw2c, np.float64)
        t = np.zeros(3) if t is None else np.asarray(t, np.float64)
        self.Rw2c, self.t = Rw2c, t
        # utils/graphics_utils.py:98-100 focal2fov
        self.FoVx = 2 * math.atan(W / (2 * fx))
        self.FoVy = 2 * math.atan(H / (2 * fy))
        self.tanfovx = math.tan(self.FoVx * 0.5)
        self.tanfovy = math.tan(self.FoVy * 0.5)
        # getWorld2View2 (graphics_utils.py:52-63) with translate=0, scale=1, then .transpose(0,1) (cameras.py:138-140)
        Rt = np.zeros((4, 4))
        Rt[:3, :3] = Rw2c
        Rt[:3, 3] = t
        Rt[3, 3] = 1.0
        self.Rt = Rt
        self.world_view_transform = np.float32(Rt).T.copy()
        # getProjectionMatrix (graphics_utils.py:66-86), built in float32 like torch.zeros(4,4)
        tanY, tanX = math.tan(self.FoVy / 2), math.tan(self.FoVx / 2)
        top, right = tanY * znear, tanX * znear
        Pm = np.zeros((4, 4), np.float32)
        Pm[0, 0] = 2.0 * znear / (2 * right)
        Pm[1, 1] = 2.0 * znear / (2 * top)
        Pm[3, 2] = 1.0
        Pm[2, 2] = zfar / (zfar - znear)
        Pm[2, 3] = -(zfar * znear) / (zfar - znear)
        self.projection_matrix = Pm.T.copy()
        # cameras.py:149-155
        self.full_proj_transform = (self.world_view_transform @ self.projection_matrix).astype(np.float32)
        self.camera_center = np.linalg.inv(self.world_view_transform.astype(np.float64))[3, :3].astype(np.float32)
        self.K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)

    def P34(self):
        """K @ [R|t], the 3x4 projection used by the quadric residual (quadrics.py:2269)."""
        return self.K @ self.Rt[:3, :]


def quat_from_two_vectors(a, b):
    """Unit quaternion (r,x,y,z) rotating unit vector a onto b (batched)."""
    a = a / np.linalg.norm(a, axis=-1, keepdims=True)
    b = b / np.linalg.norm(b, axis=-1, keepdims=True)
    d = np.sum(a * b, -1, keepdims=True)
    axis = np.cross(a, b)
    q = np.concatenate([1.0 + d, axis], -1)
    # antiparallel: pick any orthogonal axis
    bad = (1.0 + d[..., 0]) < 1e-8
    if np.any(bad):
        alt = np.cross(a[bad], np.array([1.0, 0, 0]))
        small = np.linalg.norm(alt, axis=-1) < 1e-6
        alt[small] = np.cross(a[bad][small], np.array([0, 1.0, 0]))
        q[bad] = np.concatenate([np.zeros((alt.shape[0], 1)), alt], -1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def quat_mul(q1, q2):
    r1, x1, y1, z1 = np.moveaxis(q1, -1, 0)
    r2, x2, y2, z2 = np.moveaxis(q2, -1, 0)
    return np.stack([r1 * r2 - x1 * x2 - y1 * y2 - z1 * z2, r1 * x2 + x1 * r2 + y1 * z2 - z1 * y2,
                     r1 * y2 - x1 * z2 + y1 * r2 + z1 * x2, r1 * z2 + x1 * y2 - y1 * x2 + z1 * r2], -1)


def rot_yx(yaw_deg, pitch_deg):
    y, p = math.radians(yaw_deg), math.radians(pitch_deg)
    Ry = np.array([[math.cos(y), 0, math.sin(y)], [0, 1, 0], [-math.sin(y), 0, math.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(p), -math.sin(p)], [0, math.sin(p), math.cos(p)]])
    return Rx @ Ry


def _surfel_attrs(rng, normals, sh_degree, rest_sigma):
    P = normals.shape[0]
    lo, hi = math.log(0.001), math.log(0.05)
    s2 = np.exp(rng.uniform(lo, hi, (P, 2)))
    scales = np.concatenate([s2, 0.1 * s2.min(1, keepdims=True)], 1)  # xyz_factor [1,1,0.1]: thin along local z
    q_align = quat_from_two_vectors(np.tile(np.array([0, 0, 1.0]), (P, 1)), normals)
    jit = np.concatenate([np.ones((P, 1)), rng.normal(0, 0.03, (P, 3))], 1)
    jit /= np.linalg.norm(jit, axis=1, keepdims=True)
    spin = rng.uniform(0, 2 * math.pi, P)
    q_spin = np.stack([np.cos(spin / 2), np.zeros(P), np.zeros(P), np.sin(spin / 2)], 1)
    rot = quat_mul(quat_mul(q_align, jit), q_spin)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opac = np.where(rng.uniform(size=(P, 1)) < 0.9, 0.99, 0.1)
    M = (sh_degree + 1) ** 2
    shs = np.zeros((P, M, 3))
    shs[:, 0, :] = rgb_to_sh(rng.uniform(0, 1, (P, 3)))
    if rest_sigma > 0 and M > 1:
        shs[:, 1:, :] = rng.normal(0, rest_sigma, (P, M - 1, 3))
    f = np.float32
    return scales.astype(f), rot.astype(f), opac.astype(f), shs.astype(f)


def object_layout(n_objects):
    """(wall strips per face, first patch id, number of patch ids).  Object ids of the synthetic room (obj_id only drives the
    per-object masks / shards of bench.py, never the rasteriser): fewer than 8 ids = the walls are object 0 and the patches share
    ids 1..n-1; from 8 ids on three quarters of the ids go to the walls, like three quarters of the Gaussians — each of the 6 faces
    is cut into n // 8 strips (8 ids: 6 faces + 2 patch ids; 16: 12 + 4; 32: 24 + 8), so no single object holds most of the map."""
    if n_objects < 8:
        return 0, 1, max(1, n_objects - 1)
    k = n_objects // 8
    return k, 6 * k, n_objects - 6 * k


def surfel_room(seed, P, n_objects=1, sh_degree=3, rest_sigma=0.0, n_patches=20):
    """Surfel room: points on the 6 faces of a 6 x 3 x 4 m box plus `n_patches` random planar patches (the 'objects').
    Returns dict(xyz, scales, rotations, opacity, shs, obj_id, normals)."""
    rng = np.random.default_rng(seed)
    strips, patch0, n_patch_ids = object_layout(n_objects)
    half = np.array([3.0, 1.5, 2.0])
    n_patch_pts = P // 4 if n_patches > 0 else 0
    n_wall = P - n_patch_pts
    # faces weighted by area
    faces = []
    for ax in range(3):
        o = [a for a in range(3) if a != ax]
        area = 4 * half[o[0]] * half[o[1]]
        for sgn in (-1, 1):
            faces.append((ax, sgn, area))
    areas = np.array([f[2] for f in faces])
    counts = rng.multinomial(n_wall, areas / areas.sum())
    xyz, nrm, obj = [], [], []
    for fi, ((ax, sgn, _), n) in enumerate(zip(faces, counts)):
        p = rng.uniform(-1, 1, (n, 3)) * half
        p[:, ax] = sgn * half[ax]
        nn = np.zeros((n, 3))
        nn[:, ax] = -sgn  # facing inward
        xyz.append(p)
        nrm.append(nn)
        if strips == 0:
            obj.append(np.zeros(n, np.int32))
        else:  # strips along the face's longer in-plane axis
            o = [a for a in range(3) if a != ax]
            la = o[0] if half[o[0]] >= half[o[1]] else o[1]
            cell = np.clip(((p[:, la] / half[la] + 1.0) * 0.5 * strips).astype(np.int32), 0, strips - 1)
            obj.append((fi * strips + cell).astype(np.int32))
    if n_patches > 0:
        per = rng.multinomial(n_patch_pts, np.ones(n_patches) / n_patches)
        for k, n in enumerate(per):
            ctr = rng.uniform(-0.7, 0.7, 3) * half
            nn = rng.normal(size=3)
            nn /= np.linalg.norm(nn)
            u = np.cross(nn, [0.3, 0.5, 0.81])
            u /= np.linalg.norm(u)
            v = np.cross(nn, u)
            ext = rng.uniform(0.25, 0.7, 2)
            ab = rng.uniform(-1, 1, (n, 2)) * ext
            p = ctr + ab[:, :1] * u + ab[:, 1:] * v
            xyz.append(p)
            nrm.append(np.tile(nn, (n, 1)))
            obj.append(np.full(n, patch0 + (k % n_patch_ids) if n_objects > 1 else 0, np.int32))
    xyz = np.concatenate(xyz)
    nrm = np.concatenate(nrm)
    obj = np.concatenate(obj)
    perm = rng.permutation(xyz.shape[0])
    xyz, nrm, obj = xyz[perm], nrm[perm], obj[perm]
    scales, rot, opac, shs = _surfel_attrs(rng, nrm, sh_degree, rest_sigma)
    return dict(xyz=xyz.astype(np.float32), scales=scales, rotations=rot, opacity=opac, shs=shs, obj_id=obj,
                normals=nrm.astype(np.float32))


def frustum_cloud(seed, P, cam, sh_degree=3, zmin=0.5, zmax=5.0, rest_sigma=0.0):
    """cfg 1: uniform random Gaussians inside the camera frustum, z in [zmin, zmax]."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(zmin, zmax, P)
    x = rng.uniform(-1, 1, P) * cam.tanfovx * z
    y = rng.uniform(-1, 1, P) * cam.tanfovy * z
    pc = np.stack([x, y, z], 1)
    xyz = (pc - cam.t) @ cam.Rw2c  # Rw2c^T (pc - t)
    nrm = rng.normal(size=(P, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    scales, rot, opac, shs = _surfel_attrs(rng, nrm, sh_degree, rest_sigma)
    return dict(xyz=xyz.astype(np.float32), scales=scales, rotations=rot, opacity=opac, shs=shs,
                obj_id=np.zeros(P, np.int32), normals=nrm.astype(np.float32))


def replica_camera(W=1200, H=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5, yaw=12.0, pitch=4.0, pos=(0.3, 0.1, -1.85)):
    """Camera inside the surfel room near the back wall, looking roughly along +z."""
    Rw2c = rot_yx(yaw, pitch)
    t = -Rw2c @ np.asarray(pos, np.float64)
    return Camera(W, H, fx, fy, cx, cy, Rw2c, t)


CONFIGS = {
    # cfg -> (seed, P, W, H, fx, cx, cy, n_objects, rest_sigma)
    1: dict(seed=1, P=10_000, W=640, H=480, fx=525.0, cx=319.5, cy=239.5, n_objects=1, rest_sigma=0.0),
    2: dict(seed=2, P=100_000, W=1200, H=680, fx=600.0, cx=599.5, cy=339.5, n_objects=1, rest_sigma=0.0),
    3: dict(seed=3, P=500_000, W=1200, H=680, fx=600.0, cx=599.5, cy=339.5, n_objects=8, rest_sigma=0.0),
    4: dict(seed=4, P=1_000_000, W=1200, H=680, fx=600.0, cx=599.5, cy=339.5, n_objects=16, rest_sigma=0.0),
    5: dict(seed=5, P=2_000_000, W=1200, H=680, fx=600.0, cx=599.5, cy=339.5, n_objects=32, rest_sigma=0.05),
}


def make_config(cfg, P=None):
    c = dict(CONFIGS[cfg])
    if P is not None:
        c["P"] = P
    if cfg == 1:
        cam = Camera(c["W"], c["H"], c["fx"], c["fx"], c["cx"], c["cy"], rot_yx(7.0, -3.0), np.array([0.05, -0.02, 0.1]))
        scene = frustum_cloud(c["seed"], c["P"], cam)
    else:
        cam = replica_camera(c["W"], c["H"], c["fx"], c["fx"], c["cx"], c["cy"])
        scene = surfel_room(c["seed"], c["P"], n_objects=c["n_objects"], rest_sigma=c["rest_sigma"])
    return cam, scene
