"""Generates tests/golden/render_obj_golden.npz: the `render_obj` image (SLAM/render.py:61-132 — a few objects' ellipsoids as Gaussians
with one precomputed colour per object) at 64x48, with and without a tile mask, plus the gradients of a seeded dL / dcolour, computed by
the fp32 CPU oracle (oracle/dqo_oracle_rast.cpp; the CUDA reference cannot run here).  For dqo_harness.mapping.render_obj to be checked
against without the oracle in the loop, and as a regression pin of the oracle's colors_precomp path.

Run:  python tests/golden/make_render_obj_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dqo-map_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from dqo_harness import scenes  # noqa: E402
from oracle import oracle_lib as ol  # noqa: E402
import util_rast as U  # noqa: E402

BG = (0.0, 0.0, 0.0)


def ellipsoids_64x48(n_obj=5, seed=23):
    """`n_obj` object ellipsoids in front of a 64x48 camera, each ONE Gaussian with its object's colour (what quadrics.py hands to
    render_obj: centre, semi-axes as scales, orientation, opacity 1 clamped, obj_color)."""
    cam = scenes.Camera(64, 48, 60.0, 60.0, 31.5, 23.5, scenes.rot_yx(4.0, 3.0), np.array([0.01, 0.02, -0.03]))
    rng = np.random.default_rng(seed)
    xyz = np.c_[rng.uniform(-0.6, 0.6, n_obj), rng.uniform(-0.4, 0.4, n_obj), rng.uniform(1.2, 2.2, n_obj)].astype(np.float32)
    xyz = (xyz - cam.world_view_transform[3, :3]) @ np.linalg.inv(cam.world_view_transform[:3, :3])  # camera -> world
    q = rng.normal(size=(n_obj, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    sc = dict(xyz=xyz.astype(np.float32), scales=rng.uniform(0.05, 0.25, (n_obj, 3)).astype(np.float32), rotations=q,
              opacity=np.full((n_obj, 1), 0.99, np.float32), shs=np.zeros((n_obj, 16, 3), np.float32))
    sc["obj_color"] = rng.uniform(0.1, 1.0, (n_obj, 3)).astype(np.float32)
    return cam, sc


def main():
    cam, sc = ellipsoids_64x48()
    out = {k: sc[k] for k in ("xyz", "scales", "rotations", "opacity", "obj_color")}
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    mask = np.ones((gy, gx), np.int32)
    mask[1, 2] = 0
    out["tile_mask"] = mask
    rng = np.random.default_rng(6)
    dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), np.zeros((1, cam.H, cam.W), np.float32))
    out["dL_dcolor"] = dL[0]
    for tag, tm in (("full", None), ("masked", mask)):
        _, r, g = U.run_oracle(ol, cam, sc, tile_mask=tm, colors_precomp=sc["obj_color"], dL=dL, bg=BG)
        out[f"{tag}_render_obj"] = r["color"]
        out[f"{tag}_hit_color"] = r["hit_color"]
        for k, v in g.items():
            out[f"{tag}_grad_{k}"] = v
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_obj_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; covered pixels", int((out["full_hit_color"] >= 0).sum()))


if __name__ == "__main__":
    main()
