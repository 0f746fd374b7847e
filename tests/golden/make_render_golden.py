"""Generates tests/golden/render_golden.npz: `(args -> 9-tuple)` fixtures of the rasteriser at 64x48, computed
by the fp32 CPU oracle (oracle/dqo_oracle_rast.cpp — the restatement of forward.cu / backward.cu; the CUDA reference itself cannot
run here), for the harness' `render()` to be checked against without the oracle in the loop, and as a regression pin of the oracle.

Two cases: an all-tiles render with per-Gaussian normals (normal gather of SLAM/render.py:208-212) and a tile-masked render
(masked tiles keep the reference's initial fills: ids 0, quirk B7, so the normal gather aliases Gaussian 0 there).

Run:  python tests/golden/make_render_golden.py
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

NAMES = ("color", "depth", "hit_color", "hit_depth", "hit_color_weight", "hit_depth_weight", "T_map", "n_touched", "radii")


def scene_64x48(P=900, seed=17):
    cam = scenes.Camera(64, 48, 60.0, 60.0, 31.5, 23.5, scenes.rot_yx(5.0, -2.0), np.array([0.02, -0.01, 0.05]))
    sc = scenes.frustum_cloud(seed, P, cam, zmin=0.5, zmax=1.6, rest_sigma=0.05)
    return cam, sc


def main():
    cam, sc = scene_64x48()
    out = dict(xyz=sc["xyz"], scales=sc["scales"], rotations=sc["rotations"], opacity=sc["opacity"], shs=sc["shs"], normals=sc["normals"])
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    mask = np.ones((gy, gx), np.int32)
    mask[0, 1] = 0
    mask[2, 3] = 0
    out["tile_mask"] = mask
    rng = np.random.default_rng(5)
    dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))
    out["dL_dcolor"], out["dL_ddepth"] = dL
    for tag, tm in (("full", None), ("masked", mask)):
        o, r, g = U.run_oracle(ol, cam, sc, tile_mask=tm, dL=dL, bg=(0.05, 0.1, 0.15))
        for k in NAMES:
            out[f"{tag}_{k}"] = r[k]
        for k, v in g.items():
            out[f"{tag}_grad_{k}"] = v
        idx = r["hit_depth"]
        rn = np.zeros_like(r["color"])
        rn[:, idx[0] > -1] = sc["normals"][idx[idx > -1]].T  # SLAM/render.py:208-212
        out[f"{tag}_normal"] = rn
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; hit pixels", int((out["full_hit_depth"] >= 0).sum()))


if __name__ == "__main__":
    main()
