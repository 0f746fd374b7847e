"""Diagnostic: the smoke case's gradient rows beyond the bar — HIP vs fp32 oracle vs fp64 oracle, per tensor (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dqo-map_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from dqo_harness import scenes
from oracle import oracle_lib as ol
import util_rast as U

cfg, P = int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cam, sc = scenes.make_config(cfg, P=P)
rng = np.random.default_rng(0 if cfg == 1 else 11)
dL = (rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32))
hr = U.HipRun(cam, sc)
o, r, _ = U.run_oracle(ol, cam, sc, omp=True)
o64, r64, _ = U.run_oracle(ol, cam, sc, dtype=np.float64, omp=True)
bad = U.flipped_pixels(hr.res, r, r64)
keep = (~bad).astype(np.float32)
dLm = (dL[0] * keep[None], dL[1] * keep[None])
hg = hr.backward(dLm, retain=False)
og, og64 = U.oracle_backward(o, dLm), U.oracle_backward(o64, dLm)
for k in ("scales", "rotations", "means3D"):
    e, eo, eh = U._row_err(hg[k], og[k]), U._row_err(og[k], og64[k]), U._row_err(hg[k], og64[k])
    worst = np.argsort(-eh)[:6]
    print(k, "rows>1e-3 vs fp32:", int((e > 1e-3).sum()), "hip_vs_64 max", float(eh.max()), "orc_vs_64 max", float(eo.max()),
          "ratio of sums", float(eh.sum() / eo.sum()))
    for i in worst:
        print("   row", int(i), "hip_vs32 %.2e  orc32_vs64 %.2e  hip_vs64 %.2e" % (e[i], eo[i], eh[i]), "radius", int(hr.res["radii"][i]),
              "scales", sc["scales"][i])
