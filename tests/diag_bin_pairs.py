"""Diagnostic (under tests/: it runs the CPU oracle): of the (Gaussian, tile) instances of a frame, how many sit in a horizontally adjacent,
even-aligned tile pair of the SAME Gaussian — the ones one 64-bit atomic on a pair of tile counters could rank together
(third lead).   python tests/diag_bin_pairs.py [cfg]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/dqo-map_amd', R + '/tests']
from dqo_harness import scenes
from oracle import oracle_lib as ol
import util_rast as U

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cam, sc = scenes.make_config(cfg)
o = ol.OracleRasterizer(np.float32, omp=True)
st = U.oracle_settings(ol, cam)
o.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, shs=sc["shs"],
          scales=sc["scales"], rotations=sc["rotations"], pair_masks=True)
gid, m = o.ctx("point_list").astype(np.int64), o.ctx("pair_mask")
ranges = o.ctx("ranges")
tile = np.repeat(np.arange(ranges.shape[0]), (ranges[:, 1] - ranges[:, 0]).astype(np.int64))
gx = (cam.W + 15) // 16
for name, keep in (("all instances of the reference's rects", np.ones(len(gid), bool)), ("instances with arithmetic (~ the footprint cull)", m.any(1))):
    g, t = gid[keep], tile[keep]
    x = t % gx
    key = g * (1 << 20) + t
    s = set(key.tolist())
    even = x % 2 == 0
    has_right = np.fromiter(((k + 1) in s for k in key[even].tolist()), bool, int(even.sum())) & (x[even] + 1 < gx)
    pairs = int(has_right.sum())
    n = len(g)
    atom = n - pairs
    print(f"cfg {cfg}, {name}: {n} instances, {pairs} aligned pairs = {2 * pairs / n:.1%} of the instances -> {atom} atomics "
          f"({atom / n:.1%}); at 1.12x the cost for a 64-bit one: {(n - 2 * pairs + 1.12 * pairs) / n:.1%} of today's atomic time")
