"""Per-tile list lengths of a BASELINE config on the GPU (drop-in op, exact mode): the distribution the per-tile sort and the blend
loops' critical paths depend on.      python tools/list_lengths.py [cfg] [P]"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
from dqo_harness import scenes, mapping
import diff_gaussian_rasterization_depth as dgr

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
cam, sc = scenes.make_config(cfg, P=P)
dev = torch.device("cuda")
st = mapping.make_settings(cam, dev)
p = mapping.GaussianParams(sc, dev).activated()


class Ctx:
    def save_for_backward(self, *a):
        self.saved = a

    def mark_non_differentiable(self, *a):
        pass


c = Ctx()
e = torch.Tensor([])
with torch.no_grad():
    dgr._RasterizeGaussians.forward(c, p["xyz"], p["shs"], e, p["opacity"], p["scales"], p["rotations"], e, None, st)
img = c.saved[10]
T = ((cam.W + 15) // 16) * ((cam.H + 15) // 16)
al = lambda n: (n + 255) // 256 * 256
off = 2 * al(4 * T * 64) + al(4 * T)  # image layout: tile_count | tile_flag | tile_cursor | ranges (dqo_common.h)
rg = img[off:off + 8 * T].view(torch.int32).cpu().numpy().reshape(T, 2).astype(np.int64)
n = rg[:, 1] - rg[:, 0]
print(f"cfg {cfg}: {T} tiles, {int(n.sum())} instances, mean {n.mean():.0f}, median {np.median(n):.0f}, p90 {np.quantile(n, .9):.0f}, "
      f"p99 {np.quantile(n, .99):.0f}, max {n.max()}")
for lo, hi in ((0, 64), (65, 256), (257, 512), (513, 1024), (1025, 2048), (2049, 4096), (4097, 8192), (8193, 1 << 30)):
    m = (n >= lo) & (n <= hi)
    print(f"  {lo:5d}..{hi if hi < 1 << 29 else 'inf':>5}: {int(m.sum()):5d} tiles, {int(n[m].sum()):9d} entries")
# how far the blend actually walks the long lists (walk4: per quadrant, the last position any of its pixels used)
w4 = img[off + al(8 * T):off + al(8 * T) + 16 * T].view(torch.int32).cpu().numpy().reshape(T, 4).astype(np.int64)
for lo in (512, 1024, 2048):
    m = n > lo
    if m.any():
        fr = w4[m] / n[m, None]
        print(f"  lists > {lo}: {int(m.sum())} tiles; walked fraction per quadrant: mean {fr.mean():.2f}, median {np.median(fr):.2f}, "
              f"p10 {np.quantile(fr, .1):.2f}, max {fr.max():.2f}; mean walked entries {w4[m].mean():.0f}, max {w4[m].max()}")
