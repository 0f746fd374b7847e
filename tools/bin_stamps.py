"""Per-phase timestamps of bin_count_kernel's blocks (a library built with profiles/r06_bin_stamps.patch + EXTRA=-DBIN_STAMPS).
python tools/bin_stamps.py [cfg]"""
import argparse, ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import numpy as np
import torch
import bench
import _dqo_native as N
from dqo_harness.fused_mapping import FusedMapper

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
args = argparse.Namespace(cfg=cfg, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
mask = prob["render_mask"].to(torch.uint8).contiguous()
fm = FusedMapper(prob["scene"], prob["settings"], dev)
if prob.get("gate") is not None:
    fm.set_object_gate(prob["gate"][0], prob["gate"][1])
fm.capture(prob["gt_color"], prob["gt_depth"], mask, tile_mask=prob["tile_mask"])
for _ in range(50):
    fm.replay()
torch.cuda.synchronize()
lib = N.lib()
buf = np.zeros((4096, 16), np.uint64)
lib.dqo_debug_bin_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.dqo_debug_bin_stamps(buf.ctypes.data, buf.nbytes) == 0
nb = (fm.P + 255) // 256
st = buf[:nb].astype(np.int64)
st = st[st[:, 6] > 0]
t0 = st[:, 0].min()
us = lambda x: x * 0.01
names = ["start", "K1: loads + per-Gaussian early part", "LDS park + scan 1", "sweep 1: footprint tests + counts", "scan 2 + slot allocation (returning atomic)",
         "tiles_touched / rec_valid stores", "sweep 2: ranks (returning atomics) + records"]
print(f"cfg {cfg}: {len(st)} blocks; kernel span {us(st[:, 6].max() - t0):.1f} us; block starts: median {us(np.median(st[:, 0]) - t0):.1f}, max {us(st[:, 0].max() - t0):.1f} us")
life = us(st[:, 6] - st[:, 0])
print(f"block lifetime: mean {life.mean():.1f} us, median {np.median(life):.1f}, p90 {np.percentile(life, 90):.1f}, max {life.max():.1f}")
for i in range(1, 7):
    d = us(st[:, i] - st[:, i - 1])
    print(f"  phase {i} ({names[i]:>46s}): mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}")
tot = st[:, 8] & 0xffffffff; live = st[:, 8] >> 32
print(f"candidates per block: mean {tot.mean():.0f}, p90 {np.percentile(tot, 90):.0f}, max {tot.max()}; kept: mean {live.mean():.0f}, max {live.max()}")
