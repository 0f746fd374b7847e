"""Per-phase timestamps of gaussian_tail_kernel's blocks (a library built with EXTRA=-DTAIL_STAMPS from profiles/r06_tail_stamps.patch):
wall_clock64 (100 MHz) at phase boundaries of every block, read back after one replayed iteration.   python tools/tail_stamps.py [cfg]"""
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
buf = np.zeros((8192, 16), np.uint64)
lib.dqo_debug_tail_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.dqo_debug_tail_stamps(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
nb = (fm.P + 127) // 128
st = buf[:nb].astype(np.int64)
ok = st[:, 9] > 0
st = st[ok]
t0 = st[:, 0].min()
us = lambda x: x * 0.01  # 100 MHz ticks -> us
names = ["start", "A lists", "chain inputs issued", "B record sums", "C chain", "barrier", "small xyz pass", "SH pass", "rows pass", "ticket"]
print(f"cfg {cfg}: {len(st)} blocks; kernel span {us(st[:, 9].max() - t0):.1f} us (first start .. last end)")
life = us(st[:, 9] - st[:, 0])
print(f"block lifetime: mean {life.mean():.1f} us, p10 {np.percentile(life, 10):.1f}, median {np.median(life):.1f}, p90 {np.percentile(life, 90):.1f}, max {life.max():.1f}")
for i in range(1, 10):
    d = us(st[:, i] - st[:, i - 1])
    print(f"  phase {i} ({names[i]:>20s}): mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}")
if st[:, 13].max() > 0:
    one = (st[:, 12] & 0xffffffff) <= 128
    for nm, a_, b_ in (("B trip 1: words + records + sums", 2, 13), ("barrier", 13, 14), ("per-Gaussian sums + barrier", 14, 15), ("rest of B (second trip)", 15, 3)):
        d = us(st[:, b_] - st[:, a_])
        print(f"    B: {nm:>34s}: mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}   one-trip blocks {d[one].mean():6.2f}  two-trip blocks {d[~one].mean():6.2f}")
start = us(st[:, 0] - t0)
print("block start times: p10 %.1f  median %.1f  p90 %.1f  max %.1f us" % (np.percentile(start, 10), np.median(start), np.percentile(start, 90), start.max()))
h, edges = np.histogram(start, bins=12)
print("start histogram (us):", [f"{edges[i]:.0f}-{edges[i+1]:.0f}: {h[i]}" for i in range(len(h))])
slots = (st[:, 12] & 0xffffffff); rows = st[:, 12] >> 32
print(f"slots per block: mean {slots.mean():.0f}, p90 {np.percentile(slots, 90):.0f}, max {slots.max()}; > 128: {(slots > 128).mean():.2f}; list rows per block: mean {rows.mean():.1f}, p90 {np.percentile(rows, 90):.0f}")
# wave 1 vs wave 0 in D
w1 = us(st[:, 10] - st[:, 5]); w0 = us(st[:, 8] - st[:, 5])
print(f"phase D by wave: wave 0 {w0.mean():.2f} us, wave 1 {w1.mean():.2f} us")
