"""Per-phase timestamps of tile_sort_wave_kernel's blocks (a library built with profiles/r06_sort_stamps.patch + EXTRA=-DSORT_STAMPS).
python tools/sort_stamps.py [cfg]"""
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
buf = np.zeros((8192, 8), np.uint64)
lib.dqo_debug_sort_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.dqo_debug_sort_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.astype(np.int64)
T = ((prob["settings"].image_width + 15) // 16) * ((prob["settings"].image_height + 15) // 16)
slots = 8 * ((T + 7) // 8)
us = lambda x: x * 0.01
sort, late = st[:slots], st[slots:]
late = late[late[:, 4] > 0]
t0 = min(sort[sort[:, 0] > 0][:, 0].min(), late[:, 0].min() if len(late) else 1 << 62)
print(f"cfg {cfg}: {slots} sort blocks, {len(late)} late-part blocks (of those below block 8192)")
w = sort[(sort[:, 4] > 0)]
print(f"sort blocks with a list: {len(w)}; start: median {us(np.median(w[:, 0]) - t0):.1f} us, max {us(w[:, 0].max() - t0):.1f}; end: median {us(np.median(w[:, 4]) - t0):.1f}, max {us(w[:, 4].max() - t0):.1f}")
for lo, hi in ((1, 64), (65, 128), (129, 256), (257, 512), (513, 1024)):
    m = w[(w[:, 7] >= lo) & (w[:, 7] <= hi)]
    if not len(m):
        continue
    ph = [us(m[:, 1] - m[:, 0]), us(m[:, 2] - m[:, 1]), us(m[:, 3] - m[:, 2]), us(m[:, 4] - m[:, 3])]
    print(f"  lists of {lo:4d}..{hi:4d}: {len(m):5d} blocks; head {ph[0].mean():5.2f} us, records loaded {ph[1].mean():5.2f}, network {ph[2].mean():5.2f}"
          f" (max {ph[2].max():5.2f}), stores {ph[3].mean():5.2f}; block lifetime mean {us(m[:, 4] - m[:, 0]).mean():5.2f}, max {us(m[:, 4] - m[:, 0]).max():5.2f};"
          f" last end at {us(m[:, 4].max() - t0):5.1f} us")
if len(late):
    print(f"late-part blocks: start median {us(np.median(late[:, 0]) - t0):.1f} us, max {us(late[:, 0].max() - t0):.1f}; lifetime mean {us(late[:, 4] - late[:, 0]).mean():.2f},"
          f" max {us(late[:, 4] - late[:, 0]).max():.2f}; last end at {us(late[:, 4].max() - t0):.1f} us")
print(f"kernel span (first stamp to last): {us(max(w[:, 4].max(), late[:, 4].max() if len(late) else 0) - t0):.1f} us")
