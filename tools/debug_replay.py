"""Which host-side activity between two graph replays breaks the captured iteration?  python tools/debug_replay.py <op>
ops: none | sync | header | sums | item | clone | alloc"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dqo-map_amd"))
import argparse  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402

op = sys.argv[1] if len(sys.argv) > 1 else "none"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
args = argparse.Namespace(cfg=3, P=int(os.environ.get("DBG_P", "100000")), view="room", scaling="strong")
import diff_gaussian_rasterization_depth as dgr  # noqa: E402
from dqo_harness.sharding import PackedAllReduce  # noqa: E402
dgr.set_sync_mode("lazy")
prob = bench.build_problem(args, 0, 1, dev)
r = bench.FusedRunner(prob, dev, PackedAllReduce(bench.LOSS_SPEC, dev), 1)
fm = r.fm
junk = []
pre = os.environ.get("DBG_PRE", "")
o = fm._g.out
if "sync" in pre:
    torch.cuda.synchronize()
if "header" in pre:
    fm.header()
if "loss" in pre:
    fm.loss[:3].tolist()
if "hit" in pre:
    int((o[3] >= 0).sum())
if "color" in pre:
    float(o[0].sum())
if "gt" in pre:
    float(prob["gt_color"].sum()), float(prob["gt_depth"].sum())
if "mask" in pre:
    int(r.mask_u8.sum()), int(prob["tile_mask"].sum())
if "stepdev" in pre:
    int(fm._g.step_dev.item())
if "xyz" in pre:
    float(fm.xyz.sum())
for it in range(6):
    fm.replay()
    if op == "sync":
        torch.cuda.synchronize()
    elif op == "header":
        fm.header()
    elif op == "sums":
        float(prob["gt_color"].sum())
    elif op == "item":
        int(fm._g.step_dev.item())
    elif op == "clone":
        junk.append(fm.loss.clone())
    elif op == "alloc":
        junk.append(torch.empty(1 << 20, device=dev))
torch.cuda.synchronize()
h = fm.header()
print(f"pre {pre} op {op}: loss {[round(x, 5) for x in fm.loss[:3].tolist()]} overflow {h['overflow']} N {h['num_rendered']} vis {h['num_visible']} "
      f"color_sum {float(fm._g.out[0].sum()):.1f} counters0 {int(fm._g.geom[256:260].view(torch.int32).item())}", flush=True)
