"""Diagnosis of the blank-frame losses of the 2-rank rehearsal: per-iteration state of the sharded fused path
under different forms of the per-iteration collective.  torchrun --nproc-per-node 2 tools/debug_n2.py <mode>
modes: none | async | sync | cpu (all-reduce of a host copy)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dqo-map_amd"))
import argparse  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "async"
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if world > 1:
    torch.distributed.init_process_group(os.environ.get("DQO_BENCH_BACKEND", "gloo"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
dev = torch.device("cuda", torch.cuda.current_device())
args = argparse.Namespace(cfg=3, P=int(os.environ.get("DBG_P", "100000")), view="room", scaling="strong")
import diff_gaussian_rasterization_depth as dgr  # noqa: E402
from dqo_harness.sharding import PackedAllReduce  # noqa: E402
dgr.set_sync_mode("lazy")
prob = bench.build_problem(args, rank, world, dev)
buf = PackedAllReduce(bench.LOSS_SPEC, dev)
r = bench.FusedRunner(prob, dev, buf, 1)  # world=1 inside the runner: the collective is issued here, by `mode`
fm = r.fm


def state(tag):
    torch.cuda.synchronize()
    h = fm.header()
    o = fm._g.out
    print(f"[r{rank}] {tag}: loss {[round(x, 5) for x in fm.loss[:3].tolist()]} overflow {h['overflow']} N {h['num_rendered']} "
          f"vis {h['num_visible']} hit_px {int((o[3] >= 0).sum())} color_sum {float(o[0].sum()):.1f} gt_sum {float(prob['gt_color'].sum()):.1f} "
          f"gtd_sum {float(prob['gt_depth'].sum()):.1f} mask {int(r.mask_u8.sum())} tiles {int(prob['tile_mask'].sum())} "
          f"step_dev {int(fm._g.step_dev.item())} xyz_sum {float(fm.xyz.sum()):.3f}", flush=True)


state("after capture")
for it in range(8):
    fm.replay()
    if mode == "async":
        buf.reduce_async(src=fm.loss)
    elif mode == "sync":
        buf.buf.copy_(fm.loss)
        buf.reduce()
    elif mode == "cpu":
        t = fm.loss.cpu()
        if world > 1:
            torch.distributed.all_reduce(t)
    state(f"it {it} mode {mode}")
buf.finish()
state("end")
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
