"""GPU time of every graph replay after an idle period (HIP events around each launch): where does a 20-iteration timed region lose its
0.4 ms against a 200-iteration one?   python tools/replay_times.py"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench
from dqo_harness.fused_mapping import FusedMapper

args = argparse.Namespace(cfg=3, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
fm = FusedMapper(prob["scene"], prob["settings"], dev).set_object_gate(prob["gate"][0], prob["gate"][1])
fm.capture(prob["gt_color"], prob["gt_depth"], prob["render_mask"].to(torch.uint8).contiguous(), tile_mask=prob["tile_mask"], list_split="auto", unroll=4)
for idle_ms in (0, 5, 50, 500):
    for _ in range(5):
        fm.replay()
    torch.cuda.synchronize()
    time.sleep(idle_ms / 1e3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(8):
        fm.replay()
        ev[i + 1].record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per = [round(ev[i].elapsed_time(ev[i + 1]) / 4, 4) for i in range(8)]
    print(f"idle {idle_ms:4d} ms before: ms per iteration of replays 1..8: {per}; host issue {1e3 * (t1 - t0):.3f} ms, wall to drain {1e3 * (t2 - t0):.3f} ms"
          f" = {1e3 * (t2 - t0) / 32:.4f} ms per iteration")
