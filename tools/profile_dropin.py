"""Host-side profile of the drop-in path (autograd through the op + eager loss + torch.optim.Adam) on cfg 3: where does the Python /
launch time of one iteration go?      python tools/profile_dropin.py"""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import argparse
import torch
import bench

args = argparse.Namespace(cfg=3, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=True, as_shard=None)
dev = torch.device("cuda")
import diff_gaussian_rasterization_depth as dgr
dgr.set_sync_mode("lazy")
prob = bench.build_problem(args, 0, 1, dev)
from dqo_harness.sharding import PackedAllReduce
step = bench.make_dropin_step(prob, dev, PackedAllReduce(bench.LOSS_SPEC, dev), optin=("optin" in sys.argv))
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("ms per iteration:", (time.perf_counter() - t0) / 50 * 1e3)
t0 = time.perf_counter()
for _ in range(50):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue time per iteration (no sync):", (t1 - t0) / 50 * 1e3, "ms; drained after", (time.perf_counter() - t1) * 1e3, "ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
