"""The opt-in loop through the reference's operator surface, captured into one torch.cuda.graph (bench.make_dropin_graph_step), replayed
under rocprofv3:   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cap -o cap -- python3 tools/profile_captured.py"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench

args = argparse.Namespace(cfg=3, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
step, graph = bench.make_dropin_graph_step(prob, dev)
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"captured opt-in loop: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per iteration")
