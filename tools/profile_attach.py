import argparse, os, sys, time
R = "/root/repo"
sys.path[:0] = [R, R + "/dqo-map_amd"]
import numpy as np, torch
import bench
from dqo_harness import scenes
from dqo_harness.fused_mapping import FusedMapper
args = argparse.Namespace(cfg=5, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
fm = FusedMapper(prob["scene"], prob["settings"], dev).set_object_gate(prob["gate"][0], prob["gate"][1])
P0 = fm.P
fm.reserve(32768)
sc = scenes.surfel_room(9000, 40_800, n_objects=32, rest_sigma=0.05)
nx = torch.tensor(np.ascontiguousarray(sc["xyz"], np.float32), device=dev)
nop = torch.tensor(np.ascontiguousarray(sc["opacity"], np.float32), device=dev).reshape(-1, 1)
nobj = torch.tensor(np.asarray(sc["obj_id"], np.int32), device=dev)
stable = torch.arange(fm.P, device=dev) < P0
for _ in range(3): fm._temp_points_attach(nx, nop, stable, 0.1, temp_obj=nobj)
import _dqo_native as N
N.profile_enable(True); N.profile_collect(reset=True)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5): fm._temp_points_attach(nx, nop, stable, 0.1, temp_obj=nobj)
torch.cuda.synchronize(); print("attach ms", (time.perf_counter()-t0)/5*1e3)
prof = N.profile_collect(reset=True); N.profile_enable(False)
print({k: round(v[0]/max(v[1],1)*1e3,1) for k,v in prof.items()}, "sum us", round(sum(v[0]/max(v[1],1)*1e3 for v in prof.values())))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(5): fm._temp_points_attach(nx, nop, stable, 0.1, temp_obj=nobj)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
