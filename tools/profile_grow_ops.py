"""How many torch ops does a growth step issue, and where?  One serialised FusedMapper.grow (new batch, cfg 5, object gate) under
torch.profiler (CPU activity: op counts and host time per op; the step is host-bound, so the op count is what it costs).
    python tools/profile_grow_ops.py"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import numpy as np, torch
import bench
import dqo_mapgrowth as mg
from dqo_harness import scenes
from dqo_harness.fused_mapping import FusedMapper

args = argparse.Namespace(cfg=5, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
fm = FusedMapper(prob["scene"], prob["settings"], dev).set_object_gate(prob["gate"][0], prob["gate"][1])
P0 = fm.P
fm.reserve(32768)
fm.object_cell = (8.0, 4.0, 8.0)
stable = torch.arange(fm.P, device=dev) < P0


def batch(i):
    sc = scenes.surfel_room(9100 + 17 * i, 40_800, n_objects=32, rest_sigma=0.05)
    nb = {n: torch.tensor(np.ascontiguousarray(sc[n], np.float32), device=dev) for n in ("xyz", "scales", "rotations", "opacity", "shs")}
    nb["obj_id"] = torch.tensor(np.asarray(sc["obj_id"], np.int32), device=dev)
    return nb


for i in range(3):
    fm.grow(batch(i), new_mapping_call=True, stable_mask=stable)
torch.cuda.synchronize()
import functools
from torch.profiler import profile, ProfilerActivity, record_function
def wrap(obj, name, label):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        with record_function("## " + label):
            return f(*a, **k)
    setattr(obj, name, g)
wrap(fm, "_temp_points_attach", "attach")
wrap(mg, "update_geometry_scales_per_object", "scale init")
wrap(mg, "temp_points_filter_mask_per_object", "filter")
wrap(fm, "begin_mapping_call", "new mapping call")
wrap(fm, "radius", "radius")
nb = batch(7)
with profile(activities=[ProfilerActivity.CPU]) as prof:
    with record_function("## grow"):
        fm.grow(nb, new_mapping_call=True, stable_mask=stable, attach_async=False)
    torch.cuda.synchronize()
ev = prof.key_averages()
tot_ops = sum(e.count for e in ev if e.key.startswith("aten::"))
print("aten ops (incl. nested):", tot_ops)
print(ev.table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
print("events with more than 40 us of self CPU time, in issue order (name < parents):")
for e in sorted(prof.events(), key=lambda e: e.time_range.start):
    if e.self_cpu_time_total > 40 and not e.name.startswith("##"):
        chain, p = [], e.cpu_parent
        while p is not None:
            chain.append(p.name)
            p = p.cpu_parent
        shapes = ""
        print(f"  {e.self_cpu_time_total:8.0f} us  {e.name:28s} < {' < '.join(chain)[:150]}")
