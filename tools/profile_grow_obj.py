"""Pieces of the per-object growth step (FusedMapper.grow with an object gate) on cfg 5:   python tools/profile_grow_obj.py"""
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
sc = scenes.surfel_room(9000, 40_800, n_objects=32, rest_sigma=0.05)
new = {n: torch.tensor(np.ascontiguousarray(sc[n], np.float32), device=dev) for n in ("xyz", "scales", "rotations", "opacity", "shs")}
new["obj_id"] = torch.tensor(np.asarray(sc["obj_id"], np.int32), device=dev)
stable = torch.arange(fm.P, device=dev) < P0


def timed(name, f, reps=3):
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"  {name:70s} {min(ts):8.2f} ms")
    return r


nx, nobj = new["xyz"], new["obj_id"]
rad = fm.radius()
alive = fm.alive.bool().nonzero().reshape(-1)
ex, er, eo = fm.xyz[alive], rad[alive], fm.gaussian_object[alive]
nrad = (new["scales"].sum(1) - new["scales"].min(1).values) / 2
timed("update_geometry_scales (reference decisions)", lambda: mg.update_geometry_scales(nx, nrad, ex, er, 0.001, 0.05))
timed("update_geometry_scales_per_object (16 m cells)", lambda: mg.update_geometry_scales_per_object(nx, nobj, nrad, ex, er, eo, 0.001, 0.05))
timed("update_geometry_scales_per_object (8 x 4 x 8 m cells)", lambda: mg.update_geometry_scales_per_object(nx, nobj, nrad, ex, er, eo, 0.001, 0.05, cell=fm.object_cell))
timed("  _per_object_bbox_mask", lambda: mg._per_object_bbox_mask(nx, nobj, ex, eo))
inb = mg._per_object_bbox_mask(nx, nobj, ex, eo)
print("  refs inside the per-object boxes:", int(inb.sum().item()), "of", ex.shape[0], "; global box:", int(mg.bbox_filter(nx, ex).sum().item()))
timed("  boolean gathers of the refs (xyz, radius, obj)", lambda: (ex[inb], er[inb], eo[inb]))
ex2, eo2 = ex[inb], eo[inb]
sh_q, sh_r = (nx + mg.object_offsets(nobj)).contiguous(), (ex2 + mg.object_offsets(eo2)).contiguous()
timed("  knn_points_k3 on shifted coordinates", lambda: mg.knn_points_k3(sh_q, sh_r))
timed("  knn_points_k3 on unshifted coordinates (same sets)", lambda: mg.knn_points_k3(nx, ex2))
timed("  distCUDA2 on shifted candidates", lambda: mg.distCUDA2(sh_q))
timed("temp_points_filter_mask_per_object (against all rows)", lambda: mg.temp_points_filter_mask_per_object(nx, nobj, ex, er, eo))
nop = new["opacity"].reshape(-1, 1)
timed("_temp_points_attach gated", lambda: fm._temp_points_attach(nx, nop, stable, 0.1, temp_obj=nobj))
timed("_temp_points_attach ungated", lambda: fm._temp_points_attach(nx, nop, stable, 0.1))
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = fm.grow(new, new_mapping_call=True, stable_mask=stable)
    torch.cuda.synchronize()
    print(f"grow #{i}: {1e3 * (time.perf_counter() - t0):.2f} ms", {k: v for k, v in st.items() if k not in ("rows", "kept_rows")})

# where a growth step's host time goes (the step is a chain of small kernels with host-side decisions in between)
import cProfile, pstats, io
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
fm.grow(new, new_mapping_call=True, stable_mask=stable)
torch.cuda.synchronize()
pr.disable()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(28)
print(sio.getvalue()[:6000])

# ... and the step's pieces one after the other (every piece synchronised: their GPU cost without the overlap of attach and scale init)
import functools
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    @functools.wraps(f)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
        return r
    setattr(obj, name, g)
wrap(fm, "_temp_points_attach", "attach (gated render of the stable cloud + gathers)")
wrap(mg, "update_geometry_scales_per_object", "scale init (grouped 3-NN against the map + 3-NN among the new points)")
wrap(mg, "temp_points_filter_mask_per_object", "filter (grouped 3-NN against the unstable cloud)")
wrap(fm, "begin_mapping_call", "new mapping call (Adam reset, init_stat, attach set)")
torch.cuda.synchronize(); t0 = time.perf_counter()
fm.grow(new, new_mapping_call=True, stable_mask=stable, attach_async=False)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
print(f"serialised growth step {tot:.2f} ms:", {k: round(v, 2) for k, v in acc.items()}, "rest", round(tot - sum(acc.values()), 2))

# ... and with a NEW batch per step, as bench.py --cfg 5 feeds them (every step adds ~1 300 Gaussians, so the row writes, the new
# mapping call over a changed attach set and a growing unstable cloud are in it): the serialised pieces of four such steps
for i in range(4):
    sc = scenes.surfel_room(9100 + 17 * i, 40_800, n_objects=32, rest_sigma=0.05)
    nb = {n: torch.tensor(np.ascontiguousarray(sc[n], np.float32), device=dev) for n in ("xyz", "scales", "rotations", "opacity", "shs")}
    nb["obj_id"] = torch.tensor(np.asarray(sc["obj_id"], np.int32), device=dev)
    acc.clear()
    asy = i % 2 == 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = fm.grow(nb, new_mapping_call=True, stable_mask=stable, attach_async=asy)
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
    print(f"new batch {i} ({'overlapped' if asy else 'serialised'}): {tot:.2f} ms, added {st['added']}:", {k.split(' (')[0]: round(v, 2) for k, v in acc.items()},
          "rest", round(tot - sum(acc.values()), 2) if not asy else "-")
