"""Where does a growth step (FusedMapper.grow + new mapping call + re-capture) spend its time?   python tools/profile_grow.py [cfg]"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import numpy as np, torch
import bench
import dqo_mapgrowth as mg
from dqo_harness import scenes
from dqo_harness.fused_mapping import FusedMapper

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
args = argparse.Namespace(cfg=cfg, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=True, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
mask = prob["render_mask"].to(torch.uint8).contiguous()
fm = FusedMapper(prob["scene"], prob["settings"], dev)
fm.capture(prob["gt_color"], prob["gt_depth"], mask, tile_mask=prob["tile_mask"])
for _ in range(20):
    fm.replay()
sc = scenes.surfel_room(9000, 40_800, n_objects=prob["cfgd"]["n_objects"], rest_sigma=prob["cfgd"]["rest_sigma"])
new = {n: torch.tensor(np.ascontiguousarray(sc[n], np.float32), device=dev) for n in ("xyz", "scales", "rotations", "opacity", "shs")}
delete = torch.zeros(fm.P, dtype=torch.bool, device=dev)
delete[::1500] = True


def timed(name, f, reps=3):
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"  {name:60s} {min(ts):8.2f} ms")
    return r


print("pieces of grow() on the current map (P =", fm.P, "):")
nx = new["xyz"]
rad = timed("radius()", lambda: fm.radius())
inside = timed("temp_points_filter_mask (bbox filter + knn query)", lambda: mg.temp_points_filter_mask(nx, fm.xyz, rad))
timed("  of which bbox_filter + boolean gather of the existing points", lambda: (fm.xyz[mg.bbox_filter(nx, fm.xyz)], rad[mg.bbox_filter(nx, fm.xyz)]))
ex = fm.xyz[mg.bbox_filter(nx, fm.xyz)]
timed("  of which knn_points_k3 alone", lambda: mg.knn_points_k3(nx, ex))
keep = (~inside).nonzero().reshape(-1)
nx2 = nx[keep]
nrad = (new["scales"][keep].sum(1) - new["scales"][keep].min(1).values) / 2
timed("update_geometry_scales", lambda: mg.update_geometry_scales(nx2, nrad, fm.xyz, rad, 0.001, 0.05))
keep_old = (~delete).nonzero().reshape(-1)
timed("gather of the kept rows: params (5 arrays)", lambda: [a[keep_old] for a in fm._params().values()])
timed("gather of the kept rows: moments (10 arrays)", lambda: [x[keep_old] for m, v in fm.state.values() for x in (m, v)])
print("pieces of the in-place step with the reference's two clouds (stable = the initial map):")
P0 = fm.P
fm.reserve(32768)
fm.capture(prob["gt_color"], prob["gt_depth"], mask, tile_mask=prob["tile_mask"])
stable = torch.arange(fm.P, device=dev) < P0
nop = new["opacity"].reshape(-1, 1)
timed("_temp_points_attach (stable-only render + projection)", lambda: fm._temp_points_attach(nx, nop, stable, 0.1))
timed("  of which the parked copy of xyz", lambda: torch.where(stable[:, None], fm.xyz, fm._park_position()[None, :]))
from dqo_harness import mapping
data = dict(xyz=fm.xyz, opacity=fm.opacity, scales=fm.scales, rotations=fm.rotations, shs=fm.shs)
timed("  of which mapping.render of the whole map", lambda: mapping.render(fm.settings, data))
nrad_all = (new["scales"].sum(1) - new["scales"].min(1).values) / 2
timed("update_geometry_scales on all 40 800 candidates", lambda: mg.update_geometry_scales(nx, nrad_all, fm.xyz, fm.radius(), 0.001, 0.05))
timed("begin_mapping_call (in place)", lambda: fm.begin_mapping_call(reset_optimizer=True))
timed("radius() of the map", lambda: fm.radius())
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = fm.grow(new, delete_mask=torch.cat([delete, torch.zeros(fm.P - delete.numel(), dtype=torch.bool, device=dev)]), new_mapping_call=True, stable_mask=stable,
                 attach_async=(i % 2 == 0))
    torch.cuda.synchronize()
    print(f"in-place grow #{i} ({'attach on its own thread' if i % 2 == 0 else 'in line'}): {1e3 * (time.perf_counter() - t0):.2f} ms", {k: v for k, v in st.items() if k not in ("rows", "kept_rows")})
