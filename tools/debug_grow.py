"""Where does a growth step spend its time?  python tools/debug_grow.py [cfg] [P]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dqo-map_amd"))
import numpy as np, torch
from dqo_harness import scenes
import dqo_mapgrowth as mg
from simple_knn._C import distCUDA2
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
cam, sc = scenes.make_config(cfg, P=P)
dev = torch.device("cuda")
t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
ex, esc = t(sc["xyz"]), t(sc["scales"])
er = (esc.sum(1) - esc.min(1).values) / 2
new = scenes.surfel_room(9000, 40800, n_objects=32, rest_sigma=0.05)
nx, nsc = t(new["xyz"]), t(new["scales"])
nr = (nsc.sum(1) - nsc.min(1).values) / 2
def timed(name, f, reps=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize()
    print(f"{name:50s} {(time.perf_counter() - t0) / reps * 1e3:8.2f} ms", flush=True)
    return r
inb = timed("bbox_filter(new, existing 2M)", lambda: mg.bbox_filter(nx, ex))
sub = timed("existing[inbbox] gather", lambda: (ex[inb], er[inb]))
timed("knn_points_k3(40800 queries, in-bbox refs)", lambda: mg.knn_points_k3(nx, sub[0]))
timed("temp_points_filter_mask (all of the above)", lambda: mg.temp_points_filter_mask(nx, ex, er))
keep = ~mg.temp_points_filter_mask(nx, ex, er)
kx, kr = nx[keep], nr[keep]
print("survivors", kx.shape[0], "refs in box", sub[0].shape[0])
timed("update_geometry_scales default", lambda: mg.update_geometry_scales(kx, kr, ex, er, 0.001, 0.05))
timed("update_geometry_scales literal (reference call)", lambda: mg.update_geometry_scales(kx, kr, ex, er, 0.001, 0.05, literal=True))
timed("distCUDA2(survivors)", lambda: distCUDA2(kx.contiguous()))
timed("distCUDA2(100k)", lambda: distCUDA2(ex[:100000].contiguous()))
timed("distCUDA2(540k)", lambda: distCUDA2(ex[:540000].contiguous()))
timed("distCUDA2(2M)", lambda: distCUDA2(ex))
import _dqo_native as N
N.profile_enable(True); N.profile_collect(reset=True)
mg.knn_points_k3(nx, sub[0]); distCUDA2(ex[:540000].contiguous())
torch.cuda.synchronize()
for k, v in sorted(N.profile_collect(reset=True).items(), key=lambda kv: -kv[1][0]):
    print(f"   {k:28s} total {v[0]:8.3f} ms  calls {v[1]}")
