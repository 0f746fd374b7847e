"""Debug: does a shard grow exactly like the whole map (per-object growth decisions)?  python tools/debug_growth_n1.py [P] [iters]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import numpy as np, torch
from dqo_harness import mapping, scenes
from dqo_harness.fused_mapping import FusedMapper
import dqo_mapgrowth as mg

P = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda")
cam, scene = scenes.make_config(5, P=P)
settings = mapping.make_settings(cam, dev)
tgt = mapping.perturbed_target(scene, settings, dev, 12)
go = np.asarray(scene["obj_id"], np.int32)
po = tgt["pix_obj"].cpu().numpy()
new = scenes.surfel_room(9000, 40800, n_objects=32, rest_sigma=0.05)
objs_a = list(range(0, 32, 2))
in_a = np.isin(go, objs_a)
new_in_a = np.isin(np.asarray(new["obj_id"]), objs_a)
sub = lambda m: {k: (v[m] if hasattr(v, "shape") and v.shape[:1] == (P,) else v) for k, v in scene.items()}
nsub = lambda m: {k: (np.asarray(v)[m] if hasattr(v, "shape") and v.shape[:1] == (40800,) else v) for k, v in new.items()}


n_attach_whole = int((np.clip(scene["opacity"], 1e-4, 1 - 1e-4).reshape(-1) < 0.9).sum())


def run(sc, gobj, nw, px_objs):
    fm = FusedMapper(sc, settings, dev, attach_count_reducer=lambda n: n_attach_whole).set_object_gate(gobj, po).reserve(8192)
    n0 = len(gobj)
    mask = torch.tensor(np.isin(po, px_objs), device=dev)
    if iters:
        fm.capture(tgt["gt_color"], tgt["gt_depth"], mask, list_split=0)
        for _ in range(iters):
            fm.replay()
        torch.cuda.synchronize()
        assert not fm.graph_overflowed()
    stable = torch.arange(fm.P, device=dev) < n0
    st = fm.grow(nw, new_mapping_call=True, stable_mask=stable)
    alive = fm.alive.bool()
    f = lambda a: a[alive].cpu().numpy()
    rows = np.concatenate([f(fm.xyz), f(fm.scaling_raw), f(fm.opacity_raw), f(fm.rotation_raw)], 1)
    return st, f(fm.gaussian_object), rows, f(torch.arange(fm.P, device=dev)) >= n0


all_objs = sorted(set(go.tolist()))
st_w, obj_w, rows_w, new_w = run(scene, go, new, all_objs)
st_a, obj_a, rows_a, new_a = run(sub(in_a), go[in_a], nsub(new_in_a), objs_a)
print("whole", {k: v for k, v in st_w.items() if k not in ("rows", "kept_rows")})
print("shard", {k: v for k, v in st_a.items() if k not in ("rows", "kept_rows")})
canon = lambda r: r[np.lexsort((r[:, 2], r[:, 1], r[:, 0]))]
for k in objs_a:
    a, b = canon(rows_a[obj_a == k]), canon(rows_w[obj_w == k])
    if a.shape != b.shape:
        print("object", k, "counts", a.shape[0], b.shape[0])
        sa, sb = {tuple(r[:3]) for r in a.tolist()}, {tuple(r[:3]) for r in b.tolist()}
        print("  only in shard:", list(sa - sb)[:4], " only in whole:", list(sb - sa)[:4])
    else:
        d = (a != b).any(1)
        if d.any():
            i = np.nonzero(d)[0][:3]
            print("object", k, int(d.sum()), "rows differ, e.g.", a[i], b[i])
print("done")
