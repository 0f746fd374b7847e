"""The drop-in op's pooled contexts over the bench's five window cameras (one shape, five poses): header of every frame, the pooled
context's sizes, whether the frame fitted.  python tools/diag_pool_window.py [cfg]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench
import diff_gaussian_rasterization_depth as dgr
from dqo_harness import mapping

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
args = argparse.Namespace(cfg=cfg, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
cams = bench.window_cameras(prob["cam"], 5)
params = mapping.GaussianParams(prob["full"], dev)
dgr.set_sync_mode("deferred")
for rep in range(3):
    for k, cam in enumerate(cams):
        st = mapping.make_settings(cam, dev)
        with torch.no_grad():
            mapping.render(st, params.activated())
        h = dgr.last_header()
        sets = [(cs.cap, cs.bucket, cs.leased) for v in dgr._pool.values() for cs in v]
        print(f"pass {rep} camera {k}: N {h['num_rendered']} candidates {h['num_candidates']} longest {h['max_tile_count']} overflow {h['overflow']}; "
              f"hint {list(dgr._shape_hint.values())}; pooled contexts (cap, bucket, leased) {sets}", flush=True)
        try:
            dgr.verify_pending()
        except RuntimeError as e:
            print("   ->", str(e)[:160], flush=True)
