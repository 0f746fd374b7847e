import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd", R + "/tests"]
import numpy as np, torch
from dqo_harness import scenes, mapping
import diff_gaussian_rasterization_depth as dgr
dev = torch.device("cuda")
dgr.set_sync_mode("deferred")
for P in range(200000, 200000 + 8 * 7000, 7000):
    cam, sc = scenes.make_config(3, P=P)
    sc = {k: v for k, v in sc.items() if k != "normals"}
    st = mapping.make_settings(cam, dev)
    params = mapping.GaussianParams(sc, dev)
    gC = torch.randn(3, cam.H, cam.W, device=dev); gD = torch.randn(1, cam.H, cam.W, device=dev)
    prev = None
    for it in range(200):
        out = mapping.render(st, params.activated())
        torch.autograd.backward([out["render"], out["depth"]], [gC, gD])
        for grp in params.param_groups():
            for p in grp["params"]:
                p.grad = None
        prev = out  # (held across the next forward, like DQO-MAP's loop)
    dgr.verify_pending()
    torch.cuda.synchronize()
    sets = {k[2]: len(v) for k, v in dgr._pool.items()}
    print(f"P {P}: allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB, reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB, pooled contexts per shape {sets}", flush=True)
    del params, out, prev
