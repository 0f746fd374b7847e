"""How long does one long tile list take to sort?  A few Gaussian clouds squeezed into a 20 x 20 pixel window (the scene of
tests/test_gpu_rast_edge.py::test_long_tile_lists_global_sort): per-kernel event times of the forward by list length.
    python tools/sort_times.py"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
from dqo_harness import scenes, mapping
import _dqo_native as N
import diff_gaussian_rasterization_depth as dgr

cam = scenes.Camera(96, 64, 80.0, 80.0, 47.5, 31.5)
dev = torch.device("cuda")
st = mapping.make_settings(cam, dev)
for P in (1500, 2600, 5000, 9000, 14000):
    rng = np.random.default_rng(0)
    sc = scenes.frustum_cloud(5, P, cam, zmin=1.0, zmax=4.0)
    pc = np.stack([rng.uniform(-0.12, 0.12, P), rng.uniform(-0.12, 0.12, P), rng.uniform(1.0, 4.0, P)], 1)
    pc[:, :2] *= pc[:, 2:3]
    sc["xyz"] = pc.astype(np.float32)
    sc["opacity"] = rng.uniform(0.02, 0.08, (P, 1)).astype(np.float32)
    p = mapping.GaussianParams(sc, dev).activated()
    with torch.no_grad():
        for _ in range(3):
            mapping.render(st, p)
        torch.cuda.synchronize()
        N.profile_enable(True); N.profile_collect(reset=True)
        for _ in range(10):
            mapping.render(st, p)
        torch.cuda.synchronize()
        prof = N.profile_collect(reset=True); N.profile_enable(False)
    h = dgr.last_header()
    k = {n: round(v[0] / max(v[1], 1) * 1e3, 1) for n, v in prof.items()}
    print(f"P {P}: longest list {h['max_tile_count']}, instances {h['num_rendered']}: sort_wave {k.get('tile_sort_wave_kernel')} us, "
          f"sort_long {k.get('tile_sort_kernel')} us, blend_forward {k.get('blend_forward_kernel')} us")
