"""Critical path of tile_sort_wave_kernel by list length: ONE tile (a 16 x 16 image), n Gaussians in it — the kernel's only busy wave(s).
    python tools/sort_path_times.py"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
from dqo_harness import scenes, mapping
import _dqo_native as N
import diff_gaussian_rasterization_depth as dgr

cam = scenes.Camera(16, 16, 40.0, 40.0, 7.5, 7.5)
dev = torch.device("cuda")
st = mapping.make_settings(cam, dev)
for P in (40, 60, 120, 250, 500, 700, 1000, 2000):
    rng = np.random.default_rng(0)
    sc = scenes.frustum_cloud(5, P, cam, zmin=1.0, zmax=4.0)
    pc = np.stack([rng.uniform(-0.05, 0.05, P), rng.uniform(-0.05, 0.05, P), rng.uniform(1.0, 4.0, P)], 1)
    pc[:, :2] *= pc[:, 2:3]
    sc["xyz"] = pc.astype(np.float32)
    sc["opacity"] = rng.uniform(0.02, 0.08, (P, 1)).astype(np.float32)
    p = mapping.GaussianParams(sc, dev).activated()
    with torch.no_grad():
        for _ in range(3):
            mapping.render(st, p)
        torch.cuda.synchronize()
        N.profile_enable(True); N.profile_collect(reset=True)
        for _ in range(20):
            mapping.render(st, p)
        torch.cuda.synchronize()
        prof = N.profile_collect(reset=True); N.profile_enable(False)
    h = dgr.last_header()
    k = {n: round(v[0] / max(v[1], 1) * 1e3, 1) for n, v in prof.items()}
    print(f"list of {h['max_tile_count']:5d} entries ({h['num_tiles']} tile): sort_wave {k.get('tile_sort_wave_kernel')} us, sort_long {k.get('tile_sort_kernel')} us, "
          f"zero {k.get('zero_words_kernel')} us, preprocess {k.get('preprocess_kernel')} us")
