"""Diagnostic (not a test): where does the largest colour difference between the HIP forward and the fp32 oracle come from?
python tests/diag_color.py <cfg> [P]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/dqo-map_amd', R + '/tests']
from dqo_harness import scenes
from oracle import oracle_lib as ol
import util_rast as U
cfg = int(sys.argv[1]); P = int(sys.argv[2]) if len(sys.argv) > 2 else None
cam, sc = scenes.make_config(cfg, P=P)
hr = U.HipRun(cam, sc, grad=False)
h = hr.res
o = ol.OracleRasterizer(np.float32, omp=True)
st = U.oracle_settings(ol, cam)
rr = o.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
r = {k: getattr(rr, k) for k in U.HipRun.names}
bad = U.flipped_pixels(h, r)
d = np.abs(h['color'] - r['color']).max(0)
d[bad] = 0
print("flipped", int(bad.sum()), "max colour diff", d.max())
# compare per-Gaussian colours: HIP with the oracle's rgb as precomputed colours
rgb = o.ctx('rgb')
hp = U.HipRun(cam, sc, grad=False, colors_precomp=rgb).res
d2 = np.abs(hp['color'] - r['color']).max(0); d2[bad] = 0
print("with the oracle's rgb as colors_precomp: max colour diff", d2.max())
for n in range(3):
    y, x = np.unravel_index(d.argmax(), d.shape)
    print('pixel', x, y, 'diff', d[y, x], 'with precomp', d2[y, x])
    for k in ('color', 'depth', 'hit_color', 'hit_depth', 'hit_color_weight', 'hit_depth_weight', 'T_map'):
        print('  ', k, h[k][:, y, x], r[k][:, y, x])
    gx = (cam.W + 15) // 16; t = (y // 16) * gx + x // 16
    rg = o.ctx('ranges')[t]; pl = o.ctx('point_list')[rg[0]:rg[1]]
    m2 = o.ctx('means2D'); co = o.ctx('conic_opacity')
    T = np.float32(1.0); C = np.zeros(3, np.float32); nb = 0
    for i, g in enumerate(pl):
        dx = m2[g, 0] - np.float32(x); dy = m2[g, 1] - np.float32(y)
        p = np.float32(-0.5) * (co[g, 0] * dx * dx + co[g, 2] * dy * dy) - co[g, 1] * dx * dy
        if p > 0: continue
        a = min(np.float32(0.99), co[g, 3] * np.exp(p))
        if a < np.float32(1 / 255):
            if a > np.float32(1 / 255) * 0.99: print('     near-threshold alpha', i, g, a, 'T', T)
            continue
        tT = T * (1 - a)
        if tT < 1e-4:
            print('     T-threshold entry', i, g, 'alpha', a, 'T', T, 'test_T', tT)
            T = tT
            continue
        nb += 1
        if abs(tT - 1e-4) < 3e-5 or abs(tT - 0.5) < 1e-4: print('     close to a T threshold', i, g, tT)
        T = tT
    print('   blended', nb, 'list', len(pl))
    d[y, x] = 0
