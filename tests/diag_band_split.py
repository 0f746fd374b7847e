"""Diagnostic (under tests/ because it runs the CPU oracle): what would cutting objects into SCREEN BANDS buy the strong-scaling job?
("count the straddlers first".)  For a BASELINE configuration: every object's on-screen work by tile row, the LPT
assignment of whole objects to N ranks (what bench.py --gpus N does), and the assignment when objects may be cut into horizontal bands of
tile rows — with the Gaussians that would have to live on two ranks (their tile rect crosses a band edge: rendered by both neighbours,
their gradient rows exchanged every iteration).  Work model = bench.py's: candidate (Gaussian, tile) pairs in the view + 0.15 per Gaussian.

    python tests/diag_band_split.py [cfg] [N]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/dqo-map_amd', R + '/tests']
from dqo_harness import scenes, sharding
from oracle import oracle_lib as ol
import util_rast as U

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cam, sc = scenes.make_config(cfg)
o = ol.OracleRasterizer(np.float32, omp=True)
st = U.oracle_settings(ol, cam)
r = o.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, shs=sc["shs"],
              scales=sc["scales"], rotations=sc["rotations"])
obj = np.asarray(sc["obj_id"], np.int64)
P = obj.size
gx, gy = (cam.W + 15) // 16, (cam.H + 15) // 16
radii = r.radii.astype(np.int64)
xy = o.ctx("means2D")
vis = radii > 0
# tile rect of every visible Gaussian (auxiliary.h:49-57 getRect)
x0 = np.clip(((xy[:, 0] - radii) / 16).astype(np.int64), 0, gx); x1 = np.clip(((xy[:, 0] + radii + 15) / 16).astype(np.int64), 0, gx)
y0 = np.clip(((xy[:, 1] - radii) / 16).astype(np.int64), 0, gy); y1 = np.clip(((xy[:, 1] + radii + 15) / 16).astype(np.int64), 0, gy)
x0, x1, y0, y1 = (np.where(vis, a, 0) for a in (x0, x1, y0, y1))
width = x1 - x0
work_g = width * (y1 - y0) + 0.15            # per Gaussian: candidate pairs + the fixed cost
objs = np.unique(obj)
K = objs.size
# work of object k in tile row y: every Gaussian spreads its pairs over its rows
W_ky = np.zeros((K, gy))
for k_i, k in enumerate(objs):
    m = (obj == k) & vis
    for yy in range(gy):
        W_ky[k_i, yy] = width[m & (y0 <= yy) & (yy < y1)].sum()
fixed_k = np.array([0.15 * (obj == k).sum() for k in objs])
W_k = W_ky.sum(1) + fixed_k
total = W_k.sum()
print(f"cfg {cfg}: P = {P}, visible {int(vis.sum())}, objects {K}, work (pairs + 0.15 / Gaussian) {total / 1e6:.3f} M; N = {N}")
print("  object work share:", ", ".join(f"{int(k)}: {w / total:.1%}" for k, w in zip(objs, W_k)))

# (1) whole objects, LPT — bench.py's assignment
assign = sharding.assign_objects({int(k): float(w) for k, w in zip(objs, W_k)}, N)
load = np.zeros(N)
for k, w in zip(objs, W_k):
    load[assign[int(k)]] += w
print(f"  whole objects (LPT): slowest rank holds {load.max() / total:.1%} of the work -> speed-up bound {total / load.max():.2f}x of {N}")

# (2) objects cut into bands of tile rows: greedy — repeatedly cut the piece that makes the most loaded rank, at the row that halves it
pieces = [dict(k=int(k), ki=i, ya=0, yb=gy, w=float(W_k[i])) for i, k in enumerate(objs)]
def piece_work(ki, ya, yb):
    share = W_ky[ki, ya:yb].sum() / max(W_ky[ki].sum(), 1e-9)
    return W_ky[ki, ya:yb].sum() + fixed_k[ki] * share
def lpt(ps):
    ld = np.zeros(N); where = []
    for p in sorted(ps, key=lambda p: -p["w"]):
        s = int(np.argmin(ld)); ld[s] += p["w"]; where.append((p, s))
    return ld, where
for cuts in range(0, 4 * N):
    ld, where = lpt(pieces)
    if ld.max() <= 1.03 * total / N:
        break
    big = max(pieces, key=lambda p: p["w"])
    if big["yb"] - big["ya"] < 2:
        break
    cum = np.cumsum(W_ky[big["ki"], big["ya"]:big["yb"]])
    cut = big["ya"] + 1 + int(np.searchsorted(cum, cum[-1] / 2))
    cut = min(max(cut, big["ya"] + 1), big["yb"] - 1)
    pieces.remove(big)
    for ya, yb in ((big["ya"], cut), (cut, big["yb"])):
        pieces.append(dict(k=big["k"], ki=big["ki"], ya=ya, yb=yb, w=float(piece_work(big["ki"], ya, yb))))
ld, where = lpt(pieces)
# straddlers: Gaussians of a cut object whose tile rect reaches into a band held by ANOTHER rank
rank_of = {}
for p, s in where:
    rank_of.setdefault(p["k"], []).append((p["ya"], p["yb"], s))
n_str = 0; n_cut_vis = 0; extra_pairs = 0
for k, bands in rank_of.items():
    if len(bands) < 2:
        continue
    m = (obj == k) & vis
    n_cut_vis += int(m.sum())
    ranks_touched = np.zeros((int(m.sum()), N), bool)
    for ya, yb, s in bands:
        ranks_touched[:, s] |= (y0[m] < yb) & (y1[m] > ya)
    multi = ranks_touched.sum(1) > 1
    n_str += int(multi.sum())
n_pieces = len(pieces)
print(f"  bands of tile rows ({n_pieces} pieces, {n_pieces - K} cuts): slowest rank holds {ld.max() / total:.1%} -> speed-up bound {total / ld.max():.2f}x of {N}")
print(f"  cut objects: {sorted(k for k, b in rank_of.items() if len(b) > 1)}; their visible Gaussians {n_cut_vis}, of which on two or more ranks "
      f"(tile rect crosses a band edge between ranks) {n_str} = {n_str / max(n_cut_vis, 1):.1%}; exchanged per iteration: {n_str} rows x 236 B = "
      f"{n_str * 236 / 1e6:.2f} MB of gradient rows (+ the per-object pixel counts)")
