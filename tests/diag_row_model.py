"""Diagnostic (under tests/ because it runs the CPU oracle): instruction-budget model of a forward blend kernel whose four 16-lane DPP rows
(4x4 pixel blocks of an 8x8 quadrant) each walk their own compacted sub-list, against today's wave-per-quadrant walk — VERDICT r4 item 1.
The oracle's per-entry pixel masks say which (pixel, entry) pairs have arithmetic; the footprint test of dqo_cull.h, restated in numpy, says
which entries each candidate kernel would have to step through.      python tests/diag_row_model.py [cfg]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/dqo-map_amd', R + '/tests']
from dqo_harness import scenes
from oracle import oracle_lib as ol
import util_rast as U

f32 = np.float32


def hits_rect(mx, my, A, B, C, qthr, x0, y0, x1, y1):
    """dqo_cull.h::dqo_splat_hits_rect in float32 (vectorised)."""
    with np.errstate(all="ignore"):
        dx0, dx1, dy0, dy1 = (x0 - mx).astype(f32), (x1 - mx).astype(f32), (y0 - my).astype(f32), (y1 - my).astype(f32)
        inside = (dx0 <= 0) & (dx1 >= 0) & (dy0 <= 0) & (dy1 >= 0)
        invC, invA = f32(1) / C, f32(1) / A
        qmin = np.full(mx.shape, 3e38, f32)
        for dx in (dx0, dx1):
            dy = np.minimum(dy1, np.maximum(dy0, -B * dx * invC))
            qmin = np.minimum(qmin, A * dx * dx + f32(2) * B * dx * dy + C * dy * dy)
        for dy in (dy0, dy1):
            dx = np.minimum(dx1, np.maximum(dx0, -B * dy * invA))
            qmin = np.minimum(qmin, A * dx * dx + f32(2) * B * dx * dy + C * dy * dy)
        ddx, ddy = np.maximum(np.abs(dx0), np.abs(dx1)), np.maximum(np.abs(dy0), np.abs(dy1))
        tmax = np.abs(A) * ddx * ddx + f32(2) * np.abs(B) * ddx * ddy + np.abs(C) * ddy * ddy
        margin = f32(0.05) + f32(0.01) * qthr + f32(4e-6) * tmax
        return (qthr >= 0) & (inside | ~(qmin > qthr + margin))


def bbox_rect(mx, my, hx, hy, x0, y0, x1, y1):
    """cheap conservative test: the axis-aligned bounding box of the ellipse q <= qthr against the rectangle."""
    return (mx + hx >= x0) & (mx - hx <= x1) & (my + hy >= y0) & (my - hy <= y1)


cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cam, sc = scenes.make_config(cfg)
o = ol.OracleRasterizer(np.float32, omp=True)
st = U.oracle_settings(ol, cam)
o.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, shs=sc["shs"],
          scales=sc["scales"], rotations=sc["rotations"], pair_masks=True)
m = o.ctx("pair_mask")
N = m.shape[0]
bits = np.unpackbits(m.view(np.uint8).reshape(N, 32), axis=1, bitorder="little").reshape(N, 16, 16).astype(bool)  # [N, ty, tx]
ranges = o.ctx("ranges").astype(np.int64)
T = ranges.shape[0]
tile_of = np.repeat(np.arange(T), ranges[:, 1] - ranges[:, 0])
gid = o.ctx("point_list").astype(np.int64)
xy = o.ctx("means2D")[gid]
co = o.ctx("conic_opacity")[gid]
mx, my, A, B, C, op = xy[:, 0], xy[:, 1], co[:, 0], co[:, 1], co[:, 2], co[:, 3]
with np.errstate(all="ignore"):
    qthr = (f32(2) * np.log(f32(255) * np.maximum(op, f32(1e-30)))).astype(f32)
    det = A * C - B * B
    hx = np.sqrt(np.maximum(qthr + f32(0.1) + f32(0.02) * qthr, 0) * C / det) + f32(0.01)
    hy = np.sqrt(np.maximum(qthr + f32(0.1) + f32(0.02) * qthr, 0) * A / det) + f32(0.01)
gx = (cam.W + 15) // 16
tx0 = ((tile_of % gx) * 16).astype(f32)
ty0 = ((tile_of // gx) * 16).astype(f32)
reach_t = hits_rect(mx, my, A, B, C, qthr, tx0, ty0, tx0 + 15, ty0 + 15)
live_px = bits
live_t = live_px.any((1, 2))
assert not (live_t & ~reach_t).any(), "the footprint test dropped a live entry"
print(f"cfg {cfg}: reference instances {N}; HIP list (tile-level footprint test) {int(reach_t.sum())}; live for the tile {int(live_t.sum())}")

# keep only the HIP list
k = reach_t
tile_of, mx, my, A, B, C, qthr, hx, hy, tx0, ty0, live_px = tile_of[k], mx[k], my[k], A[k], B[k], C[k], qthr[k], hx[k], hy[k], tx0[k], ty0[k], live_px[k]
N = tile_of.size
first = np.searchsorted(tile_of, np.arange(T))
pos = np.arange(N) - first[tile_of]                 # list position inside the tile

# quadrants and blocks: reach (footprint test on the rectangle of pixel centres) and live (some pixel has arithmetic)
reach_q = np.zeros((N, 4), bool); live_q = np.zeros((N, 4), bool)
reach_b = np.zeros((N, 4, 4), bool); live_b = np.zeros((N, 4, 4), bool); box_b = np.zeros((N, 4, 4), bool)
for q in range(4):
    qx0, qy0 = tx0 + (q & 1) * 8, ty0 + (q >> 1) * 8
    reach_q[:, q] = hits_rect(mx, my, A, B, C, qthr, qx0, qy0, qx0 + 7, qy0 + 7)
    live_q[:, q] = live_px[:, (q >> 1) * 8:(q >> 1) * 8 + 8, (q & 1) * 8:(q & 1) * 8 + 8].any((1, 2))
    for r in range(4):
        bx0, by0 = qx0 + (r & 1) * 4, qy0 + (r >> 1) * 4
        reach_b[:, q, r] = hits_rect(mx, my, A, B, C, qthr, bx0, by0, bx0 + 3, by0 + 3)
        box_b[:, q, r] = bbox_rect(mx, my, hx, hy, bx0, by0, bx0 + 3, by0 + 3) & reach_q[:, q]
        yy, xx = (q >> 1) * 8 + (r >> 1) * 4, (q & 1) * 8 + (r & 1) * 4
        live_b[:, q, r] = live_px[:, yy:yy + 4, xx:xx + 4].any((1, 2))
assert not (live_q & ~reach_q).any() and not (live_b & ~reach_b).any() and not (live_b & ~box_b).any()
print(f"  (entry, quadrant): reach {reach_q.sum() / 1e6:.3f} M, live {live_q.sum() / 1e6:.3f} M;  (entry, block): exact-test reach "
      f"{reach_b.sum() / 1e6:.3f} M, bbox-test reach {box_b.sum() / 1e6:.3f} M, live {live_b.sum() / 1e6:.3f} M")


def last_true_pos(flag):  # per (tile): last list position with flag, -1 if none
    out = np.full(T, -1, np.int64)
    idx = np.nonzero(flag)[0]
    np.maximum.at(out, tile_of[idx], pos[idx])
    return out


# ---- today's kernel: one wave per quadrant, every entry with reach_q is a step until the quadrant's last live entry ----
tot = dict(chunks=0, valid=0, idle=0, waves=0)
endq = np.zeros((T, 4), np.int64)
for q in range(4):
    e = last_true_pos(live_q[:, q]); endq[:, q] = e
    walked = pos <= e[tile_of]
    tot["waves"] += int((e >= 0).sum())
    tot["chunks"] += int(((e[e >= 0] + 64) // 64).sum())
    tot["valid"] += int((live_q[:, q] & walked).sum())
    tot["idle"] += int((reach_q[:, q] & ~live_q[:, q] & walked).sum())
print(f"  today: waves with work {tot['waves']}, chunks {tot['chunks']}, steps with a valid lane {tot['valid'] / 1e6:.3f} M, steps without {tot['idle'] / 1e6:.3f} M")


def model(name, reach_rows, group_kind, test_cost):
    """rows of a quadrant wave walk their own sub-lists (entries with reach_rows[:, q, r]) in step inside a group; group_kind: 'pos64' /
    'pos128' = chunks of list positions, 'surv64' = groups of 64 entries that reach the quadrant, 'free' = no synchronisation at all."""
    steps = valid_steps = groups = 0
    for q in range(4):
        rq = reach_rows[:, q, :].any(1)
        if group_kind == "free":
            per = []
            for r in range(4):
                e = last_true_pos(live_b[:, q, r])
                w = reach_rows[:, q, r] & (pos <= e[tile_of])
                per.append(np.bincount(tile_of, weights=w, minlength=T))
            steps += int(np.stack(per, 1).max(1).sum()); continue
        if group_kind.startswith("pos"):
            G = int(group_kind[3:]); gkey = tile_of * 4096 + pos // G
        else:
            G = 64
            rank = np.cumsum(rq) - rq
            rank = rank - rank[first[tile_of]]
            gkey = tile_of * 4096 + rank // G
        uniq, ginv = np.unique(gkey, return_inverse=True)
        cnt = np.zeros((uniq.size, 4), np.int64)
        vmark = np.zeros(uniq.size * 64 * (2 if group_kind == "pos128" else 1), bool)
        stride = vmark.size // uniq.size
        ewave = endq[:, q]
        for r in range(4):
            e = last_true_pos(live_b[:, q, r])
            w = reach_rows[:, q, r] & (pos <= e[tile_of])
            np.add.at(cnt[:, r], ginv[w], 1)
            # rank of the entry inside its (group, row) sub-list = the step at which the row takes it
            order = np.nonzero(w)[0]
            g_ = ginv[order]
            start = np.searchsorted(g_, g_)  # first index with the same group (order is sorted by group)
            kk = np.arange(order.size) - start
            vmark[g_ * stride + kk] |= live_b[order, q, r]
        gsteps = cnt.max(1)
        steps += int(gsteps.sum()); valid_steps += int(vmark.sum())
        groups += int((np.bincount(ginv, weights=rq & (pos <= ewave[tile_of]), minlength=uniq.size) > 0).sum()) if group_kind == "surv64" else 0
    return steps, valid_steps, groups


print("  per-row sub-lists (steps = what the wave pays; 'valid' = steps in which some row has a valid lane):")
for name, rr, kind in (("exact block test, chunks of 64 list positions", reach_b, "pos64"),
                       ("exact block test, chunks of 128 list positions", reach_b, "pos128"),
                       ("exact block test, groups of 64 quadrant survivors", reach_b, "surv64"),
                       ("exact block test, rows free-running", reach_b, "free"),
                       ("bbox block test, chunks of 64 list positions", box_b, "pos64"),
                       ("bbox block test, groups of 64 quadrant survivors", box_b, "surv64"),
                       ("live entries only (the backward's lists), chunks of 64", live_b, "pos64"),
                       ("live entries only, chunks of 128", live_b, "pos128"),
                       ("live entries only, free-running", live_b, "free")):
    s, v, g = model(name, rr, kind, 0)
    print(f"    {name:58s} steps {s / 1e6:6.3f} M  valid {v / 1e6:6.3f} M  idle {(s - v) / 1e6:6.3f} M" + (f"  groups {g}" if g else ""))

# ---- the backward's groups: G quadrant-live entries per group (whatever list positions they sit at), rows in step inside a group ----
print("  backward, groups of G live entries of the quadrant (walk order), rows in step inside a group, batches of 7 steps:")
for G in (28, 35, 42, 49, 56, 63, 96, 128):
    steps = batches = groups = 0
    for q in range(4):
        lq_ = live_q[:, q]
        rank = np.cumsum(lq_) - lq_
        rank = rank - rank[first[tile_of]]
        gkey = (tile_of * 8192 + rank // G)[lq_]
        uniq, ginv = np.unique(gkey, return_inverse=True)
        cnt = np.stack([np.bincount(ginv, weights=live_b[lq_, q, r], minlength=uniq.size) for r in range(4)], 1)
        gs_ = cnt.max(1)
        steps += int(gs_.sum()); batches += int(np.ceil(gs_ / 7).sum()); groups += uniq.size
    print(f"    G = {G:3d}: groups {groups:6d}  steps {steps / 1e6:.3f} M ({steps / live_q.sum():.1%} of today's {live_q.sum() / 1e6:.3f} M)  "
          f"reduce batches of 7: {batches}")
b0 = 0
for q in range(4):  # today: one reduce batch per 7 live entries of a 64-position chunk (walk order from the quadrant's last live entry)
    lq_ = live_q[:, q]
    e = endq[:, q][tile_of]
    ck = (tile_of * 4096 + (e - pos) // 64)[lq_]
    b0 += int(np.ceil(np.unique(ck, return_counts=True)[1] / 7).sum())
print(f"    today: steps {live_q.sum() / 1e6:.3f} M, reduce batches of 7: {b0}")
