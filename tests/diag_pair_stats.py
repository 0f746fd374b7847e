"""Diagnostic (lives under tests/ because it runs the CPU oracle, which is test infrastructure): workload statistics of the blend loops:
for every list entry of every tile the set of pixels that has arithmetic for it, and from that how well different wave -> pixel
mappings of a blend kernel would be utilised.   python tests/diag_pair_stats.py [cfg] [P]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/dqo-map_amd', R + '/tests']
from dqo_harness import scenes
from oracle import oracle_lib as ol
import util_rast as U

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
cam, sc = scenes.make_config(cfg, P=P)
o = ol.OracleRasterizer(np.float32, omp=True)
st = U.oracle_settings(ol, cam)
r = o.forward(st, sc["xyz"], sc["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, shs=sc["shs"],
              scales=sc["scales"], rotations=sc["rotations"], pair_masks=True)
m = o.ctx("pair_mask")  # [N, 4] uint64: rows 0-3, 4-7, 8-11, 12-15 of the tile, 16 bits per row
N = m.shape[0]
bits = np.unpackbits(m.view(np.uint8).reshape(N, 32), axis=1, bitorder="little").reshape(N, 16, 16).astype(bool)  # [N, ty, tx]
live_tile = bits.any((1, 2))
print(f"cfg {cfg}: reference instances {N}, with any arithmetic {int(live_tile.sum())} ({live_tile.mean():.1%}); (pixel, entry) pairs "
      f"{int(bits.sum())} = {bits.sum() / (cam.W * cam.H):.1f} per pixel")
pairs = bits.sum()
def report(name, blocks):  # blocks: [N, nblocks, lanes] bool
    live = blocks.any(2)
    steps = live.sum()
    print(f"  {name:46s} wave steps {steps / 1e6:6.3f} M   lane utilisation {pairs / (steps * blocks.shape[2]):.1%}")
    return live
q = bits.reshape(N, 2, 8, 2, 8).transpose(0, 1, 3, 2, 4).reshape(N, 4, 64)          # 8x8 quadrants
lq = report("8x8 quadrant per wave (current)", q)
h = bits.reshape(N, 4, 4, 2, 8).transpose(0, 1, 3, 2, 4).reshape(N, 8, 32)           # 8 wide x 4 high half-quadrants
lh = report("8x4 half-quadrant (one 32-lane half)", h)
# a wave = quadrant, but a half with no active lane costs nothing (if the hardware skips an all-zero EXEC half):
lh4 = lh.reshape(N, 2, 2, 2)  # [N, quad row, half, quad col]
both = (lh4[:, :, 0, :] & lh4[:, :, 1, :]).sum(); one = (lh4[:, :, 0, :] ^ lh4[:, :, 1, :]).sum()
print(f"     quadrant steps with both halves active {both / 1e6:.3f} M, with one half only {one / 1e6:.3f} M -> issue cost "
      f"{(both + 0.5 * one) / 1e6:.3f} M full-wave equivalents if an idle half is skipped; {(both + 0) / 1e6:.3f} + max-packing of the rest")
b44 = bits.reshape(N, 4, 4, 4, 4).transpose(0, 1, 3, 2, 4).reshape(N, 16, 16)       # 4x4 blocks (one DPP row)
l44 = report("4x4 block (one DPP row of 16 lanes)", b44)
w16 = bits.reshape(N, 1, 256)
report("16x16 tile on 4 waves, all in step", w16)
s164 = bits.reshape(N, 4, 4, 16).reshape(N, 4, 64)                                    # 16 wide x 4 high strips
report("16x4 strip per wave", s164)
s416 = bits.reshape(N, 16, 4, 4).transpose(0, 2, 1, 3).reshape(N, 4, 64)            # 4 wide x 16 high strips
report("4x16 strip per wave", s416)
# per-tile packing bound for half-waves: per (tile, quadrant) the live halves could be paired across DIFFERENT entries
ranges = o.ctx("ranges")
tile_of = np.repeat(np.arange(ranges.shape[0]), (ranges[:, 1] - ranges[:, 0]).astype(np.int64))
steps_packed = 0
for qr in range(2):
    for qc in range(2):
        top, bot = lh4[:, qr, 0, qc], lh4[:, qr, 1, qc]
        bo = np.bincount(tile_of, weights=(top & bot), minlength=ranges.shape[0])
        to = np.bincount(tile_of, weights=(top & ~bot), minlength=ranges.shape[0])
        bt = np.bincount(tile_of, weights=(~top & bot), minlength=ranges.shape[0])
        steps_packed += (bo + np.maximum(to, bt)).sum()
print(f"  half-entries of one quadrant packed pairwise (top-only with bottom-only): {steps_packed / 1e6:.3f} M wave steps "
      f"(utilisation {pairs / (steps_packed * 64):.1%})")

# ---- critical path: live entries per (tile, quadrant) wave ----
T = ranges.shape[0]
lq_i = lq.astype(np.int64)  # [N, 4]
per_wave = np.stack([np.bincount(tile_of, weights=lq_i[:, q_], minlength=T) for q_ in range(4)], 1).reshape(-1)
per_wave = per_wave[per_wave > 0]
print(f"  quadrant waves with work: {per_wave.size}; live entries per wave: mean {per_wave.mean():.1f}, median {np.median(per_wave):.0f}, "
      f"p90 {np.quantile(per_wave, .9):.0f}, p99 {np.quantile(per_wave, .99):.0f}, max {per_wave.max():.0f}")
tot = per_wave.sum()
for slots in (1024 * 1, 1024 * 2, 1024 * 4):
    print(f"     perfectly balanced over {slots} wave slots: {tot / slots:.0f} live entries per slot (the longest wave alone has {per_wave.max():.0f})")
lens = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
print(f"  reference list length per tile: mean {lens.mean():.0f}, max {lens.max()}")

# ---- aligned pairing (round 4): the two halves of a quadrant walk their own sub-lists, but an entry live in BOTH halves is taken by both in
# the same step (so that its record stays one per (quadrant, entry)): between two such entries the top-only and the bottom-only
# entries are paired up.  Steps per (tile, quadrant) = #both + sum over the segments between them of max(#top-only, #bottom-only).
def aligned_steps(first, second):
    """first / second: [N] bool liveness of the two halves of one quadrant position (tile-major list order)."""
    both = first & second
    seg = np.cumsum(both) - both  # segment id inside the whole array: entries between consecutive 'both' entries share it ...
    seg = seg + tile_of * (N + 1)  # ... and never across tiles
    uniq, inv = np.unique(seg, return_inverse=True)
    t = np.bincount(inv, weights=(first & ~second), minlength=uniq.size)
    b = np.bincount(inv, weights=(~first & second), minlength=uniq.size)
    return both.sum() + np.maximum(t, b).sum()
tot_tb = sum(aligned_steps(lh4[:, qr, 0, qc], lh4[:, qr, 1, qc]) for qr in range(2) for qc in range(2))
print(f"  top / bottom halves, aligned on the entries live in both: {tot_tb / 1e6:.3f} M wave steps ({tot_tb / lq.sum():.1%} of the quadrant steps)")
hv = bits.reshape(N, 2, 8, 4, 4).transpose(0, 1, 3, 2, 4).reshape(N, 2, 4, 32).any(3)  # [N, quad row, 4 column strips of 4 px]: left / right halves
tot_lr = sum(aligned_steps(hv[:, qr, 2 * qc], hv[:, qr, 2 * qc + 1]) for qr in range(2) for qc in range(2))
print(f"  left / right halves (4 wide x 8 high), aligned: {tot_lr / 1e6:.3f} M wave steps ({tot_lr / lq.sum():.1%})")

# ---- per-DPP-row sub-lists (round 5): a quadrant wave whose four 16-lane rows each walk their OWN compacted sub-list ----
def row_sublists(name, blocks4):
    """blocks4: [N, 4 quadrants, 4 rows] bool — is the entry live for that 16-pixel group of that quadrant."""
    tot_async = 0.0
    tot_sync = {64: 0.0, 128: 0.0}
    # list position inside the tile among the entries with any arithmetic (approximates the culled HIP list)
    pos_live = np.cumsum(live_tile) - live_tile
    first = np.zeros(T, np.int64); first[tile_of[::-1]] = np.arange(N)[::-1]  # first instance index of each tile
    pos_live = pos_live - pos_live[first[tile_of]]
    for q_ in range(4):
        rows = blocks4[:, q_, :]  # [N, 4]
        per_row = np.stack([np.bincount(tile_of, weights=rows[:, r_], minlength=T) for r_ in range(4)], 1)
        tot_async += per_row.max(1).sum()
        for CH in tot_sync:
            key = tile_of * 4096 + pos_live // CH
            uniq, inv = np.unique(key[live_tile], return_inverse=True)
            pr = np.stack([np.bincount(inv, weights=rows[live_tile, r_], minlength=uniq.size) for r_ in range(4)], 1)
            tot_sync[CH] += pr.max(1).sum()
    s = f"  {name:34s} rows free-running: {tot_async / 1e6:.3f} M steps ({tot_async / lq.sum():.1%} of today's)"
    for CH, v_ in tot_sync.items():
        s += f"; in step per {CH}-entry chunk: {v_ / 1e6:.3f} M ({v_ / lq.sum():.1%})"
    print(s)
l44q = l44.reshape(N, 2, 2, 2, 2).transpose(0, 1, 3, 2, 4).reshape(N, 4, 4)   # [N, by, bx] -> [N, quadrant (qy, qx), row (ry, rx)]
row_sublists("4x4 blocks as DPP rows", l44q)
s82 = bits.reshape(N, 8, 2, 2, 8).transpose(0, 1, 3, 2, 4).reshape(N, 8, 2, 16).any(3)   # [N, strip y (8), qx (2)]
s82q = s82.reshape(N, 2, 4, 2).transpose(0, 1, 3, 2).reshape(N, 4, 4)                      # [N, quadrant (qy, qx), strip]
row_sublists("8x2 strips as DPP rows (today's map)", s82q)
