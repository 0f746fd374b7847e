// Tile binning for gfx950: which (Gaussian, tile) pairs become list entries, and where.  Replaces the per-Gaussian tile
// loops of
//   /root/reference/submodules/diff-gaussian-rasterizer-depth/cuda_rasterizer/forward.cu:344-353 (tiles_touched) and
//   cuda_rasterizer/rasterizer_impl.cu:70-115 (duplicateWithKeys) + the cub scan of :303,
// where one thread walks its Gaussian's whole tile rect: a wave then runs as long as its largest splat and every
// iteration waits for its own atomic.
//
// bin_count_kernel: a block owns 256 consecutive Gaussians, scans their rect areas in LDS and hands out ONE candidate
//   (Gaussian, tile) pair per thread-iteration (binary search in the scanned offsets), so lanes are evenly loaded.  Each
//   candidate takes the output-invariant footprint test of dqo_cull.h ONCE; the survivors are numbered per Gaussian (rect
//   order -> gaussian-major slot = slot_base + k: a fixed order, so the backward's per-Gaussian sum is reproducible) and
//   take their rank inside their tile from the tile histogram's atomic counter.  (tile, rank, Gaussian) goes into the
//   slot-indexed info table.
// tile_scan_kernel (rast_forward.hip) turns the histogram into the tiles' list ranges.
// bin_place_kernel: one thread per slot, no atomics, no decoding: position = range start + rank; writes the sort key and
//   the slot payload.
//
// Device-scope atomics execute memory-side on MI355X; the counters are spread one per 256 bytes (DQO_TSTRIDE) and every
// thread keeps up to four of them in flight.
#include "dqo_common.h"
#include "dqo_cull.h"
#include "dqo_k1_early.h"

namespace {

constexpr int BIN_THREADS = 256;
constexpr int BIN_CHUNK = BIN_THREADS;  // Gaussians per block, one per thread
constexpr int BIN_WINDOW = 16384;       // candidate pairs whose live bits fit the LDS bit array at once
constexpr int BIN_FLIGHT = 4;           // returning atomics in flight per thread
#ifndef BIN_WAVES
#define BIN_WAVES 8  // waves per SIMD the register allocation leaves room for (8: at most 64 VGPRs)
#endif

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of bits[] in the bit range [a, b)
__device__ __forceinline__ uint32_t popcount_range(const uint32_t* bits, uint32_t a, uint32_t b) {
    if (a >= b) return 0;
    uint32_t wa = a >> 5, wb = (b - 1) >> 5;
    const uint32_t ma = ~0u << (a & 31), mb = ~0u >> (31 - ((b - 1) & 31));
    if (wa == wb) return __popc(bits[wa] & ma & mb);
    uint32_t n = __popc(bits[wa] & ma) + __popc(bits[wb] & mb);
    for (uint32_t w = wa + 1; w < wb; w++) n += __popc(bits[w]);
    return n;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = block sum.  Two barriers.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t lane, uint32_t wave, uint32_t* total) {
    // wave-level inclusive scan through DPP (four row_shr steps inside the 16-lane rows, then lane 15 / lane 31 broadcast into the rows
    // behind them): six vector instructions instead of six dependent trips through the LDS crossbar (__shfl_up), on the critical path
    // of every block, twice
    uint32_t incl = v;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);  // row_shr:1 (lanes without a source add 0)
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);  // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);  // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);  // row_shr:8  -> inclusive within each row
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1, 3
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2, 3
    __syncthreads();  // s_wave free for reuse
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < BIN_THREADS / 64; w++) {
        const uint32_t t = s_wave[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    *total = tot;
    return wbase + incl - v;
}

struct Cand {
    int gi, tile;
};

// The raw parameters of the Gaussians for bin_count_kernel<true> (the early part of the per-Gaussian forward at its head, dqo_k1_early.h)
struct DqoK1Raw {
    DqoView v;
    const float *means3D, *scales, *rotations, *opacities;
    const int32_t* gobj;
    int32_t *radii_out, *n_touched_out;
};

// tile_objects: DqoObjectGate.tile_objects or NULL — a candidate whose Gaussian's object (the spare word of its xy record) owns no pixel of
// the tile is dropped like one whose footprint cannot reach the tile
//
// K1: the block first runs the early part of the per-Gaussian forward for its own 256 Gaussians (k1_early: the statements of
// preprocess_kernel, which is then not launched) instead of reading their rect / conic / pixel position back from the tables — only for
// a frame whose tile histogram, flags and per-frame scalars the previous frame's dqo_rast_backward_adam has cleared
// (DqoRastCtx.frame_prezeroed: preprocess_kernel is also the launch that zeroes the histogram in front of this kernel's atomics).
template <bool K1>
__global__ __launch_bounds__(BIN_THREADS, BIN_WAVES) void bin_count_kernel(int P, int gx, const int32_t* __restrict__ tile_mask, DqoGeomLayout g,
                                                                uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_flag,
                                                                DqoBinLayout bin, int64_t capacity,
                                                                const unsigned long long* __restrict__ tile_objects, const DqoK1Raw k1,
                                                                const uint8_t* __restrict__ row_flags) {
    __shared__ uint32_t s_off[BIN_CHUNK + 1];  // exclusive prefix of the rect areas
    __shared__ uint2 s_rect[BIN_CHUNK];        // packed tile rects
    __shared__ float4 s_con[BIN_CHUNK];        // conic + opacity
    __shared__ float4 s_xyq[BIN_CHUNK];        // (pix.x, pix.y, q threshold, depth bits)
    __shared__ uint32_t s_cnt[BIN_CHUNK];      // live candidates of each Gaussian
    __shared__ uint32_t s_prev[BIN_CHUNK];     // ... of them in earlier windows (placement sweep)
    __shared__ uint32_t s_gb[BIN_CHUNK];       // exclusive prefix of s_cnt
    __shared__ uint32_t s_bits[BIN_WINDOW / 32];
    __shared__ uint32_t s_wave[BIN_THREADS / 64];
    __shared__ uint32_t s_base;
    __shared__ int s_gobj[BIN_CHUNK];          // object id (object gate only)
    const int tid = threadIdx.x;
    const uint32_t lane = lane_id(), wave = tid >> 6;
    const int chunk0 = blockIdx.x * BIN_CHUNK;
    const int my_idx = dqo_spread_index(chunk0 + tid, P);  // which Gaussian this thread owns (dqo_common.h)

    // ---- this thread's Gaussian: rect area (0 = culled by K1) and the inputs of the footprint test, parked in LDS ----
    uint32_t area = 0;
    {
        uint2 rc = make_uint2(0u, 0u);
        float4 co = make_float4(0.f, 0.f, 0.f, 0.f), xy = co;
        if constexpr (K1) {
            float view[16], proj[16];
#pragma unroll
            for (int i = 0; i < 16; i++) view[i] = k1.v.view[i], proj[i] = k1.v.proj[i];
            if (blockIdx.x == 0 && tid == 0) {
                g.header->stage = 1u;  // (the header still holds the previous frame: "stage 1" until the sort kernels rewrite it)
                // the frame_prezeroed promise: the previous frame's tail leaves a stamp behind its clearing; a frame that finds none — a
                // forward-only render, dqo_rast_backward, an error return in between — is flagged (folded into header.overflow)
                if (g.counters[9] != DQO_CLEARED_STAMP) atomicOr(&g.counters[8], 1u);
                g.counters[9] = 0u;
            }
            bool vis = false;
            uint32_t ncand = 0;
            if (my_idx < P) {
                const K1Early e = k1_early<false>(k1.v, view, proj, 0.f, 0.f, 0.f, my_idx, k1.means3D, k1.scales, k1.rotations, k1.opacities,
                                                  nullptr, nullptr, k1.gobj, g, k1.radii_out, k1.n_touched_out);
                rc = make_uint2((uint32_t)e.rminx | ((uint32_t)e.rmaxx << 16), (uint32_t)e.rminy | ((uint32_t)e.rmaxy << 16));
                co = e.co, xy = e.xy;
                ncand = (uint32_t)((e.rmaxx - e.rminx) * (e.rmaxy - e.rminy));
                area = ncand;
                vis = e.radius > 0;
            }
            // visible count and (Gaussian, tile) pairs in the tile rects (the header's statistics): one pair of atomics per wave
            const uint32_t nv = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(vis));
            const uint32_t nc = dqo_wave_sum_u32(ncand, (int)lane);
            if (lane == 0 && (nv | nc) != 0u) {
                uint32_t* const my_line = g.spread + (size_t)((blockIdx.x * (BIN_THREADS / 64) + wave) % DQO_SPREAD) * 64;
                if (nv) atomicAdd(&my_line[0], nv);
                if (nc) atomicAdd(&my_line[1], nc);
            }
        } else if (my_idx < P) {
            // all three loads together (the tables of a culled Gaussian hold stale values that are never looked at): loading the
            // conic only after the rect says "visible" would put two memory latencies in series at the head of every block
            rc = g.rect16[my_idx];
            co = g.conic_opacity[my_idx];
            xy = g.xy_depth[my_idx];
            area = ((rc.x >> 16) - (rc.x & 0xffffu)) * ((rc.y >> 16) - (rc.y & 0xffffu));
        }
        s_rect[tid] = rc;
        if (area) {
            s_con[tid] = co;
            s_xyq[tid] = make_float4(xy.x, xy.y, dqo_q_threshold(co.w), xy.z);
            s_gobj[tid] = __float_as_int(xy.w);
        }
        s_cnt[tid] = 0;
        s_prev[tid] = 0;
    }
    uint32_t total;
    const uint32_t my_off = block_exclusive_scan(area, s_wave, lane, wave, &total);
    s_off[tid] = my_off;
    if (tid == BIN_THREADS - 1) s_off[BIN_CHUNK] = total;
    __syncthreads();
    if (total == 0) {  // nothing visible in this chunk
        if (my_idx < P) g.tiles_touched[my_idx] = 0, g.slot_base[my_idx] = 0;
        return;
    }

    // candidate w of the block -> (Gaussian, tile)
    auto decode = [&](uint32_t w) {
        int lo = 0, hi = BIN_CHUNK;  // largest gi with s_off[gi] <= w
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_off[mid] <= w) lo = mid;
            else hi = mid;
        }
        const uint32_t r = w - s_off[lo];
        const uint2 rc = s_rect[lo];
        const uint32_t minx = rc.x & 0xffffu, rw = (rc.x >> 16) - minx, miny = rc.y & 0xffffu;
        const uint32_t ry = r / rw, rx = r - ry * rw;
        Cand c;
        c.gi = lo;
        c.tile = (int)((miny + ry) * (uint32_t)gx + minx + rx);
        return c;
    };
    // footprint test of every candidate of the window [win, wend) -> s_bits (and tile_flag for the dead ones)
    auto cull_window = [&](uint32_t win, uint32_t wend, bool flag_dead) {
        for (int i = tid; i < BIN_WINDOW / 32; i += BIN_THREADS) s_bits[i] = 0;
        __syncthreads();
        for (uint32_t w = win + tid; w < wend; w += BIN_THREADS) {
            const Cand c = decode(w);
            if (tile_mask != nullptr && !tile_mask[c.tile]) continue;
            const float4 co = s_con[c.gi], xyq = s_xyq[c.gi];
            const int x = c.tile % gx, y = c.tile / gx;
            bool live = dqo_splat_hits_rect(xyq.x, xyq.y, co.x, co.y, co.z, xyq.z, (float)(x * DQO_TILE), (float)(y * DQO_TILE),
                                            (float)(x * DQO_TILE + DQO_TILE - 1), (float)(y * DQO_TILE + DQO_TILE - 1));
            if (tile_objects != nullptr) live = live && ((tile_objects[c.tile] >> (s_gobj[c.gi] & 63)) & 1ull) != 0ull;
            if (live) {
                atomicOr(&s_bits[(w - win) >> 5], 1u << ((w - win) & 31));
            } else if (flag_dead && tile_flag[c.tile] == 0u) {
                // active in the reference (its list holds this dead entry): render the tile, do not leave the initial fills
                tile_flag[c.tile] = 1u;
            }
        }
        __syncthreads();
    };
    // per-Gaussian live count of the window, added to acc[]
    auto carry_window = [&](uint32_t win, uint32_t wend, uint32_t* acc) {
        const uint32_t a = max(my_off, win), b = min(my_off + area, wend);
        if (a < b) acc[tid] += popcount_range(s_bits, a - win, b - win);
    };

    // ---- sweep 1: live candidates per Gaussian ----
    const bool one_window = total <= (uint32_t)BIN_WINDOW;
    for (uint32_t win = 0; win < total; win += BIN_WINDOW) {
        const uint32_t wend = min(total, win + (uint32_t)BIN_WINDOW);
        cull_window(win, wend, true);
        carry_window(win, wend, s_cnt);
        if (!one_window) __syncthreads();  // s_bits is rebuilt by the next window
    }
    // ---- tiles_touched + gaussian-major slot allocation: block scan of the live counts, one atomic per block ----
    // A FROZEN row (DqoRastInputs.row_flags, bucket mode = the captured mapping iteration) is rendered — its list entries are written
    // below like any other's — but it is no parameter of the mapping call: its instances get NO gradient slots (slot = 0xffffffff in the
    // list record: every slot-indexed write of the backward is skipped, as it is for a slot beyond the capacity), so the backward's
    // blend kernel stores no partial gradient record for it and the per-Gaussian tail has nothing to read.
    const bool no_slots = bin.bucket > 0 && row_flags != nullptr && my_idx < P && (row_flags[my_idx] & DQO_ROW_FROZEN) != 0u;
    const uint32_t my_cnt = no_slots ? 0u : s_cnt[tid];
    __shared__ uint8_t s_noslot[BIN_CHUNK];
    s_noslot[tid] = no_slots ? (uint8_t)1 : (uint8_t)0;
    if (row_flags != nullptr && bin.bucket > 0) {  // (kernel-uniform) the header's num_rendered stays the sum of the LIST lengths
        const uint32_t nf = dqo_wave_sum_u32(no_slots ? s_cnt[tid] : 0u, (int)lane);
        if (lane == 0 && nf != 0u) atomicAdd(&g.spread[(size_t)((blockIdx.x * (BIN_THREADS / 64) + wave) % DQO_SPREAD) * 64 + 5], nf);
    }
    uint32_t block_live;
    const uint32_t my_gb = block_exclusive_scan(my_cnt, s_wave, lane, wave, &block_live);
    s_gb[tid] = my_gb;
    if (tid == 0) {
        if (block_live == 0u) {
            s_base = 0u;
        } else if (bin.bucket <= 0) {
            s_base = atomicAdd(&g.counters[0], block_live);
        } else {
            // Bucket mode (a caller that can re-run a frame): the slot space is cut into DQO_SPREAD regions with an allocator each — the
            // ONE same-address returning atomic of this kernel is otherwise taken by all ~2 000 blocks at about the same time and served
            // one per ~11 ns (same-address atomics serialise memory-side): up to 64 counters on 64 lines, 30 blocks each on cfg 3.  A region that
            // runs out of its share invalidates the frame like running out of the capacity does (counters[7]; nothing is written out of
            // bounds: the block's slots are placed past the capacity, where every slot-indexed write is skipped).
            // (at least 16 blocks per region, so that a region's load stays near the mean — the capacity is ~3x the instances kept:
            // a small map uses fewer regions, down to one)
            const uint32_t regions = min((uint32_t)DQO_SPREAD, max(1u, gridDim.x / 16u));
            const uint32_t r = blockIdx.x % regions, share = (uint32_t)(capacity / regions);
            const uint32_t off = atomicAdd(&g.spread[(size_t)r * 64 + 4], block_live);
            if (off + block_live > share) {
                g.counters[7] = 1u;
                s_base = (uint32_t)capacity;
            } else {
                s_base = r * share + off;
            }
        }
    }
    __syncthreads();
    const uint32_t base = s_base;
    if (my_idx < P) {
        g.tiles_touched[my_idx] = my_cnt;
        g.slot_base[my_idx] = base + my_gb;
    }
    if (block_live == 0) return;
    if (bin.bucket > 0) {
        // bucket mode has no placement pass to clear the validity words of the backward's partial gradient records: the block
        // clears its own contiguous slot range here, coalesced
        for (uint32_t i = tid; i < block_live; i += BIN_THREADS)
            if ((int64_t)(base + i) < capacity) bin.rec_valid[base + i] = 0u;
    }

    // ---- sweep 2: every live candidate takes its rank in its tile and records (tile, rank, Gaussian) at its slot ----
    for (uint32_t win = 0; win < total; win += BIN_WINDOW) {
        const uint32_t wend = min(total, win + (uint32_t)BIN_WINDOW);
        if (!one_window) cull_window(win, wend, false);  // (a single window's bits are still in LDS)
        for (uint32_t w0 = win + tid; w0 < wend; w0 += BIN_THREADS * BIN_FLIGHT) {
            uint32_t slot[BIN_FLIGHT], rank[BIN_FLIGHT];
            int tile[BIN_FLIGHT], gid[BIN_FLIGHT];
            float depth[BIN_FLIGHT];
            bool ok[BIN_FLIGHT];
#pragma unroll
            for (int u = 0; u < BIN_FLIGHT; u++) {
                const uint32_t w = w0 + u * BIN_THREADS;
                ok[u] = w < wend && ((s_bits[(w - win) >> 5] >> ((w - win) & 31)) & 1u);
                if (ok[u]) {
                    const Cand c = decode(w);
                    const uint32_t first = max(s_off[c.gi], win);
                    slot[u] = s_noslot[c.gi] ? 0xffffffffu : base + s_gb[c.gi] + s_prev[c.gi] + popcount_range(s_bits, first - win, w - win);
                    tile[u] = c.tile;
                    gid[u] = dqo_spread_index(chunk0 + c.gi, P);
                    depth[u] = s_xyq[c.gi].w;
                    rank[u] = atomicAdd(&tile_count[(size_t)c.tile * DQO_TSTRIDE], 1u);
                }
            }
#pragma unroll
            for (int u = 0; u < BIN_FLIGHT; u++) {
                if (ok[u] && ((int64_t)slot[u] < capacity || slot[u] == 0xffffffffu)) {
                    if (bin.bucket > 0) {
                        // fixed per-tile buckets: the rank IS the list position (what bin_place_kernel derives from the scanned
                        // ranges in the packed mode); an instance beyond the bucket is dropped — tile_scan_kernel flags the frame
                        if (rank[u] < (uint32_t)bin.bucket) {
                            const size_t pos = (size_t)tile[u] * (size_t)bin.bucket + rank[u];
                            bin.recs[pos] = make_uint4((uint32_t)gid[u], __float_as_uint(depth[u]), slot[u], 0u);
                        }
                    } else {
                        bin.slot_info[slot[u]] = make_uint2((uint32_t)tile[u], rank[u]);
                        bin.slot_gid[slot[u]] = (uint32_t)gid[u];
                    }
                }
            }
        }
        if (!one_window) {
            __syncthreads();  // the loop above still reads s_prev and s_bits
            carry_window(win, wend, s_prev);
            __syncthreads();
        }
    }
}

// One thread per instance slot: list position = start of its tile's range + its rank there.
__global__ __launch_bounds__(256) void bin_place_kernel(DqoGeomLayout g, DqoImageLayout img, DqoBinLayout bin, int64_t capacity) {
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = min((int64_t)g.counters[0], capacity);
    if (slot >= n) return;
    const uint2 info = bin.slot_info[slot];
    const uint32_t gid = bin.slot_gid[slot];
    const uint2 range = img.ranges[info.x];
    const uint32_t pos = range.x + info.y;
    if (pos >= range.y) {  // only when the forward overflowed its capacity (ranges are emptied then)
        bin.rec_valid[slot] = 0u;
        return;
    }
    bin.recs[pos] = make_uint4(gid, __float_as_uint(g.xy_depth[gid].z), (uint32_t)slot, 0u);
    bin.rec_valid[slot] = 0u;  // no partial gradient record of this slot exists yet (backward)
}

}  // namespace

int dqo_launch_bin_count(int P, int gx, const int32_t* tile_mask, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                         int64_t capacity, const unsigned long long* tile_objects, hipStream_t s, const uint8_t* row_flags) {
    DqoK1Raw none{};
    DQO_LAUNCH("bin_count_kernel", bin_count_kernel<false>, dim3(dqo_spread_blocks(P)), dim3(BIN_THREADS), s, P, gx, tile_mask, g,
               img.tile_count, img.tile_flag, bin, capacity, tile_objects, none, row_flags);
    return DQO_OK;
}

// ... with the early part of the per-Gaussian forward at the head of every block (no preprocess_kernel launch in front)
int dqo_launch_bin_count_k1(const DqoView& v, const DqoRastInputs* in, const DqoRastOutputs* out, const int32_t* gobj, const DqoGeomLayout& g,
                            const DqoImageLayout& img, const DqoBinLayout& bin, int64_t capacity, const unsigned long long* tile_objects,
                            hipStream_t s) {
    DqoK1Raw k1;
    k1.v = v, k1.means3D = in->means3D, k1.scales = in->scales, k1.rotations = in->rotations, k1.opacities = in->opacities;
    k1.gobj = gobj, k1.radii_out = out->radii, k1.n_touched_out = out->n_touched;
    DQO_LAUNCH("bin_count_kernel", bin_count_kernel<true>, dim3(dqo_spread_blocks(v.P)), dim3(BIN_THREADS), s, v.P, v.gx, in->tile_mask, g,
               img.tile_count, img.tile_flag, bin, capacity, tile_objects, k1, v.row_flags);
    return DQO_OK;
}

int dqo_launch_bin_place(const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin, int64_t capacity, hipStream_t s) {
    if (capacity <= 0) return DQO_OK;
    DQO_LAUNCH("bin_place_kernel", bin_place_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), s, g, img, bin, capacity);
    return DQO_OK;
}
