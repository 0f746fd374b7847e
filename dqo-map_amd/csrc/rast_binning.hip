// Tile binning for gfx950: per-tile instance counts and the instance emit, both as a load-balanced expansion over the
// candidate (Gaussian, tile) pairs.  Replaces the per-Gaussian tile loops of
//   /root/reference/submodules/diff-gaussian-rasterizer-depth/cuda_rasterizer/forward.cu:344-353 (tiles_touched) and
//   cuda_rasterizer/rasterizer_impl.cu:70-115 (duplicateWithKeys),
// where one thread walks its Gaussian's whole tile rect: a wave then runs as long as its largest splat and every
// iteration waits for its own atomic.  Here a block owns a chunk of 1024 consecutive Gaussians, scans their rect areas in
// LDS and hands out ONE candidate pair per thread-iteration (binary search in the scanned offsets), so lanes are evenly
// loaded and all their atomics are independent and in flight together.
//
// The live / dead decision per candidate is the output-invariant footprint test of dqo_cull.h; it is evaluated with the
// same inputs and IEEE-only arithmetic in both kernels, so count and emit agree bit for bit.  A Gaussian's k-th live tile
// (rect order) gets gaussian-major slot slot_base + k: fixed order => the backward's per-Gaussian sum is reproducible.
#include "dqo_common.h"
#include "dqo_cull.h"

namespace {

constexpr int BIN_THREADS = 256;
constexpr int BIN_ITEMS = 1;
constexpr int BIN_CHUNK = BIN_THREADS * BIN_ITEMS;  // Gaussians per block
constexpr int BIN_WINDOW = 16384;                   // candidate pairs whose live bits fit the LDS bit array at once

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

struct RectD {
    int minx, maxx, miny, maxy;
};
__device__ __forceinline__ RectD unpack_rect(uint2 r) {
    RectD d;
    d.minx = r.x & 0xffff;
    d.maxx = r.x >> 16;
    d.miny = r.y & 0xffff;
    d.maxy = r.y >> 16;
    return d;
}

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

// EMIT = false: count pass (tile histogram, tiles_touched, slot_base, tile_flag).
// EMIT = true : emit pass (keys + slots into the tile segments through the per-tile cursors).
template <bool EMIT>
__global__ __launch_bounds__(BIN_THREADS) void bin_kernel(int P, int gx, const int32_t* __restrict__ tile_mask, DqoGeomLayout g,
                                                          uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_flag,
                                                          uint32_t* __restrict__ tile_cursor, DqoBinLayout bin, int64_t capacity) {
    __shared__ uint32_t s_off[BIN_CHUNK + 1];  // exclusive prefix of the rect areas
    __shared__ uint32_t s_prev[BIN_CHUNK];     // live candidates of each Gaussian in earlier windows
    __shared__ uint32_t s_bits[BIN_WINDOW / 32];
    __shared__ uint32_t s_wave[BIN_THREADS / 64];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x;
    const uint32_t lane = lane_id(), wave = tid >> 6;
    const int chunk0 = blockIdx.x * BIN_CHUNK;

    // ---- rect areas of this block's Gaussians, thread-major (thread t owns items 4t..4t+3), exclusive scan in LDS ----
    uint32_t area[BIN_ITEMS], mine = 0;
#pragma unroll
    for (int it = 0; it < BIN_ITEMS; it++) {
        const int idx = chunk0 + tid * BIN_ITEMS + it;
        uint32_t a = 0;
        if (idx < P) {
            const RectD r = unpack_rect(g.rect16[idx]);
            a = (uint32_t)((r.maxx - r.minx) * (r.maxy - r.miny));
        }
        area[it] = a;
        mine += a;
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < wave; w++) wbase += s_wave[w];
    {
        uint32_t run = wbase + incl - mine;
#pragma unroll
        for (int it = 0; it < BIN_ITEMS; it++) {
            s_off[tid * BIN_ITEMS + it] = run;
            s_prev[tid * BIN_ITEMS + it] = 0;
            run += area[it];
        }
        if (tid == BIN_THREADS - 1) s_off[BIN_CHUNK] = run;
    }
    __syncthreads();
    const uint32_t total = s_off[BIN_CHUNK];

    for (uint32_t win = 0; win < total; win += BIN_WINDOW) {
        const uint32_t wend = min(total, win + (uint32_t)BIN_WINDOW);
        for (int i = tid; i < BIN_WINDOW / 32; i += BIN_THREADS) s_bits[i] = 0;
        __syncthreads();
        // ---- phase B: one candidate pair per thread-iteration: decode, mask + footprint test, count ----
        for (uint32_t w = win + tid; w < wend; w += BIN_THREADS) {
            int lo = 0, hi = BIN_CHUNK;  // largest gi with s_off[gi] <= w
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_off[mid] <= w) lo = mid;
                else hi = mid;
            }
            const int gi = lo, idx = chunk0 + gi;
            const uint32_t r = w - s_off[gi];
            const RectD rc = unpack_rect(g.rect16[idx]);
            const int rw = rc.maxx - rc.minx;
            const int x = rc.minx + (int)(r % (uint32_t)rw), y = rc.miny + (int)(r / (uint32_t)rw);
            const int t = y * gx + x;
            if (tile_mask != nullptr && !tile_mask[t]) continue;
            const float4 xyd = g.xy_depth[idx];
            const float4 co = g.conic_opacity[idx];
            const bool live = dqo_splat_hits_rect(xyd.x, xyd.y, co.x, co.y, co.z, dqo_q_threshold(co.w), (float)(x * DQO_TILE),
                                                  (float)(y * DQO_TILE), (float)(x * DQO_TILE + DQO_TILE - 1),
                                                  (float)(y * DQO_TILE + DQO_TILE - 1));
            if (live) {
                atomicOr(&s_bits[(w - win) >> 5], 1u << ((w - win) & 31));
                if (!EMIT) atomicAdd(&tile_count[(size_t)t * DQO_TSTRIDE], 1u);
            } else if (!EMIT && tile_flag[t] == 0u) {
                // active in the reference (its list holds this dead entry): render the tile, do not leave the initial fills
                tile_flag[t] = 1u;
            }
        }
        __syncthreads();
        if (EMIT) {
            // ---- phase C: live candidates take a position in their tile segment; slot = slot_base + rank within the Gaussian ----
            for (uint32_t w = win + tid; w < wend; w += BIN_THREADS) {
                if (!((s_bits[(w - win) >> 5] >> ((w - win) & 31)) & 1u)) continue;
                int lo = 0, hi = BIN_CHUNK;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_off[mid] <= w) lo = mid;
                    else hi = mid;
                }
                const int gi = lo, idx = chunk0 + gi;
                const uint32_t r = w - s_off[gi];
                const RectD rc = unpack_rect(g.rect16[idx]);
                const int rw = rc.maxx - rc.minx;
                const int x = rc.minx + (int)(r % (uint32_t)rw), y = rc.miny + (int)(r / (uint32_t)rw);
                const int t = y * gx + x;
                const uint32_t first = max(s_off[gi], win);
                const uint32_t rank = s_prev[gi] + popcount_range(s_bits, first - win, w - win);
                const uint32_t pos = atomicAdd(&tile_cursor[(size_t)t * DQO_TSTRIDE], 1u);
                if ((int64_t)pos < capacity) {
                    bin.keys[pos] = ((uint64_t)__float_as_uint(g.xy_depth[idx].z) << 32) | (uint32_t)idx;
                    bin.slots[pos] = g.slot_base[idx] + rank;
                }
            }
        }
        __syncthreads();  // phase C still reads s_prev
        // ---- phase D: carry each Gaussian's live count of this window ----
        for (int gi = tid; gi < BIN_CHUNK; gi += BIN_THREADS) {
            const uint32_t a = max(s_off[gi], win), b = min(s_off[gi + 1], wend);
            if (a < b) s_prev[gi] += popcount_range(s_bits, a - win, b - win);
        }
        __syncthreads();
    }
    if (EMIT) return;

    // ---- tiles_touched + gaussian-major slot allocation: block scan of the live counts, one atomic per block ----
    uint32_t cnt[BIN_ITEMS], my2 = 0;
#pragma unroll
    for (int it = 0; it < BIN_ITEMS; it++) {
        cnt[it] = s_prev[tid * BIN_ITEMS + it];
        my2 += cnt[it];
    }
    uint32_t inc2 = my2;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc2, off);
        if (lane >= (uint32_t)off) inc2 += o;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc2;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < BIN_THREADS / 64; w++) {
            const uint32_t t = s_wave[w];
            s_wave[w] = tot;
            tot += t;
        }
        s_base = tot ? atomicAdd(&g.counters[0], tot) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base + s_wave[wave] + (inc2 - my2);
#pragma unroll
    for (int it = 0; it < BIN_ITEMS; it++) {
        const int idx = chunk0 + tid * BIN_ITEMS + it;
        if (idx < P) {
            g.tiles_touched[idx] = cnt[it];
            g.slot_base[idx] = base;
        }
        base += cnt[it];
    }
}

}  // namespace

int dqo_launch_bin_count(int P, int gx, const int32_t* tile_mask, const DqoGeomLayout& g, const DqoImageLayout& img, hipStream_t s) {
    DqoBinLayout none = {};
    DQO_LAUNCH("bin_count_kernel", bin_kernel<false>, dim3((P + BIN_CHUNK - 1) / BIN_CHUNK), dim3(BIN_THREADS), s, P, gx, tile_mask, g,
               img.tile_count, img.tile_flag, img.tile_cursor, none, (int64_t)0);
    return DQO_OK;
}

int dqo_launch_bin_emit(int P, int gx, const int32_t* tile_mask, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                        int64_t capacity, hipStream_t s) {
    DQO_LAUNCH("bin_emit_kernel", bin_kernel<true>, dim3((P + BIN_CHUNK - 1) / BIN_CHUNK), dim3(BIN_THREADS), s, P, gx, tile_mask, g,
               img.tile_count, img.tile_flag, img.tile_cursor, bin, capacity);
    return DQO_OK;
}
