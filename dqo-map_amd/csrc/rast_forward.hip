// Forward pass of the depth-aware Gaussian rasteriser for gfx950 (MI355X).
//
// Replaces (semantics, not structure) /root/reference/submodules/diff-gaussian-rasterizer-depth/
//   cuda_rasterizer/forward.cu:238-354   preprocessCUDA            -> preprocess_kernel (its statements: dqo_k1_early.h / dqo_k1_late.h; a replayed
//                                                                      iteration runs the early part at the head of bin_count_kernel<true>)
//   cuda_rasterizer/rasterizer_impl.cu:70-142, 303-365 (cub scan, duplicateWithKeys, cub radix sort,
//   identifyTileRanges, host tile compaction with two D2H syncs)   -> bin_count_kernel / bin_place_kernel (rast_binning.hip),
//                                                                      tile_scan_kernel, tile_sort_wave_kernel, tile_sort_kernel
//   cuda_rasterizer/forward.cu:636-866   renderCUDA_withMask       -> blend_forward_kernel (rast_forward_blend.hip)
//
// MI355X design: no host synchronisation anywhere; binning is a per-tile counting sort (footprint test +
// atomic tile histogram with ranks -> one-block scan -> atomic-free placement) followed by a register-resident sort of each
// tile's (depth, id) keys by one wave, so an instance crosses HBM once as an 8-byte key instead of 6 radix passes over
// 12-byte pairs; everything the blend loop needs per Gaussian is precomputed once into 16-byte SoA records (the reference
// rebuilds the quaternion rotation and does four uncoalesced gathers per (pixel, Gaussian) pair, forward.cu:779-791).
#include <cstdlib>

#include "dqo_common.h"
#include "dqo_cull.h"
#include "dqo_gauss_chain.h"
#include "dqo_k1_early.h"

#ifndef K1_WAVES
#define K1_WAVES 4     // preprocess_kernel: waves per SIMD the register allocation leaves room for
#endif
#ifndef SORTW_WAVES
#define SORTW_WAVES 7  // tile_sort_wave_kernel
#endif
#ifndef SORTW_LATE_WAVES
#define SORTW_LATE_WAVES 4  // tile_sort_wave_kernel<true>: + the late part of the per-Gaussian forward in its extra blocks (106 VGPRs)
#endif
namespace {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ------------------------------------------------------------------------------------------------------------------
// K1: per-Gaussian preprocess.  Bit-faithful to the oracle: IEEE ops only, no FMA contraction in this kernel (it is
// bandwidth bound; contraction would only move discrete decisions such as ceil(radius) and the tile rect).
// ------------------------------------------------------------------------------------------------------------------
constexpr int K1_THREADS = 256;
constexpr int K1_ITEMS = 1;

// LATE: the late part runs here (one kernel); !LATE: it runs in the shadow of the per-tile sorts (tile_sort_wave_kernel<true>)
template <bool LATE>
__global__ __launch_bounds__(K1_THREADS, K1_WAVES) void preprocess_kernel(const DqoView v, const float* __restrict__ means3D,
                                                                const float* __restrict__ scales,
                                                                const float* __restrict__ rotations,
                                                                const float* __restrict__ opacities,
                                                                const float* __restrict__ shs,
                                                                const float* __restrict__ colors_precomp,
                                                                const int32_t* __restrict__ tile_mask, DqoGeomLayout g,
                                                                int32_t* __restrict__ radii_out,
                                                                int32_t* __restrict__ n_touched_out, uint32_t* __restrict__ zero_base,
                                                                size_t zero_words, const int32_t* __restrict__ gobj,
                                                                const int check_prezeroed) {
#pragma clang fp contract(off)
    __shared__ uint32_t s_visible;
    const int tid = threadIdx.x;
    const int P = v.P;
    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) view[i] = v.view[i], proj[i] = v.proj[i];
    const float cam0 = v.campos[0], cam1 = v.campos[1], cam2 = v.campos[2];
    __shared__ uint32_t s_cand;
    if (tid == 0) s_visible = 0, s_cand = 0;
    // (a frame_prezeroed frame still holds the previous frame's header: mark it "stage 1" until the sort kernels rewrite it)
    if (blockIdx.x == 0 && tid == 0) g.header->stage = 1u;
    // DqoRastCtx.frame_prezeroed is a promise of the caller (the previous frame on this ctx ended in dqo_rast_backward_adam, whose tail
    // clears the per-frame scalars).  A broken promise — a forward-only render, dqo_rast_backward, an error return in between — would
    // give wrong slot bases and doubled statistics without a sign: the first block looks at the words no kernel of THIS frame has
    // touched yet (slot allocators, queue counters, loss-tap sums; not words 0, 1 of the lines, which this launch is adding to) and
    // raises counters[8], which both header writers fold into header.overflow.
    if (check_prezeroed && blockIdx.x == 0 && tid < DQO_SPREAD) {
        const uint32_t* line = g.spread + (size_t)tid * 64;
        uint32_t bad = line[2] | line[3] | line[4] | line[5];
#pragma unroll
        for (int i = 8; i < 16; i++) bad |= line[i];
        if (tid < 8) bad |= g.counters[tid];
        if (bad != 0u) atomicOr(&g.counters[8], 1u);
    }
    // this launch also zeroes the tile histogram + tile flags for bin_count_kernel (one contiguous range, a slice per block)
    {
        const size_t per = (zero_words + gridDim.x - 1) / gridDim.x;
        const size_t z0 = (size_t)blockIdx.x * per, z1 = min(zero_words, z0 + per);
        for (size_t i = z0 + tid; i < z1; i += K1_THREADS) zero_base[i] = 0u;
    }
    __syncthreads();

    uint32_t nvis = 0, ncand = 0;
#pragma unroll 1
    for (int it = 0; it < K1_ITEMS; it++) {
        const int idx = blockIdx.x * (K1_THREADS * K1_ITEMS) + it * K1_THREADS + tid;
        if (idx >= P) continue;
        const K1Early e = k1_early<LATE>(v, view, proj, cam0, cam1, cam2, idx, means3D, scales, rotations, opacities, shs, colors_precomp, gobj, g,
                                         radii_out, n_touched_out);
        const int radius = e.radius, rminx = e.rminx, rminy = e.rminy, rmaxx = e.rmaxx, rmaxy = e.rmaxy;
        nvis += radius > 0 ? 1u : 0u;
        ncand += (uint32_t)((rmaxx - rminx) * (rmaxy - rminy));
    }
    // visible count (statistics) and the number of (Gaussian, tile) pairs in the tile rects — the reference's num_rendered
    // (rasterizer_impl.cu:303-309) and an upper bound of the instances the binning keeps
    if (nvis) atomicAdd(&s_visible, nvis);
    if (ncand) atomicAdd(&s_cand, ncand);
    __syncthreads();
    uint32_t* const my_line = g.spread + (size_t)(blockIdx.x % DQO_SPREAD) * 64;
    if (tid == 0 && s_visible) atomicAdd(&my_line[0], s_visible);
    if (tid == 0 && s_cand) atomicAdd(&my_line[1], s_cand);
}

// ------------------------------------------------------------------------------------------------------------------
// One block: exclusive scan over the per-tile histogram (list ranges), launch order of the tiles, header.
// Replaces cub::DeviceScan + identifyTileRanges + the host compaction loop (rasterizer_impl.cu:303, 338-365).
// ------------------------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 1024;
constexpr int LPT_BUCKETS = 256;
constexpr int SCAN_CACHE = 8192;

__global__ __launch_bounds__(SCAN_THREADS) void tile_scan_kernel(int T, DqoImageLayout img, DqoGeomLayout g, int64_t capacity, int bucket) {
    __shared__ uint32_t s_part[SCAN_THREADS / 64];
    __shared__ uint32_t s_carry, s_max;
    __shared__ uint32_t s_bucket[8 * (LPT_BUCKETS + 1)];
    __shared__ uint32_t s_seg_start[8], s_seg_len[8], s_free_pre[8], s_over_pre[8], s_n_empty;
    __shared__ uint32_t s_stat[2];
    __shared__ uint32_t s_tc[SCAN_CACHE];
    __shared__ uint32_t s_st[SCAN_CACHE];  // exclusive prefix of the list lengths
    const int tid = threadIdx.x;
    const uint32_t lane = lane_id(), wave = tid >> 6;
    if (tid == 0) s_carry = 0, s_max = 0;
    if (tid < 64) {  // totals of K1's spread statistics counters
        uint32_t nv = 0, nc = 0;
        for (int j = tid; j < DQO_SPREAD; j += 64) nv += g.spread[(size_t)j * 64], nc += g.spread[(size_t)j * 64 + 1];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nv += __shfl_xor((int)nv, off), nc += __shfl_xor((int)nc, off);
        if (tid == 0) s_stat[0] = nv, s_stat[1] = nc;
    }
    // instance total of bin_count_kernel's slot allocator (bucket mode: one allocator per region, an exhausted one raises counters[7])
    bool overflow = (bucket > 0 ? g.counters[7] != 0u : (int64_t)g.counters[0] > capacity) || g.counters[8] != 0u;
    // the padded histogram is read from HBM once; the passes below work on an LDS copy (images up to ~2M pixels)
    const bool cached = T <= SCAN_CACHE;
    if (cached)
        for (int t = tid; t < T; t += SCAN_THREADS) s_tc[t] = img.tile_count[(size_t)t * DQO_TSTRIDE];
    __syncthreads();
    auto tcount = [&](int t) { return cached ? s_tc[t] : img.tile_count[(size_t)t * DQO_TSTRIDE]; };
    if (bucket > 0) {  // fixed per-tile buckets: a tile that outgrew its bucket invalidates the frame like a capacity overflow
        uint32_t mx0 = 0;
        for (int t = tid; t < T; t += SCAN_THREADS) mx0 = max(mx0, tcount(t));
        if (mx0) atomicMax(&s_max, mx0);
        __syncthreads();
        overflow = overflow || s_max > (uint32_t)bucket;
    }

    // ---- ranges: chunked block scan ----
    uint32_t local_max = 0;
    for (int base = 0; base < T; base += SCAN_THREADS) {
        const int t = base + tid;
        const uint32_t c = t < T ? tcount(t) : 0u;
        local_max = max(local_max, c);
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += o;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; w++) wbase += s_part[w];
        const uint32_t start = s_carry + wbase + incl - c;
        if (t < T) {
            // More instances than the binning buffer holds: the slot tables are incomplete, so every list is emptied (the
            // frame is invalid and flagged as such in the header; nothing indexes past a buffer).
            const bool keep = c != 0u && !overflow;
            const uint32_t first = bucket > 0 ? (uint32_t)t * (uint32_t)bucket : start;  // bucket mode: the list sits in the tile's bucket
            img.ranges[t] = make_uint2(keep ? first : 0u, keep ? first + c : 0u);  // empty tiles keep (0,0): rasterizer_impl.cu:338
            if (cached) s_st[t] = start;
            else img.tile_cursor[(size_t)t * DQO_TSTRIDE] = start;
        }
        __syncthreads();
        if (tid == SCAN_THREADS - 1) s_carry = start + c;
    }
    if (local_max) atomicMax(&s_max, local_max);
    // ---- launch order of the per-tile kernels (tile_order): XCD-affine and longest-processing-time-first ----
    // Blocks b and b + 8 share an XCD and its 4 MB L2 (MI355X_MICROARCH.md, workgroup dispatch).  The tiles are cut, in
    // row-major order, into 8 bands of equal total list length (equal blend work); band x goes to the blocks with
    // b % 8 == x, so the Gaussian records an XCD gathers belong to one image band (~1/8 of the visible Gaussians: fits its L2)
    // instead of the whole frame.  Inside a band: descending list length (256 levels, LDS counting sort; empty tiles — they
    // still get blocks, which write the reference's initial fills — last).  tile_order is [8][T8]; a band with more than T8
    // tiles hands its smallest ones to the free slots of shorter bands.  Only the block -> tile mapping depends on any of
    // this, no result does.
    constexpr int NSEG = 8, SEGB = LPT_BUCKETS + 1;
    const int T8 = (T + NSEG - 1) / NSEG;
    for (int i = tid; i < NSEG * SEGB; i += SCAN_THREADS) s_bucket[i] = 0;
    for (int i = tid; i < NSEG * T8; i += SCAN_THREADS) img.tile_order[i] = 0xffffffffu, img.slot_info[i] = make_uint4(0xffffffffu, 0u, 0u, 0u);
    __syncthreads();  // (also: img.ranges of this launch are visible to the block from here on)
    const uint32_t mx = s_max, total = s_carry;
    int shift = 0;
    while ((mx >> shift) >= (uint32_t)LPT_BUCKETS) shift++;
    auto key_of = [&](int t) {
        const uint32_t c = tcount(t);
        // band from the exclusive prefix of the list lengths; row-major position for frames without instances
        const uint64_t pos = total ? (uint64_t)(cached ? s_st[t] : img.tile_cursor[(size_t)t * DQO_TSTRIDE]) : (uint64_t)t;
        const uint64_t den = total ? (uint64_t)total : (uint64_t)T;
        const int seg = (int)min((uint64_t)(NSEG - 1), pos * NSEG / den);
        return seg * SEGB + (c ? LPT_BUCKETS - 1 - (int)(c >> shift) : LPT_BUCKETS);
    };
    for (int t = tid; t < T; t += SCAN_THREADS) atomicAdd(&s_bucket[key_of(t)], 1u);
    __syncthreads();
    {  // exclusive scan of the NSEG * SEGB bucket sizes by the whole block (3 consecutive buckets per thread)
        constexpr int PER = (NSEG * SEGB + SCAN_THREADS - 1) / SCAN_THREADS;
        uint32_t cnt[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid * PER + k;
            cnt[k] = i < NSEG * SEGB ? s_bucket[i] : 0u;
            sum += cnt[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += o;
        }
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wave; w++) run += s_part[w];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid * PER + k;
            if (i < NSEG * SEGB) s_bucket[i] = run;
            run += cnt[k];
        }
    }
    __syncthreads();
    if (tid == 0) {  // band starts / lengths, empty-tile count, free-slot and overflow prefixes
        uint32_t n_empty = 0, fre = 0, ovf = 0;
        for (int x = 0; x < NSEG; x++) {
            const uint32_t st = s_bucket[x * SEGB], en = x + 1 < NSEG ? s_bucket[(x + 1) * SEGB] : (uint32_t)T;
            const uint32_t emp_start = s_bucket[x * SEGB + LPT_BUCKETS];
            n_empty += en - emp_start;
            s_seg_start[x] = st;
            s_seg_len[x] = en - st;
            s_free_pre[x] = fre;
            s_over_pre[x] = ovf;
            fre += (en - st) < (uint32_t)T8 ? (uint32_t)T8 - (en - st) : 0u;
            ovf += (en - st) > (uint32_t)T8 ? (en - st) - (uint32_t)T8 : 0u;
        }
        s_n_empty = n_empty;
    }
    __syncthreads();
    for (int t = tid; t < T; t += SCAN_THREADS) {
        const int key = key_of(t), seg = key / SEGB;
        const uint32_t r = atomicAdd(&s_bucket[key], 1u) - s_seg_start[seg];  // rank inside the band, longest lists first
        uint32_t slot;
        if (r < (uint32_t)T8) {
            slot = (uint32_t)seg * T8 + r;
        } else {  // the band is longer than T8: its o-th surplus tile takes the o-th free slot of the shorter bands
            const uint32_t o = s_over_pre[seg] + (r - (uint32_t)T8);
            int y = 0;
            while (y + 1 < NSEG && s_free_pre[y + 1] <= o) y++;
            slot = (uint32_t)y * T8 + s_seg_len[y] + (o - s_free_pre[y]);
        }
        img.tile_order[slot] = (uint32_t)t;
        // (DqoImageLayout.slot_info: tile_sort_wave_kernel writes it again for the frames it runs in; a frame without Gaussians has
        // no sort launch, and its blend kernel still reads its slots)
        const uint2 rg_t = img.ranges[t];
        img.slot_info[slot] = make_uint4((uint32_t)t, rg_t.x, rg_t.y, 0u);
    }
    const uint32_t n_empty = s_n_empty;
    if (tid == 0) {
        DqoRastHeader h;
        h.num_rendered = s_carry;
        h.num_tiles = (uint32_t)T - n_empty;
        // (`overflow` covers the slot allocator running past the capacity as well: whenever the lists were emptied the frame says so)
        h.overflow = (overflow || (int64_t)s_carry > capacity || (bucket > 0 && s_max > (uint32_t)bucket)) ? 1u : 0u;
        h.max_tile_count = s_max;
        h.num_visible = s_stat[0];
        h.num_candidates = s_stat[1];
        h.stage = 2u, h.reserved = 0u;
        *g.header = h;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-tile sort of (depth, id) keys with the slot as payload (replaces the device-wide cub radix sort of
// rasterizer_impl.cu:316-330; ties in depth fall back to the Gaussian id = the reference's stable order).
//   tile_sort_wave_kernel : lists up to 1024 entries (in practice all of them): ONE WAVE per tile, four tiles per block,
//       the whole list in registers (E = 2..16 elements per lane, blocked layout).  Bitonic network: the steps with
//       distance < E are compare-exchanges between a lane's own registers, the others exchange with lane ^ (distance / E)
//       through DPP moves and the gfx950 row / half swaps (lane_xor_value).  No LDS round trips, no barriers: the sort is
//       latency-bound (a few hundred thousand keys in all), so the serial chain per tile is what counts.
//   tile_sort_kernel : longer lists, one block per tile: registers + LDS + (beyond 4096 entries) in-place steps in
//       global memory (correctness path, not a fast path).
// ------------------------------------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 512;   // tile_sort_kernel: eight waves = the eight 512-key runs of a 4096-key segment in one round
constexpr int SORT_GRID = 768;      // its persistent grid: 256 CUs x 3 blocks of 48 KB LDS
#ifndef DQO_SORTW_CAP
#define DQO_SORTW_CAP 1024
#endif
constexpr int SORTW_CAP = DQO_SORTW_CAP;

// The value of lane ^ D, D a power of two, without the LDS crossbar (ds_bpermute costs 20-60 cycles per dependent use and shares
// the LDS pipe of the CU; DPP moves 3-5, the gfx950 row / half swaps 4-9: tools/ubench_valu.hip):
//   D = 1, 2 : DPP quad_perm;  D = 4 : row_shl:4 / row_shr:4 on alternate banks;  D = 8 : row_ror:8
//   D = 16, 32 : v_permlane16/32_swap of the register with a copy of itself leaves (own rows, partner rows) in the two results
template <int D>
__device__ __forceinline__ uint32_t lane_xor_value(uint32_t x, int lane) { return dqo_lane_xor<D>(x, lane); }  // (dqo_common.h)

// one cross-lane step of the network: every register r meets register r of lane ^ D
template <int E, int D>
__device__ __forceinline__ void wave_bitonic_cross(uint64_t (&key)[E], uint32_t (&val)[E], int lane, int k) {
    const bool up = ((lane * E) & k) == 0;  // ascending sub-sequence (k >= 2E here, so the bit is a lane bit)
    const bool keep_min = (((lane & D) == 0) == up);
    // all of the step's exchanges are issued before any result is used; the selects are branch-free
    uint64_t ok[E];
    uint32_t ov[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint32_t olo = lane_xor_value<D>((uint32_t)key[r], lane);
        const uint32_t ohi = lane_xor_value<D>((uint32_t)(key[r] >> 32), lane);
        ov[r] = lane_xor_value<D>(val[r], lane);
        ok[r] = ((uint64_t)ohi << 32) | olo;
    }
#pragma unroll
    for (int r = 0; r < E; r++) {
        const bool take = (ok[r] < key[r]) == keep_min;  // keys are distinct (padding carries identical payloads)
        key[r] = take ? ok[r] : key[r];
        val[r] = take ? ov[r] : val[r];
    }
}

// one merge phase of the bitonic network: the steps j = k/2 ... 1 for sub-sequences of length k
template <int E>
__device__ __forceinline__ void wave_bitonic_phase(uint64_t (&key)[E], uint32_t (&val)[E], int lane, int k) {
    {
#pragma unroll 1
        for (int j = min(k >> 1, 32 * E); j >= E; j >>= 1) {
            // partner element lives in lane ^ d, same register (d stays a run-time value: one copy of each distance's code)
            switch (j / E) {
                case 1: wave_bitonic_cross<E, 1>(key, val, lane, k); break;
                case 2: wave_bitonic_cross<E, 2>(key, val, lane, k); break;
                case 4: wave_bitonic_cross<E, 4>(key, val, lane, k); break;
                case 8: wave_bitonic_cross<E, 8>(key, val, lane, k); break;
                case 16: wave_bitonic_cross<E, 16>(key, val, lane, k); break;
                default: wave_bitonic_cross<E, 32>(key, val, lane, k); break;
            }
        }
#pragma unroll
        for (int j = E >> 1; j > 0; j >>= 1) {
            if (j < k) {
                // both elements in this lane's registers r and r | j
#pragma unroll
                for (int r = 0; r < E; r++) {
                    if ((r & j) == 0) {
                        const int p = r | j;
                        const bool up = (((lane * E + r) & k) == 0);
                        const bool sw = (key[r] > key[p]) == up;
                        const uint64_t kr = key[r], kp = key[p];
                        const uint32_t vr = val[r], vp = val[p];
                        key[r] = sw ? kp : kr;
                        key[p] = sw ? kr : kp;
                        val[r] = sw ? vp : vr;
                        val[p] = sw ? vr : vp;
                    }
                }
            }
        }
    }
}

template <int E>
__device__ __forceinline__ void wave_bitonic(uint64_t (&key)[E], uint32_t (&val)[E], int lane) {
    // k and the cross-lane distances stay run-time loop variables: only the register pairings of the in-lane steps are
    // unrolled, which keeps the kernel's instantiations inside the instruction cache.
#pragma unroll 1
    for (int k = 2; k <= 64 * E; k <<= 1) wave_bitonic_phase<E>(key, val, lane, k);
}

template <int E>
__device__ __forceinline__ void wave_sort_tile(const DqoBinLayout& bin, uint32_t base, int n, int lane) {
    uint64_t key[E];
    uint32_t val[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int i = lane * E + r;
        const uint4 e = i < n ? bin.recs[base + i] : make_uint4(~0u, ~0u, 0u, 0u);  // padding sorts behind every real key
        key[r] = ((uint64_t)e.y << 32) | e.x;                                        // (depth bits of a finite float)
        val[r] = e.z;
    }
    wave_bitonic<E>(key, val, lane);
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int i = lane * E + r;
        if (i < n) {
            bin.point_list[base + i] = (uint32_t)key[r];
            bin.slot_list[base + i] = val[r];
        }
    }
}

// Lists of 513..1024 entries: TWO waves.  Each sorts one half of the list in registers (E = 8), the halves meet once in LDS —
// element i of the lower run against element 511 - i of the upper run, the lower wave keeps the smaller, the upper wave the
// larger: both halves are then bitonic — and every wave finishes with the nine merge steps of its own half.  The serial chain
// of the longest lists (which is what the kernel's duration is) is that of a 512-element sort + one exchange + one merge
// phase instead of a 1024-element sort.
constexpr int SORTP_E = 8, SORTP_RUN = 64 * SORTP_E;
__device__ __forceinline__ void pair_sort_tile(const DqoBinLayout& bin, uint32_t base, int n, int lane, int wave, uint64_t* s_key,
                                               uint32_t* s_val) {
    constexpr int E = SORTP_E;
    uint64_t key[E];
    uint32_t val[E];
    const int run0 = wave * SORTP_RUN;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int i = run0 + lane * E + r;
        const uint4 e = i < n ? bin.recs[base + i] : make_uint4(~0u, ~0u, 0u, 0u);  // padding sorts behind every real key
        key[r] = ((uint64_t)e.y << 32) | e.x;
        val[r] = e.z;
    }
    wave_bitonic<E>(key, val, lane);
#pragma unroll
    for (int r = 0; r < E; r++) s_key[run0 + lane * E + r] = key[r], s_val[run0 + lane * E + r] = val[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int o = (1 - wave) * SORTP_RUN + (SORTP_RUN - 1 - (lane * E + r));  // mirrored element of the other run
        const uint64_t ok = s_key[o];
        const uint32_t ov = s_val[o];
        const bool take = (ok < key[r]) == (wave == 0);  // (equal keys only among the padding: identical payloads)
        key[r] = take ? ok : key[r];
        val[r] = take ? ov : val[r];
    }
    wave_bitonic_phase<E>(key, val, lane, 2 * SORTP_RUN);  // ascending merge of this wave's (bitonic) half
#pragma unroll
    for (int r = 0; r < E; r++) {
        const int i = run0 + lane * E + r;
        if (i < n) {
            bin.point_list[base + i] = (uint32_t)key[r];
            bin.slot_list[base + i] = val[r];
        }
    }
}

// one block (two waves) per tile slot; the second wave only works on lists longer than 512 entries
constexpr int SORTW_THREADS = 128;

template <bool LATE>
__global__ __launch_bounds__(SORTW_THREADS, LATE ? SORTW_LATE_WAVES : SORTW_WAVES) void tile_sort_wave_kernel(int T, DqoImageLayout img, DqoBinLayout bin, DqoGeomLayout g,
                                                                       int64_t capacity, int keep_order, int list_split, const DqoK1Late late) {
    if constexpr (LATE) {
        if ((int)blockIdx.x >= late.first_block) {  // (block-uniform; the sort blocks come first: the longest lists start at once)
            k1_late_block<SORTW_THREADS>(late, g, (int)blockIdx.x - late.first_block);
            return;
        }
    }
    __shared__ uint64_t s_key[2 * SORTP_RUN];
    __shared__ uint32_t s_val[2 * SORTP_RUN];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ti = blockIdx.x;
    // the threshold the queue is built with, for the backward: its split kernel leaves to the queue exactly the lists that are in it
    // (whatever the caller's backward context says)
    if (ti == 0 && threadIdx.x == 0) g.counters[6] = (uint32_t)list_split;
    const uint32_t tile = img.tile_order[ti];  // [8][T8] slots, unused ones hold ~0
    if (tile >= (uint32_t)T) {
        if (threadIdx.x == 0) img.slot_info[ti] = make_uint4(0xffffffffu, 0u, 0u, 0u);
        return;
    }
    uint2 rg;
    if (keep_order) {
        const uint32_t c = img.tile_count[(size_t)tile * DQO_TSTRIDE];
        // the slot tables are incomplete when the instance capacity ran out: every list is emptied, as tile_scan_kernel does; a
        // list that outgrew its bucket is cut at the bucket (bin_count_kernel dropped the rest); both invalidate the frame
        const bool lost = g.counters[7] != 0u;  // (bucket mode: a slot region ran out of its share, bin_count_kernel)
        const uint32_t n_keep = lost ? 0u : min(c, (uint32_t)bin.bucket);
        const uint32_t first = tile * (uint32_t)bin.bucket;
        rg = make_uint2(n_keep ? first : 0u, n_keep ? first + n_keep : 0u);
        if (threadIdx.x == 0) {
            img.ranges[tile] = rg;
            if (c) {
                uint32_t* const line = g.spread + (size_t)(ti % DQO_SPREAD) * 64;
                atomicMax(&line[2], c);
                atomicAdd(&line[3], 1u);
            }
        }
    } else {
        rg = img.ranges[tile];
    }
    if (threadIdx.x == 0) img.slot_info[ti] = make_uint4(tile, rg.x, rg.y, 0u);  // (DqoImageLayout.slot_info: the blend kernels' one-round head)
    const int n = (int)(rg.y - rg.x);
    if (n <= 0) return;
    // DqoRastCtx.list_split: the blend kernels' queue of lists shared between eight waves (longest first, like the one below)
    if (list_split > 0 && n > list_split && threadIdx.x == 0) img.split_tiles[atomicAdd(&g.counters[4], 1u)] = tile;
    if (n > SORTW_CAP) {  // tile_sort_kernel's: queued (the blocks run longest list first, so the queue is close to that order too)
        if (threadIdx.x == 0) img.long_tiles[atomicAdd(&g.counters[1], 1u)] = tile;
        return;
    }
    if (n > SORTP_RUN) {
        pair_sort_tile(bin, rg.x, n, lane, wave, s_key, s_val);
        return;
    }
    if (wave != 0) return;
    if (n <= 64) wave_sort_tile<1>(bin, rg.x, n, lane);
    else if (n <= 128) wave_sort_tile<2>(bin, rg.x, n, lane);
    else if (n <= 256) wave_sort_tile<4>(bin, rg.x, n, lane);
    else wave_sort_tile<8>(bin, rg.x, n, lane);
}

// Lists longer than SORTW_CAP: queued by tile_sort_wave_kernel (img.long_tiles, geom counters[1]) and sorted by a persistent grid of
// 512-thread blocks, one tile at a time per block, through three levels of the same ascending network.
//   registers : runs of 512 keys, one wave each (wave_bitonic<8>): every compare-exchange at distance < 512
//   LDS       : segments of SORTL_SEG = 4096 keys: the merge steps at distance 512 .. 2048 (flip step + half cleaners), after which
//               each run finishes in registers again (wave_bitonic_phase)
//   global    : lists longer than a segment: the merge steps at distance >= 4096 in place in the tile's key / slot arrays (one block
//               owns the tile; __syncthreads orders its global accesses), then every segment goes through LDS + registers again
// Entries past the end of the list behave as +infinity: the network only uses ascending comparators, so they never move and need
// no storage.  (The first version ran the whole network in LDS with 256 threads and a barrier per step — 78 steps for 4096 keys:
// 193 us on config 4, 515 us on config 5, where 40 % of the tiles are longer than 1024.)
constexpr int SORTL_SEG = 4096;
constexpr int SORTL_RUN = SORTP_RUN;  // 512

// keep_order (DqoRastCtx.keep_tile_order, bucket mode): no tile_scan_kernel ran for this frame.  tile_order is the one an earlier
// frame left in the image buffer (any permutation of the tiles gives the same results), a list's range follows from its own
// counter, and the frame statistics the header needs go to the spread lines (words 2..3), which tile_sort_kernel's first block
// sums up (header_from_spread).
template <bool LATE>
__global__ __launch_bounds__(SORT_THREADS, LATE ? 4 : 6) void tile_sort_kernel(int T, DqoImageLayout img, DqoBinLayout bin, DqoGeomLayout g, int64_t capacity,
                                                                 int keep_order, const DqoK1Late late) {
    if constexpr (LATE) {  // (the late part of the per-Gaussian forward behind the long-list sort blocks: dqo_k1_where == 2)
        if ((int)blockIdx.x >= late.first_block) {
            k1_late_block<SORT_THREADS>(late, g, (int)blockIdx.x - late.first_block);
            return;
        }
    }
    const uint32_t sort_blocks = LATE ? (uint32_t)late.first_block : gridDim.x;
    __shared__ uint64_t s_keys[SORTL_SEG];
    __shared__ uint32_t s_vals[SORTL_SEG];
    if (keep_order && blockIdx.x == 0 && threadIdx.x < 64) dqo_header_from_spread(g, capacity, bin.bucket, (int)threadIdx.x);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t n_long = min(g.counters[1], (uint32_t)T);  // tiles queued by tile_sort_wave_kernel
    for (uint32_t q = blockIdx.x; q < n_long; q += sort_blocks) {  // (block-uniform trip count; every helper ends with a barrier)
    const uint32_t tile = img.long_tiles[q];
    const uint2 rg = img.ranges[tile];
    const int n = (int)(rg.y - rg.x);
    uint4* gr = bin.recs + rg.x;
#ifdef DQO_LONG_SORT_RADIX
    // (build variant for the A/B recorded in DESIGN.md §2: lists of up to 2048 entries by north_star's "wavefront ballot / prefix-sum
    // radix sort" — a stable LSD radix sort in LDS, 8-bit digits, a key's rank among the equal digits of its round from eight wave
    // ballots + per-wave counts, the scheme of knn.hip's radix_scatter_kernel — instead of the bitonic network; longer lists keep it)
    if (n <= 2048) {
        __shared__ uint32_t s_run[256];
        __shared__ uint32_t s_wc[SORT_THREADS / 64][256];
        __shared__ uint32_t s_ws[4];
        uint64_t* ka = s_keys;
        uint64_t* kb = s_keys + 2048;
        uint32_t* va = s_vals;
        uint32_t* vb = s_vals + 2048;
        for (int i = tid; i < n; i += SORT_THREADS) {
            const uint4 e = gr[i];
            ka[i] = ((uint64_t)e.y << 32) | e.x, va[i] = e.z;
        }
        const int shifts[7] = {0, 8, 16, 32, 40, 48, 56};  // Gaussian ids below 2^24, then the depth bits
#pragma unroll 1
        for (int pass = 0; pass < 7; pass++) {
            const int shift = shifts[pass];
            if (tid < 256) s_run[tid] = 0;
#pragma unroll
            for (int w = 0; w < SORT_THREADS / 64; w++)
                if (tid < 256) s_wc[w][tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += SORT_THREADS) atomicAdd(&s_run[(uint32_t)(ka[i] >> shift) & 255u], 1u);
            __syncthreads();
            if (tid < 256) {  // exclusive scan of the 256 digit counts (four waves)
                const uint32_t c = s_run[tid];
                uint32_t incl = c;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = __shfl_up(incl, off);
                    if (lane >= off) incl += o;
                }
                if (lane == 63) s_ws[wave] = incl;
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                s_run[tid] = incl - c;  // (wave-local; the wave bases are added below, after the barrier)
            }
            __syncthreads();
            if (tid < 256) {
                uint32_t add = 0;
                for (int w = 0; w < wave; w++) add += s_ws[w];
                s_run[tid] += add;
            }
            __syncthreads();
            for (int base = 0; base < n; base += SORT_THREADS) {
                const int i = base + tid;
                const bool ok = i < n;
                const uint64_t key = ok ? ka[i] : 0ull;
                const uint32_t val = ok ? va[i] : 0u;
                const uint32_t d = (uint32_t)(key >> shift) & 255u;
                unsigned long long peers = __builtin_amdgcn_ballot_w64(ok);
#pragma unroll
                for (int bit = 0; bit < 8; bit++) {
                    const bool one = ((d >> bit) & 1u) != 0u;
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(one);
                    peers &= one ? bal : ~bal;
                }
                const uint32_t rank_w = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
                if (ok && rank_w == 0u) s_wc[wave][d] = (uint32_t)__popcll(peers);
                __syncthreads();
                if (ok) {
                    uint32_t pos = s_run[d] + rank_w;
                    for (int w = 0; w < wave; w++) pos += s_wc[w][d];
                    kb[pos] = key, vb[pos] = val;
                }
                __syncthreads();
                if (tid < 256) {
                    uint32_t c = 0;
#pragma unroll
                    for (int w = 0; w < SORT_THREADS / 64; w++) c += s_wc[w][tid], s_wc[w][tid] = 0;
                    s_run[tid] += c;
                }
                __syncthreads();
            }
            uint64_t* tk = ka;
            ka = kb, kb = tk;
            uint32_t* tv = va;
            va = vb, vb = tv;
        }
        for (int i = tid; i < n; i += SORT_THREADS) {
            bin.point_list[rg.x + i] = (uint32_t)(ka[i] & 0xffffffffu);
            bin.slot_list[rg.x + i] = va[i];
        }
        __syncthreads();
        continue;
    }
#endif
    int n2 = 2 * SORTL_RUN;
    while (n2 < n) n2 <<= 1;
    const int seg_len = min(n2, SORTL_SEG);

    // one ascending compare-exchange step of the network on the segment in LDS: `flip` pairs i with its mirror inside blocks of k
    // (first step of a merge of two ascending halves), otherwise i with i + j
    auto lds_step = [&](int k, int j, bool flip) {
        for (int t = tid; t < seg_len / 2; t += SORT_THREADS) {
            int i, p;
            if (flip) {
                const int h = k >> 1, blk = t / h, off = t - blk * h;
                i = blk * k + off, p = blk * k + k - 1 - off;
            } else {
                i = 2 * j * (t / j) + (t % j), p = i + j;
            }
            const uint64_t a = s_keys[i], b = s_keys[p];
            if (a > b) {
                s_keys[i] = b, s_keys[p] = a;
                const uint32_t va = s_vals[i];
                s_vals[i] = s_vals[p], s_vals[p] = va;
            }
        }
        __syncthreads();
    };
    // every run of the segment through the registers of one wave: full sort (first) or the last nine steps of a merge
    auto runs_in_registers = [&](bool full_sort) {
        for (int run = wave; run < seg_len / SORTL_RUN; run += SORT_THREADS / 64) {
            uint64_t key[SORTP_E];
            uint32_t val[SORTP_E];
            const int base = run * SORTL_RUN + lane * SORTP_E;
#pragma unroll
            for (int r = 0; r < SORTP_E; r++) key[r] = s_keys[base + r], val[r] = s_vals[base + r];
            if (full_sort) wave_bitonic<SORTP_E>(key, val, lane);
            else wave_bitonic_phase<SORTP_E>(key, val, lane, 2 * SORTL_RUN);
#pragma unroll
            for (int r = 0; r < SORTP_E; r++) s_keys[base + r] = key[r], s_vals[base + r] = val[r];
        }
        __syncthreads();
    };
    auto load_segment = [&](int s0) {
        for (int i = tid; i < seg_len; i += SORT_THREADS) {
            const bool in = s0 + i < n;
            const uint4 e = in ? gr[s0 + i] : make_uint4(~0u, ~0u, 0u, 0u);  // padding sorts behind every real key
            s_keys[i] = ((uint64_t)e.y << 32) | e.x;
            s_vals[i] = e.z;
        }
        __syncthreads();
    };
    auto store_segment = [&](int s0, bool final_lists) {
        for (int i = tid; i < seg_len; i += SORT_THREADS) {
            if (s0 + i >= n) continue;
            if (final_lists) {
                bin.point_list[rg.x + s0 + i] = (uint32_t)(s_keys[i] & 0xffffffffu);
                bin.slot_list[rg.x + s0 + i] = s_vals[i];
            } else {
                gr[s0 + i] = make_uint4((uint32_t)s_keys[i], (uint32_t)(s_keys[i] >> 32), s_vals[i], 0u);
            }
        }
        __syncthreads();
    };
    const int nseg = n2 / seg_len;
    // ---- every segment sorted on its own ----
    for (int sg = 0; sg < nseg; sg++) {
        load_segment(sg * seg_len);
        runs_in_registers(true);
        for (int k = 2 * SORTL_RUN; k <= seg_len; k <<= 1) {
            lds_step(k, 0, true);
            for (int j = k >> 2; j >= SORTL_RUN; j >>= 1) lds_step(k, j, false);
            runs_in_registers(false);
        }
        store_segment(sg * seg_len, nseg == 1);
    }
    // ---- merges across segments ----
    for (int k = 2 * seg_len; k <= n2 && nseg > 1; k <<= 1) {
        for (int j = k >> 1; j >= seg_len; j >>= 1) {  // global steps: the flip at distance k, then half cleaners down to one segment
            const bool flip = j == (k >> 1);
            for (int t = tid; t < n2 / 2; t += SORT_THREADS) {
                int i, p;
                if (flip) {
                    const int blk = t / j, off = t - blk * j;
                    i = blk * k + off, p = blk * k + k - 1 - off;
                } else {
                    i = 2 * j * (t / j) + (t % j), p = i + j;
                }
                if (p < n) {  // (a partner past the end is +infinity: nothing to exchange)
                    const uint4 a = gr[i], b = gr[p];
                    if ((((uint64_t)a.y << 32) | a.x) > (((uint64_t)b.y << 32) | b.x)) gr[i] = b, gr[p] = a;
                }
            }
            __syncthreads();
        }
        const bool last = (k << 1) > n2;
        for (int sg = 0; sg < nseg; sg++) {
            if (sg * seg_len >= n) break;  // a segment of padding only
            load_segment(sg * seg_len);
            for (int j = seg_len >> 1; j >= SORTL_RUN; j >>= 1) lds_step(2 * j, j, false);
            runs_in_registers(false);
            store_segment(sg * seg_len, last);
        }
    }
    }  // queue loop
}

// Zero fill of the per-frame scalars (header, slot allocator, statistics counters).  A kernel, not hipMemsetAsync: as a memset NODE
// of a captured graph the fill was observed (ROCm 7.2, gfx950) to write garbage once enough other runtime activity had happened
// between the capture and a replay — the lists were then emptied by a slot allocator that started at a random value (round 1's
// blank frames of the 2-rank rehearsal).  A kernel node carries its arguments by value.
__global__ void zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    const float* __restrict__ proj, uint8_t* __restrict__ present) {
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    const float hx = proj[0] * px + proj[4] * py + proj[8] * pz + proj[12];
    const float hy = proj[1] * px + proj[5] * py + proj[9] * pz + proj[13];
    const float hw = proj[3] * px + proj[7] * py + proj[11] * pz + proj[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float projx = hx * p_w, projy = hy * p_w;
    const float tvz = view[2] * px + view[6] * py + view[10] * pz + view[14];
    present[idx] = !(tvz <= 0.2f || (double)projx < -1.3 || (double)projx > 1.3 || (double)projy < -1.3 || (double)projy > 1.3);
}

}  // namespace

int dqo_launch_blend_forward(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                             const DqoRastOutputs& out, int T, const DqoTapDev& tap, const DqoGateDev& gate, int list_split, hipStream_t s,
                             int64_t header_capacity);
int dqo_launch_bin_count(int P, int gx, const int32_t* tile_mask, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                         int64_t capacity, const unsigned long long* tile_objects, hipStream_t s, const uint8_t* row_flags);
int dqo_launch_bin_place(const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin, int64_t capacity, hipStream_t s);

int dqo_launch_zero_words(uint32_t* p, size_t n_words, hipStream_t s) {
    if (n_words == 0) return DQO_OK;
    DQO_LAUNCH("zero_words_kernel", zero_words_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), s, p, n_words);
    return DQO_OK;
}

int dqo_launch_forward_prepare(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, hipStream_t s) {
    const DqoView v = dqo_make_view(p, in);
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    DqoImageLayout img = dqo_image_layout(ctx->image, p->W, p->H);
    const int T = v.gx * v.gy;
    if (!(ctx->frame_prezeroed != 0 && p->P > 0)) {  // header + counters + spread statistics counters
        // (+ the per-object loss counters, which sit directly behind them, when the loss tap is per object)
        // (frame_prezeroed: the previous frame's dqo_rast_backward_adam left counters .. loss counters at zero; the header is rewritten
        // by every frame)
        int rc = dqo_launch_zero_words(reinterpret_cast<uint32_t*>(g.header), 256 / 4 + dqo_frame_scalar_words(ctx), s);
        if (rc) return rc;
    }
    const size_t zero_words = (size_t)((img.tile_flag + T) - img.tile_count);  // histogram (padded) + flags
    if (p->P <= 0) {
        int rc = dqo_launch_zero_words(img.tile_count, zero_words, s);
        if (rc) return rc;
    }
    if (p->P > 0 && !dqo_fuse_k1(p, ctx)) {  // (fused: bin_count_kernel<true> does this part, dqo_launch_forward_render)
        const int per_block = K1_THREADS * K1_ITEMS;
        const int grid = (p->P + per_block - 1) / per_block;
        const int32_t* gobj = ctx->object_gate ? ctx->object_gate->gaussian_object : nullptr;
        const int prez = ctx->frame_prezeroed != 0 ? 1 : 0;
        if (dqo_k1_where(p->P) != 0) {
            DQO_LAUNCH("preprocess_kernel", preprocess_kernel<false>, dim3(grid), dim3(K1_THREADS), s, v, in->means3D, in->scales, in->rotations,
                       in->opacities, in->shs, in->colors_precomp, in->tile_mask, g, out->radii, out->n_touched, img.tile_count, zero_words, gobj, prez);
        } else {
            DQO_LAUNCH("preprocess_kernel", preprocess_kernel<true>, dim3(grid), dim3(K1_THREADS), s, v, in->means3D, in->scales, in->rotations,
                       in->opacities, in->shs, in->colors_precomp, in->tile_mask, g, out->radii, out->n_touched, img.tile_count, zero_words, gobj, prez);
        }
    }
    return DQO_OK;
}

// header_host / header_event (both optional): once the frame's device header is final — behind the sort kernels, BEFORE the blend kernel
// — its 32 bytes are copied to (pinned) host memory on the launch stream and the event is recorded: a caller that carries its instance
// capacity over from earlier frames (no host synchronisation in the forward) learns whether the capacity was enough a whole blend
// kernel earlier than from a copy issued behind the forward.
int dqo_launch_forward_render(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, hipStream_t s,
                              DqoRastHeader* header_host, hipEvent_t header_event) {
    const DqoView v = dqo_make_view(p, in);
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    DqoImageLayout img = dqo_image_layout(ctx->image, p->W, p->H);
    const int bucket = ctx->tile_bucket_capacity;
    const int64_t cap = (int64_t)ctx->inst_capacity;
    DqoBinLayout bin = dqo_bin_layout(ctx->binning, cap, dqo_list_cap(cap, p->W, p->H, bucket), bucket);
    const int T = v.gx * v.gy;
    if (p->P > 0) {
        // footprint test, per-tile histogram + ranks, tiles_touched, gaussian-major slots (forward.cu:344-353, rasterizer_impl.cu:303)
        const unsigned long long* tobj = ctx->object_gate ? reinterpret_cast<const unsigned long long*>(ctx->object_gate->tile_objects) : nullptr;
        int rc = dqo_fuse_k1(p, ctx) ? dqo_launch_bin_count_k1(v, in, out, ctx->object_gate ? ctx->object_gate->gaussian_object : nullptr, g, img,
                                                               bin, cap, tobj, s)
                                     : dqo_launch_bin_count(p->P, v.gx, in->tile_mask, g, img, bin, cap, tobj, s, v.row_flags);
        if (rc) return rc;
    }
    // bucket mode with a tile_order kept from an earlier frame on the same image buffer: nothing of tile_scan_kernel is needed
    // (ranges and header come from the sort kernels)
    const int keep_order = (bucket > 0 && ctx->keep_tile_order != 0 && p->P > 0) ? 1 : 0;
    if (!keep_order) DQO_LAUNCH("tile_scan_kernel", tile_scan_kernel, dim3(1), dim3(SCAN_THREADS), s, T, img, g, cap, bucket);
    if (p->P > 0) {
        if (bucket <= 0) {  // (bucket mode: bin_count_kernel has already written every instance to tile * bucket + rank)
            int rc = dqo_launch_bin_place(g, img, bin, cap, s);
            if (rc) return rc;
        }
        const int slots = 8 * ((T + 7) / 8);  // tile_order is [8][T8]
        DqoK1Late late;
        late.v = v, late.means3D = in->means3D, late.scales = in->scales, late.rotations = in->rotations, late.shs = in->shs;
        late.colors_precomp = in->colors_precomp, late.first_block = slots;
        if (dqo_k1_where(p->P) == 1) {
            DQO_LAUNCH("tile_sort_wave_kernel", tile_sort_wave_kernel<true>, dim3(slots + (p->P + SORTW_THREADS - 1) / SORTW_THREADS),
                       dim3(SORTW_THREADS), s, T, img, bin, g, cap, keep_order, dqo_list_split(ctx), late);
        } else {
            DQO_LAUNCH("tile_sort_wave_kernel", tile_sort_wave_kernel<false>, dim3(slots), dim3(SORTW_THREADS), s, T, img, bin, g, cap, keep_order,
                       dqo_list_split(ctx), late);
        }
        if (dqo_skip_long_sort(p, ctx)) {
            // no list can be longer than the per-tile sort reaches (a tile that outgrows its bucket is flagged): no long-list launch;
            // the frame's header — that launch's first block forms it in keep_order frames — comes from an extra block of the blend launch
        } else if (dqo_k1_where(p->P) == 2) {
            late.first_block = SORT_GRID;
            DQO_LAUNCH("tile_sort_kernel", tile_sort_kernel<true>, dim3(SORT_GRID + (p->P + SORT_THREADS - 1) / SORT_THREADS), dim3(SORT_THREADS), s, T,
                       img, bin, g, cap, keep_order, late);
        } else {
            DQO_LAUNCH("tile_sort_kernel", tile_sort_kernel<false>, dim3(SORT_GRID), dim3(SORT_THREADS), s, T, img, bin, g, cap, keep_order, late);
        }
    }
    // In a frame without the long-list sort launch the header is formed by an extra block of the BLEND launch: the copy and the event
    // go behind it there (the caller learns about an overflow one blend kernel later, never from a stale header).
    const bool header_in_blend = dqo_skip_long_sort(p, ctx);
    auto hand_over_header = [&]() -> int {
        if (header_host != nullptr) DQO_CHECK_HIP(hipMemcpyAsync(header_host, g.header, sizeof(DqoRastHeader), hipMemcpyDeviceToHost, s));
        if (header_event != nullptr) DQO_CHECK_HIP(hipEventRecord(header_event, s));
        return DQO_OK;
    };
    if (!header_in_blend) {
        const int rc = hand_over_header();
        if (rc) return rc;
    }
    const int rc = dqo_launch_blend_forward(v, g, img, bin, *out, T, dqo_tap_dev(ctx->loss_tap), dqo_gate_dev(ctx->object_gate),
                                            dqo_list_split(ctx), s, header_in_blend ? cap : (int64_t)-1);
    if (rc) return rc;
    return header_in_blend ? hand_over_header() : DQO_OK;
}

// A K1-fused frame launches nothing in its first stage, so the device header still shows the PREVIOUS frame's stage 2: a caller that
// runs the two stages as separate calls gets the header marked "stage 1 not available" (dqo_rast_read_header then reports zeros, as
// include/dqo_raster.h says) by one single-word launch; dqo_rast_forward / _async never pay it.
int dqo_launch_mark_header_stage0(const DqoRastParams* p, DqoRastCtx* ctx, hipStream_t s) {
    if (!dqo_fuse_k1(p, ctx)) return DQO_OK;
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    return dqo_launch_zero_words(&g.header->stage, 1, s);
}

bool dqo_skip_long_sort(const DqoRastParams* p, const DqoRastCtx* ctx) {
    static const bool on = [] {
        const char* e = getenv("DQO_SKIP_LONG_SORT");
        return !(e != nullptr && e[0] == '0');
    }();
    return on && p->P > 0 && ctx->tile_bucket_capacity > 0 && ctx->tile_bucket_capacity <= SORTW_CAP && ctx->keep_tile_order != 0 &&
           dqo_k1_where(p->P) != 2 && dqo_list_split(ctx) == 0;
}

bool dqo_fuse_k1(const DqoRastParams* p, const DqoRastCtx* ctx) {
    static const bool on = [] {
        const char* e = getenv("DQO_K1_FUSE");
        return !(e != nullptr && e[0] == '0');
    }();
    return on && ctx->frame_prezeroed != 0 && p->P > 0 && ctx->tile_bucket_capacity > 0 && dqo_k1_where(p->P) != 0;
}

int dqo_launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present, hipStream_t s) {
    if (P <= 0) return DQO_OK;
    DQO_LAUNCH("mark_visible_kernel", mark_visible_kernel, dim3((P + 255) / 256), dim3(256), s, P, means3D, view, proj, present);
    return DQO_OK;
}
