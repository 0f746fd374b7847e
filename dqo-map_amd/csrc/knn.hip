// Exact 3-nearest-neighbour search (mean squared distance + neighbour indices) for gfx950.
//
// Replaces /root/reference/submodules/simple-knn/simple_knn.cu:45-252 (SimpleKNN::knn) and spatial.cu:15-28 (distCUDA2):
// same result = for every point the three smallest squared distances to OTHER positions, ties resolved by the order of
// a stable sort on the 30-bit Morton code (the order in which the reference's scan meets the candidates).
//
// MI355X design: no cudaMalloc / thrust allocations / D2H copies inside the call (the reference does 2 blocking 12-byte
// reads for the bounding box and allocates five temporaries per call): one caller-owned workspace, everything on the
// caller's stream.  Points are gathered ONCE into Morton order as float4 (xyz + original index), so the candidate scan
// reads contiguous, wave-uniform addresses (broadcast loads) instead of chasing an index indirection per candidate.
// The sort is a stable LSD radix sort of the 30-bit codes (index as payload: equal codes keep index order = the reference's stable
// sort), four 8-bit passes of histogram -> scan -> scatter; the rank of a key among the equal digits of its block comes from wave
// ballots (eight ballots match the lanes with the same digit) + per-wave counts in LDS.  12 launches for any size; the first
// version — a 64-bit bitonic network, 4096-key runs in LDS, merge steps >= 4096 in global memory — needed 55 launches and 0.73 ms
// for 2 M keys (`make EXTRA=-DKNN_BITONIC_SORT` keeps it for comparison).
#include <float.h>
#include <limits.h>

#include <algorithm>

#include "dqo_common.h"

namespace {

constexpr int KNN_BOX = 1024;  // simple_knn.cu:16 BOX_SIZE (part of the pruning structure only)
constexpr int KNN_SUB = 64;    // second pruning level of the query search: one wave-load of sorted points
constexpr int SORT_RUN = 4096;
constexpr int SORT_T = 256;
constexpr int RDX_T = 256, RDX_KPT = 8, RDX_BLK = RDX_T * RDX_KPT;  // radix sort: keys per block

struct KnnWs {
    float* bbox;        // [8] min xyz, max xyz
    uint64_t* keys;     // [P2] (morton << 32 | original index), padded to a power of two with ~0
    float4* sorted;     // [P] (x, y, z, bits(original index)) in Morton order
    float* boxes;       // [nb][8] min xyz, max xyz of each run of 1024 sorted points
    float* sub;         // [ceil(P / 64)][8] ... of each run of 64 sorted points (query search only)
    uint32_t* hist;     // [256][ceil(P / RDX_BLK)] digit histograms / scatter offsets of one radix pass
    float* groups;      // [ceil(nb / 64)][8] min / max of each run of 64 level-1 boxes = 65 536 sorted points (query search only)
    size_t total;
};

inline int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

inline KnnWs knn_ws(void* base, int P) {
    KnnWs w;
    char* p = (char*)base;
    auto take = [&](size_t b) {
        char* r = p;
        p += dqo_align_up(b, 256);
        return r;
    };
    const int P2 = next_pow2(P < SORT_RUN ? SORT_RUN : P);
    w.bbox = (float*)take(32);
    w.keys = (uint64_t*)take(8 * (size_t)P2);
    w.sorted = (float4*)take(16 * (size_t)P);
    w.boxes = (float*)take(32 * (size_t)((P + KNN_BOX - 1) / KNN_BOX));
    w.sub = (float*)take(32 * (size_t)((P + KNN_SUB - 1) / KNN_SUB));
    w.hist = (uint32_t*)take(4 * 256 * ((size_t)((P + RDX_BLK - 1) / RDX_BLK) + 1));  // + the 256 row totals behind the table
    w.groups = (float*)take(32 * (size_t)(((P + KNN_BOX - 1) / KNN_BOX + 63) / 64));
    w.total = (size_t)(p - (char*)base);
    return w;
}

// min / max over all points with init {0,0,0}: the bounding box always contains the origin (simple_knn.cu:222-231, B15).
// Two stages: every block reduces a strided slice of the points into a partial box, one block folds the partial boxes (a single
// block over all points took 0.68 ms on a 2 M-point map — a quarter of a map-growth query).
constexpr int BBOX_BLOCKS_MAX = 512;
__device__ __forceinline__ void bbox_block_reduce(float mn[3], float mx[3], float* __restrict__ out) {
    __shared__ float s[6][16];
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int a = 0; a < 3; a++) s[a][wave] = mn[a], s[3 + a][wave] = mx[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = s[threadIdx.x][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) r = threadIdx.x < 3 ? fminf(r, s[threadIdx.x][w]) : fmaxf(r, s[threadIdx.x][w]);
        out[threadIdx.x] = r;
    }
}
// group (dqo_knn3_query_grouped): points without a group (id outside [0, 64): spare rows parked far away, deleted Gaussians) take no part
// in the search — they do not stretch the bounding box the Morton grid is laid over, sort behind every real point and are gathered as
// a far sentinel, so the boxes they fill are pruned by distance
__device__ __forceinline__ bool knn_in_group(const int32_t* __restrict__ group, int i) {
    if (group == nullptr) return true;
    const int gid = group[i];
    return gid >= 0 && gid < 64;
}
__global__ __launch_bounds__(1024) void bbox_kernel(int P, const float* __restrict__ xyz, float* __restrict__ partial,
                                                    const int32_t* __restrict__ group) {
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        if (!knn_in_group(group, i)) continue;
        for (int a = 0; a < 3; a++) {
            const float v = xyz[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    }
    bbox_block_reduce(mn, mx, partial + 8 * blockIdx.x);
}
__global__ __launch_bounds__(512) void bbox_fold_kernel(int n, const float* __restrict__ partial, float* __restrict__ bbox) {
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        for (int a = 0; a < 3; a++) mn[a] = fminf(mn[a], partial[8 * i + a]), mx[a] = fmaxf(mx[a], partial[8 * i + 3 + a]);
    bbox_block_reduce(mn, mx, bbox);
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
// CUDA float -> uint32 conversion semantics (NaN -> 0, saturating) for the degenerate 0/0 axis.
__device__ __forceinline__ uint32_t f2u(float v) {
    if (!(v == v) || v <= 0.f) return 0u;
    if (v >= 4294967296.f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

// coord2Morton, simple_knn.cu:54-70
__global__ void morton_kernel(int P, int P2, const float* __restrict__ xyz, const float* __restrict__ bbox, uint64_t* __restrict__ keys) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P2) return;
    if (i >= P) {
        keys[i] = ~0ull;
        return;
    }
    const float mnx = bbox[0], mny = bbox[1], mnz = bbox[2], mxx = bbox[3], mxy = bbox[4], mxz = bbox[5];
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const uint32_t cx = prep_morton(f2u(((x - mnx) / (mxx - mnx)) * (float)((1 << 10) - 1)));
    const uint32_t cy = prep_morton(f2u(((y - mny) / (mxy - mny)) * (float)((1 << 10) - 1)));
    const uint32_t cz = prep_morton(f2u(((z - mnz) / (mxz - mnz)) * (float)((1 << 10) - 1)));
    const uint32_t code = cx | (cy << 1) | (cz << 2);
    keys[i] = ((uint64_t)code << 32) | (uint32_t)i;  // unique keys: order == stable sort by code
}

// The query search's reference set is not bound to the reference's 10 bits per axis (its ties are "resolved arbitrarily"): 13 bits per
// axis (39-bit code above a 25-bit index: up to 33 M points) keep the 1024-point Morton boxes tight when the points spread over tens of
// metres — the per-object growth search moves every object into a cell of its own (dqo_mapgrowth.object_offsets), and at 10 bits a
// 30 m extent means 3 cm cells: boxes of neighbouring surfels overlap and the scan visits 2.3x the sub-boxes.
constexpr int FINE_BITS = 13, FINE_IDX_BITS = 25;
__device__ __forceinline__ uint64_t prep_morton64(uint64_t x) {  // bit i -> bit 3 i, up to 21 bits
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ void morton_fine_kernel(int P, int P2, const float* __restrict__ xyz, const float* __restrict__ bbox, uint64_t* __restrict__ keys,
                                   const int32_t* __restrict__ group) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P2) return;
    if (i >= P) {
        keys[i] = ~0ull;
        return;
    }
    if (!knn_in_group(group, i)) {  // behind every real point (the padding keys ~0 are larger still)
        keys[i] = (((1ull << (3 * FINE_BITS)) - 1ull) << FINE_IDX_BITS) | (uint64_t)i;
        return;
    }
    const float mnx = bbox[0], mny = bbox[1], mnz = bbox[2], mxx = bbox[3], mxy = bbox[4], mxz = bbox[5];
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const float top = (float)((1 << FINE_BITS) - 1);
    const uint64_t cx = prep_morton64(min(f2u(((x - mnx) / (mxx - mnx)) * top), (uint32_t)top));
    const uint64_t cy = prep_morton64(min(f2u(((y - mny) / (mxy - mny)) * top), (uint32_t)top));
    const uint64_t cz = prep_morton64(min(f2u(((z - mnz) / (mxz - mnz)) * top), (uint32_t)top));
    keys[i] = ((cx | (cy << 1) | (cz << 2)) << FINE_IDX_BITS) | (uint64_t)i;
}

// classic bitonic network on a power-of-two array: LDS kernel handles every step with j < SORT_RUN of one k-level
// (or all levels k <= SORT_RUN when `first`), the global kernel one step with j >= SORT_RUN.
__global__ __launch_bounds__(SORT_T) void bitonic_lds_kernel(uint64_t* __restrict__ keys, int k_level, int first) {
    __shared__ uint64_t s[SORT_RUN];
    const int base = blockIdx.x * SORT_RUN;
    for (int i = threadIdx.x; i < SORT_RUN; i += SORT_T) s[i] = keys[base + i];
    __syncthreads();
    const int k_lo = first ? 2 : k_level, k_hi = first ? SORT_RUN : k_level;
    for (int k = k_lo; k <= k_hi; k <<= 1) {
        for (int j = min(k >> 1, SORT_RUN >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < SORT_RUN / 2; t += SORT_T) {
                const int i = 2 * j * (t / j) + (t % j);
                const int p = i + j;
                const bool up = (((base + i) & k) == 0);
                const uint64_t a = s[i], b = s[p];
                if ((a > b) == up) {
                    s[i] = b;
                    s[p] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < SORT_RUN; i += SORT_T) keys[base + i] = s[i];
}

__global__ void bitonic_global_kernel(uint64_t* __restrict__ keys, int n_half, int k, int j) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_half) return;
    const int i = 2 * j * (t / j) + (t % j);
    const int p = i + j;
    const bool up = ((i & k) == 0);
    const uint64_t a = keys[i], b = keys[p];
    if ((a > b) == up) {
        keys[i] = b;
        keys[p] = a;
    }
}

// ---- stable LSD radix sort, one 8-bit pass = three kernels ----
// (1) digit histogram of every block of RDX_BLK keys, stored digit-major: hist[digit][block]
__global__ __launch_bounds__(RDX_T) void radix_hist_kernel(int n, const uint64_t* __restrict__ keys, int shift, uint32_t* __restrict__ hist,
                                                            int nblk) {
    __shared__ uint32_t s[256];
    s[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RDX_BLK;
#pragma unroll
    for (int r = 0; r < RDX_KPT; r++) {
        const int i = base + r * RDX_T + threadIdx.x;
        if (i < n) atomicAdd(&s[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = s[threadIdx.x];
}
// (2) one block per digit: exclusive prefix over that digit's row (its count in block 0, 1, ...) in place, the row's total to tot[digit]
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ hist, int nblk, uint32_t* __restrict__ tot) {
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* row = hist + (size_t)blockIdx.x * nblk;
    const int per = (nblk + 255) / 256, a = tid * per, b = min(nblk, a + per);
    uint32_t sum = 0;
    for (int i = a; i < b; i++) sum += row[i];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - sum;
    for (int w = 0; w < wave; w++) pre += s_w[w];
    for (int i = a; i < b; i++) {
        const uint32_t h = row[i];
        row[i] = pre;
        pre += h;
    }
    if (tid == 255) tot[blockIdx.x] = pre;  // (the last thread's running sum is the row total)
}
// (3) scatter: key -> offs[digit][block] + its rank among the block's keys of the same digit, in block order (stable)
__global__ __launch_bounds__(RDX_T) void radix_scatter_kernel(int n, const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int shift,
                                                               const uint32_t* __restrict__ offs, int nblk,
                                                               const uint32_t* __restrict__ tot) {
    __shared__ uint32_t s_run[256];              // next free position of every digit (earlier rounds of this block included)
    __shared__ uint32_t s_wc[RDX_T / 64][256];   // this round's keys per (wave, digit)
    __shared__ uint32_t s_t[RDX_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // where digit `tid` starts in the output = the keys of all smaller digits: exclusive prefix over the 256 row totals
        const uint32_t t = tot[tid];
        uint32_t incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_t[wave] = incl;
        __syncthreads();
        uint32_t pre = incl - t;
        for (int w = 0; w < wave; w++) pre += s_t[w];
        s_run[tid] = pre + offs[(size_t)tid * nblk + blockIdx.x];
    }
#pragma unroll
    for (int w = 0; w < RDX_T / 64; w++) s_wc[w][tid] = 0;
    __syncthreads();
    const int base = blockIdx.x * RDX_BLK;
    for (int r = 0; r < RDX_KPT; r++) {
        const int i = base + r * RDX_T + tid;
        const bool valid = i < n;
        const uint64_t key = valid ? in[i] : 0ull;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        // the lanes of this wave that hold the same digit: eight ballots
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool one = ((d >> bit) & 1u) != 0u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(one);
            peers &= one ? bal : ~bal;
        }
        const uint32_t rank_w = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank_w == 0u) s_wc[wave][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = s_run[d] + rank_w;
            for (int w = 0; w < wave; w++) pos += s_wc[w][d];
            out[pos] = key;
        }
        __syncthreads();
        {   // digit `tid`: advance its position by this round's keys, clear the round's counts
            uint32_t c = 0;
#pragma unroll
            for (int w = 0; w < RDX_T / 64; w++) c += s_wc[w][tid], s_wc[w][tid] = 0;
            s_run[tid] += c;
        }
        __syncthreads();
    }
}

// group (dqo_knn3_query_grouped): the point's group id + 1 rides in bits 25..31 of the index word (0 = the point belongs to no group)
constexpr int KNN_GROUP_SHIFT = 25;
__global__ void gather_sorted_kernel(int P, const float* __restrict__ xyz, const uint64_t* __restrict__ keys, float4* __restrict__ sorted,
                                     uint64_t idx_mask, const int32_t* __restrict__ group) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const uint32_t src = (uint32_t)(keys[i] & idx_mask);
    uint32_t w = src;
    if (group != nullptr) {
        const int gid = group[src];
        if (!(gid >= 0 && gid < 64)) {
            sorted[i] = make_float4(1e18f, 1e18f, 1e18f, __uint_as_float(w));  // no group: a far sentinel (group word 0 matches no query)
            return;
        }
        w |= (uint32_t)(gid + 1) << KNN_GROUP_SHIFT;
    }
    sorted[i] = make_float4(xyz[3 * src], xyz[3 * src + 1], xyz[3 * src + 2], __uint_as_float(w));
}

// Every pruning record (level-1 box, run of 64 points, group of 64 boxes: 8 floats, 6 of them min / max) carries in its two spare words
// the SET OF GROUPS present among its points (dqo_knn3_query_grouped; all zero without groups): a record without the query's group is
// skipped whatever its distance — objects are spatially coherent, so most Morton boxes hold one or two of them.
__device__ __forceinline__ unsigned long long knn_group_bit(float w) {
    const uint32_t g1 = __float_as_uint(w) >> KNN_GROUP_SHIFT;  // group id + 1, 0 = none (always 0 in an ungrouped build of the set)
    return g1 ? 1ull << (g1 - 1u) : 0ull;
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= (unsigned long long)__shfl_xor((long long)v, off);
    return v;
}
__device__ __forceinline__ unsigned long long knn_record_groups(const float* rec) {
    return ((unsigned long long)__float_as_uint(rec[7]) << 32) | (unsigned long long)__float_as_uint(rec[6]);
}
// boxMinMax, simple_knn.cu:78-117
__global__ __launch_bounds__(256) void box_minmax_kernel(int P, const float4* __restrict__ sorted, float* __restrict__ boxes, int grouped) {
    __shared__ float s[6][4];
    __shared__ unsigned long long s_g[4];
    const int b = blockIdx.x;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    unsigned long long gm = 0ull;
    for (int i = b * KNN_BOX + threadIdx.x; i < min(P, (b + 1) * KNN_BOX); i += blockDim.x) {
        const float4 p = sorted[i];
        mn[0] = fminf(mn[0], p.x), mn[1] = fminf(mn[1], p.y), mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x), mx[1] = fmaxf(mx[1], p.y), mx[2] = fmaxf(mx[2], p.z);
        if (grouped) gm |= knn_group_bit(p.w);
    }
    gm = wave_or64(gm);
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        for (int a = 0; a < 3; a++) s[a][wave] = mn[a], s[3 + a][wave] = mx[a];
        s_g[wave] = gm;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = s[threadIdx.x][0];
        for (int w = 1; w < 4; w++) r = threadIdx.x < 3 ? fminf(r, s[threadIdx.x][w]) : fmaxf(r, s[threadIdx.x][w]);
        boxes[8 * b + threadIdx.x] = r;
    } else if (threadIdx.x < 8) {
        const unsigned long long g = s_g[0] | s_g[1] | s_g[2] | s_g[3];
        boxes[8 * b + threadIdx.x] = __uint_as_float(threadIdx.x == 6 ? (uint32_t)g : (uint32_t)(g >> 32));
    }
}

// updateKBest_idx, simple_knn.cu:147-167 (strict '>' keeps the earlier-visited candidate on ties)
__device__ __forceinline__ void kbest(const float4 ref, const float4 cand, float (&best)[3], int (&bidx)[3]) {
#pragma clang fp contract(off)
    const float dx = cand.x - ref.x, dy = cand.y - ref.y, dz = cand.z - ref.z;
    float dist = dx * dx + dy * dy + dz * dz;
    int id = (int)__float_as_uint(cand.w);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
            const int r = bidx[j];
            bidx[j] = id;
            id = r;
        }
    }
}

// distBoxPoint, simple_knn.cu:119-129
__device__ __forceinline__ float dist_box_point(const float* bx, const float4 p) {
#pragma clang fp contract(off)
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (p.x < bx[0] || p.x > bx[3]) dx = fminf(fabsf(p.x - bx[0]), fabsf(p.x - bx[3]));
    if (p.y < bx[1] || p.y > bx[4]) dy = fminf(fabsf(p.y - bx[1]), fabsf(p.y - bx[4]));
    if (p.z < bx[2] || p.z > bx[5]) dz = fminf(fabsf(p.z - bx[2]), fabsf(p.z - bx[5]));
    return dx * dx + dy * dy + dz * dz;
}

// boxMeanDist, simple_knn.cu:169-214.  One lane per (Morton-sorted) query; a wave's 64 queries are spatially close, so a
// box is scanned by the whole wave if ANY lane still needs it: candidate loads are wave-uniform (one broadcast load
// feeds 64 lanes) and lanes that did not need the box are unaffected (its points can never enter their top 3).
// Second level (not in the reference; changes no result): a box of 1024 Morton-consecutive points is a loose volume, so before its
// points are scanned each of its sixteen runs of 64 points is tested against its own bounding box with the same predicate — a run
// that fails it for every lane holds no point that could enter any lane's top 3 (strict '>' in kbest), and the runs that are scanned
// are scanned in the reference's order.  540 k points: 9.8 -> 1.5 ms.
__global__ __launch_bounds__(256) void knn_scan_kernel(int P, const float4* __restrict__ sorted, const float* __restrict__ boxes,
                                                       const float* __restrict__ sub, float* __restrict__ mean_d2,
                                                       int32_t* __restrict__ idx3) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < P;
    const float4 me = live ? sorted[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    int bidx[3] = {INT_MAX, INT_MAX, INT_MAX};
    if (live)
        for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++) {
            if (i == idx) continue;
            kbest(me, sorted[i], best, bidx);
        }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    bidx[0] = bidx[1] = bidx[2] = INT_MAX;
    const int nb = (P + KNN_BOX - 1) / KNN_BOX;
    for (int b = 0; b < nb; b++) {
        float bx[6];
#pragma unroll
        for (int a = 0; a < 6; a++) bx[a] = boxes[8 * b + a];
        const float d = dist_box_point(bx, me);
        const bool need = live && !(d > reject || d > best[2]);
        if (__ballot(need) == 0) continue;
        const int lo = b * KNN_BOX, hi = min(P, (b + 1) * KNN_BOX);
        for (int r0 = lo; r0 < hi; r0 += KNN_SUB) {
            float sx[6];
#pragma unroll
            for (int a = 0; a < 6; a++) sx[a] = sub[8 * (r0 / KNN_SUB) + a];  // wave-uniform address
            const float ds = dist_box_point(sx, me);
            const bool need_run = need && !(ds > reject || ds > best[2]);
            if (__ballot(need_run) == 0) continue;
            const int r1 = min(hi, r0 + KNN_SUB);
            for (int i = r0; i < r1; i++) {
                const float4 c = sorted[i];  // wave-uniform address
                if (need_run && i != idx) kbest(me, c, best, bidx);
            }
        }
    }
    if (live) {
        const uint32_t dst = __float_as_uint(me.w);
        mean_d2[dst] = (best[0] + best[1] + best[2]) / 3.0f;
        idx3[dst * 3 + 0] = bidx[0];
        idx3[dst * 3 + 1] = bidx[1];
        idx3[dst * 3 + 2] = bidx[2];
    }
}

// min / max of each run of 64 sorted points: one wave per run
__global__ __launch_bounds__(256) void sub_minmax_kernel(int P, const float4* __restrict__ sorted, float* __restrict__ sub) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int ns = (P + KNN_SUB - 1) / KNN_SUB;
    if (w >= ns) return;
    const int i = w * KNN_SUB + lane;
    const float4 p = sorted[min(i, P - 1)];  // (the run's last point again for the padding lanes: no effect on min / max)
    float mn[3] = {p.x, p.y, p.z}, mx[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    const unsigned long long g = wave_or64(i < P ? knn_group_bit(p.w) : 0ull);
    if (lane < 3) sub[8 * w + lane] = lane == 0 ? mn[0] : (lane == 1 ? mn[1] : mn[2]);
    else if (lane < 6) sub[8 * w + lane] = lane == 3 ? mx[0] : (lane == 4 ? mx[1] : mx[2]);
    else if (lane < 8) sub[8 * w + lane] = __uint_as_float(lane == 6 ? (uint32_t)g : (uint32_t)(g >> 32));
}

// third pruning level of the query search: min / max over runs of 64 level-1 boxes (one wave per group)
__global__ __launch_bounds__(256) void group_minmax_kernel(int nb, const float* __restrict__ boxes, float* __restrict__ groups) {
    const int gidx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int ng = (nb + 63) / 64;
    if (gidx >= ng) return;
    const float* bx = boxes + 8 * (size_t)min(gidx * 64 + lane, nb - 1);  // (the last box again for the padding lanes)
    float mn[3] = {bx[0], bx[1], bx[2]}, mx[3] = {bx[3], bx[4], bx[5]};
    for (int a = 0; a < 3; a++)
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    const unsigned long long g = wave_or64(gidx * 64 + lane < nb ? knn_record_groups(bx) : 0ull);
    if (lane < 3) groups[8 * gidx + lane] = lane == 0 ? mn[0] : (lane == 1 ? mn[1] : mn[2]);
    else if (lane < 6) groups[8 * gidx + lane] = lane == 3 ? mx[0] : (lane == 4 ? mx[1] : mx[2]);
    else if (lane < 8) groups[8 * gidx + lane] = __uint_as_float(lane == 6 ? (uint32_t)g : (uint32_t)(g >> 32));
}

// wave64 minimum, result wave-uniform.  DPP lanes that are not written keep `v` itself (old = v), so the minimum is unaffected.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_self(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_min_f(float v) {
    v = fminf(v, dpp_self<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
    v = fminf(v, dpp_self<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
    v = fminf(v, dpp_self<0x141, 0xF>(v));  // row_half_mirror
    v = fminf(v, dpp_self<0x140, 0xF>(v));  // row_mirror: every lane holds its row's minimum
    v = fminf(v, dpp_self<0x142, 0xA>(v));  // row_bcast15 into rows 1, 3
    v = fminf(v, dpp_self<0x143, 0xC>(v));  // row_bcast31 into rows 2, 3: lane 63 holds the wave's minimum
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// K = 3 nearest REFERENCE points of every QUERY point (row f3: Mapping.temp_points_filter, SLAM/multiprocess/mapper.py:
// 1351-1380, which calls pytorch3d.ops.knn_points(temp_xyz, exist_xyz, K=3, norm=2): exact squared L2 distances, ascending).
// ONE WAVE PER QUERY.  The references are Morton-sorted and boxed on two levels (runs of 1024 and of 64 points).  The wave's 64
// lanes test 64 level-1 boxes per trip; a box that cannot be excluded (box distance <= current third-best) has its 16 sub-boxes
// tested by 16 lanes, and a sub-box that cannot be excluded is ONE coalesced wave load: 64 candidates, one per lane, merged into
// the (wave-uniform) top 3 by up to three wave minima.  The first pass finds the level-1 box nearest to the query and scans it, so
// the second pass starts with finite bounds.  (The first version gave each query a lane and scanned a box with the whole wave as
// soon as ANY of its 64 queries needed it: with few queries against a large map — the growth step's 40 800 new points against
// 2 M Gaussians — a wave's queries lie far apart and it scanned ~45 boxes of 1024 points each: 39 ms, now 1 ms.)
// GROUPED (dqo_knn3_query_grouped): a reference only counts for a query of the same group (ids in [0, 64); a point with another id
// belongs to no group) and, with `group_box`, if it lies strictly inside its group's box [lo xyz, hi xyz] — the per-object growth
// decisions of the mapper in one search over the map as it is stored: no shifted copies, no gathered subsets.
template <bool GROUPED>
__global__ __launch_bounds__(256) void knn_query_wave_kernel(int Q, const float* __restrict__ q_xyz, int R,
                                                             const float4* __restrict__ sorted_r, const float* __restrict__ boxes,
                                                             const float* __restrict__ sub, const float* __restrict__ groups,
                                                             float* __restrict__ dist2, int32_t* __restrict__ idx3, const float bound2,
                                                             const int32_t* __restrict__ q_group, const float* __restrict__ group_box) {
    const int q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (q >= Q) return;
    const float4 me = make_float4(q_xyz[3 * q], q_xyz[3 * q + 1], q_xyz[3 * q + 2], 0.f);
    uint32_t my_group = 0u;  // group id + 1
    unsigned long long my_bit = ~0ull;
    float gl0 = -FLT_MAX, gl1 = -FLT_MAX, gl2 = -FLT_MAX, gh0 = FLT_MAX, gh1 = FLT_MAX, gh2 = FLT_MAX;
    if (GROUPED) {
        const int gid = q_group[q];
        if (!(gid >= 0 && gid < 64)) {  // a query without a group has no neighbours
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 3; k++) dist2[3 * q + k] = FLT_MAX, idx3[3 * q + k] = -1;
            }
            return;
        }
        my_group = (uint32_t)(gid + 1);
        my_bit = 1ull << gid;
        if (group_box != nullptr) {
            const float* b = group_box + 6 * gid;
            gl0 = b[0], gl1 = b[1], gl2 = b[2], gh0 = b[3], gh1 = b[4], gh2 = b[5];
        }
    }
    // wave-uniform top 3 (ascending).  bound2 (dqo_knn3_query_within): only references closer than sqrt(bound2) count — the search starts
    // with that bound instead of an open one, so a query far from every reference prunes the whole map at once
    float b0 = bound2, b1 = bound2, b2 = bound2;
    int i0 = -1, i1 = -1, i2 = -1;
    const int nb = (R + KNN_BOX - 1) / KNN_BOX, ns = (R + KNN_SUB - 1) / KNN_SUB;

    auto scan_sub = [&](int sb) {
        const int i = sb * KNN_SUB + lane;
        const float4 c = sorted_r[min(i, R - 1)];
        float dist;
        {
#pragma clang fp contract(off)
            const float dx = c.x - me.x, dy = c.y - me.y, dz = c.z - me.z;
            dist = i < R ? dx * dx + dy * dy + dz * dz : FLT_MAX;
        }
        if (GROUPED) {
            const bool ok = (__float_as_uint(c.w) >> KNN_GROUP_SHIFT) == my_group && c.x > gl0 && c.y > gl1 && c.z > gl2 && c.x < gh0 &&
                            c.y < gh1 && c.z < gh2;
            dist = ok ? dist : FLT_MAX;
        }
        const int cid = (int)(__float_as_uint(c.w) & (GROUPED ? (1u << KNN_GROUP_SHIFT) - 1u : 0xffffffffu));
        for (int rep = 0; rep < 3; rep++) {
            const float m = wave_min_f(dist);
            if (!(m < b2)) break;
            const int owner = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(dist == m));
            const int id = __builtin_amdgcn_readlane(cid, owner);
            if (m < b0) b2 = b1, i2 = i1, b1 = b0, i1 = i0, b0 = m, i0 = id;
            else if (m < b1) b2 = b1, i2 = i1, b1 = m, i1 = id;
            else b2 = m, i2 = id;
            if (lane == owner) dist = FLT_MAX;
        }
    };
    auto scan_box = [&](int b) {
        const int s = b * (KNN_BOX / KNN_SUB) + lane;
        const bool in = lane < KNN_BOX / KNN_SUB && s < ns;
        const float ds = in ? dist_box_point(sub + 8 * (size_t)s, me) : FLT_MAX;
        const bool has = !GROUPED || (in && (knn_record_groups(sub + 8 * (size_t)s) & my_bit) != 0ull);
        unsigned long long m2 = __builtin_amdgcn_ballot_w64(in && has && !(ds > b2));
        while (m2 != 0ull) {
            const int l = (int)__builtin_ctzll(m2);
            m2 &= m2 - 1ull;
            const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ds), l));
            if (dl > b2) continue;  // the bound has tightened since the ballot
            scan_sub(b * (KNN_BOX / KNN_SUB) + l);
        }
    };
    // pass 1: the level-1 box nearest to the query
    int bmin = 0;
    {
        float dmin = FLT_MAX;
        int bl = 0;
        for (int bb = 0; bb < nb; bb += 64) {
            const int b = bb + lane;
            float d = b < nb ? dist_box_point(boxes + 8 * (size_t)b, me) : FLT_MAX;
            if (GROUPED && b < nb && (knn_record_groups(boxes + 8 * (size_t)b) & my_bit) == 0ull) d = FLT_MAX;  // (nothing of the query's group)
            if (d < dmin) dmin = d, bl = b;
        }
        const float m = wave_min_f(dmin);
        const int owner = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(dmin == m));
        bmin = __builtin_amdgcn_readlane(bl, owner);
    }
    scan_box(bmin);
    // pass 2: every other box that cannot be excluded, group by group (runs of 64 boxes = 65 536 sorted points: a group that cannot hold
    // a better candidate is skipped whole — 31 group tests instead of 1954 box tests against a 2 M-point map)
    const int ng = (nb + 63) / 64;
    for (int gg = 0; gg < ng; gg += 64) {
        const int gr = gg + lane;
        const float dg = gr < ng ? dist_box_point(groups + 8 * (size_t)gr, me) : FLT_MAX;
        const bool hasg = !GROUPED || (gr < ng && (knn_record_groups(groups + 8 * (size_t)gr) & my_bit) != 0ull);
        unsigned long long mg = __builtin_amdgcn_ballot_w64(gr < ng && hasg && !(dg > b2));
        while (mg != 0ull) {
            const int lg = (int)__builtin_ctzll(mg);
            mg &= mg - 1ull;
            const float dgl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dg), lg));
            if (dgl > b2) continue;  // the bound has tightened since the ballot
            const int b = (gg + lg) * 64 + lane;
            const bool in = b < nb && b != bmin;
            const float d = in ? dist_box_point(boxes + 8 * (size_t)b, me) : FLT_MAX;
            const bool hasb = !GROUPED || (in && (knn_record_groups(boxes + 8 * (size_t)b) & my_bit) != 0ull);
            unsigned long long m1 = __builtin_amdgcn_ballot_w64(in && hasb && !(d > b2));
            while (m1 != 0ull) {
                const int l = (int)__builtin_ctzll(m1);
                m1 &= m1 - 1ull;
                const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), l));
                if (dl > b2) continue;
                scan_box((gg + lg) * 64 + l);
            }
        }
    }
    if (lane == 0) {  // (a slot no reference made it into: FLT_MAX / -1, whatever the starting bound was)
        dist2[3 * q + 0] = i0 >= 0 ? b0 : FLT_MAX, dist2[3 * q + 1] = i1 >= 0 ? b1 : FLT_MAX, dist2[3 * q + 2] = i2 >= 0 ? b2 : FLT_MAX;
        idx3[3 * q + 0] = i0, idx3[3 * q + 1] = i1, idx3[3 * q + 2] = i2;
    }
}

}  // namespace

size_t dqo_knn3_ws_bytes(int P) { return knn_ws(nullptr, P).total; }

// bounding box -> Morton keys -> sort -> gather into Morton order (-> boxes)
// fine: morton_fine_kernel's 39-bit codes (the query search's reference set) instead of the reference's 30-bit ones
static int knn_build(int P, const float* xyz, const KnnWs& w, bool with_boxes, hipStream_t s, bool fine = false, const int32_t* group = nullptr) {
    fine = fine && P < (1 << FINE_IDX_BITS);
    const int P2 = next_pow2(P < SORT_RUN ? SORT_RUN : P);
    {   // (the partial boxes live at the start of the key array: P2 >= 4096 keys = 32 KB, and the keys are written after the fold)
        const int nb = std::min(BBOX_BLOCKS_MAX, (P + 1023) / 1024);
        float* partial = reinterpret_cast<float*>(w.keys);
        DQO_LAUNCH("bbox_kernel", bbox_kernel, dim3(nb), dim3(1024), s, P, xyz, partial, group);
        DQO_LAUNCH("bbox_fold_kernel", bbox_fold_kernel, dim3(1), dim3(512), s, nb, partial, w.bbox);
    }
    if (fine) DQO_LAUNCH("morton_kernel", morton_fine_kernel, dim3((P2 + 255) / 256), dim3(256), s, P, P2, xyz, w.bbox, w.keys, group);
    else DQO_LAUNCH("morton_kernel", morton_kernel, dim3((P2 + 255) / 256), dim3(256), s, P, P2, xyz, w.bbox, w.keys);
#ifdef KNN_BITONIC_SORT
    const int runs = P2 / SORT_RUN;
    DQO_LAUNCH("bitonic_lds_kernel", bitonic_lds_kernel, dim3(runs), dim3(SORT_T), s, w.keys, SORT_RUN, 1);
    for (int k = SORT_RUN * 2; k <= P2; k <<= 1) {
        for (int j = k >> 1; j >= SORT_RUN; j >>= 1) {
            DQO_LAUNCH("bitonic_global_kernel", bitonic_global_kernel, dim3((P2 / 2 + 255) / 256), dim3(256), s, w.keys, P2 / 2, k, j);
        }
        DQO_LAUNCH("bitonic_lds_kernel", bitonic_lds_kernel, dim3(runs), dim3(SORT_T), s, w.keys, k, 0);
    }
#else
    {   // the Morton code sits in bits 32..61 of the key: four stable 8-bit passes; the second key buffer is `sorted` (16 B per point,
        // written by the gather only after the sort); an even number of passes leaves the result in w.keys
        const int nblk = (P + RDX_BLK - 1) / RDX_BLK;
        uint64_t* a = w.keys;
        uint64_t* b = reinterpret_cast<uint64_t*>(w.sorted);
        // (fine codes: 39 bits above the 25-bit index — five passes; an even pass count leaves the result in w.keys, so a sixth pass
        // sorts by the top byte once more: a stable sort of sorted keys by a prefix of the key moves nothing but the buffer)
        const int n_pass = fine ? 6 : 4, shift0 = fine ? FINE_IDX_BITS : 32;
        for (int pass = 0; pass < n_pass; pass++) {
            const int shift = pass < 5 ? shift0 + 8 * pass : 56;
            DQO_LAUNCH("radix_hist_kernel", radix_hist_kernel, dim3(nblk), dim3(RDX_T), s, P, a, shift, w.hist, nblk);
            DQO_LAUNCH("radix_scan_kernel", radix_scan_kernel, dim3(256), dim3(256), s, w.hist, nblk, w.hist + (size_t)256 * nblk);
            DQO_LAUNCH("radix_scatter_kernel", radix_scatter_kernel, dim3(nblk), dim3(RDX_T), s, P, a, b, shift, w.hist, nblk,
                       w.hist + (size_t)256 * nblk);
            std::swap(a, b);
        }
    }
#endif
    DQO_LAUNCH("gather_sorted_kernel", gather_sorted_kernel, dim3((P + 255) / 256), dim3(256), s, P, xyz, w.keys, w.sorted,
               fine ? ((1ull << FINE_IDX_BITS) - 1ull) : 0xffffffffull, group);
    if (with_boxes) {
        const int nb = (P + KNN_BOX - 1) / KNN_BOX;
        DQO_LAUNCH("box_minmax_kernel", box_minmax_kernel, dim3(nb), dim3(256), s, P, w.sorted, w.boxes, group != nullptr ? 1 : 0);
    }
    return DQO_OK;
}

int dqo_launch_knn3(int P, const float* xyz, float* mean_d2, int32_t* idx3, void* ws, size_t ws_bytes, hipStream_t s) {
    (void)ws_bytes;
    KnnWs w = knn_ws(ws, P);
    int rc = knn_build(P, xyz, w, true, s);
    if (rc) return rc;
    const int ns = (P + KNN_SUB - 1) / KNN_SUB;
    DQO_LAUNCH("sub_minmax_kernel", sub_minmax_kernel, dim3((ns * 64 + 255) / 256), dim3(256), s, P, w.sorted, w.sub);
    DQO_LAUNCH("knn_scan_kernel", knn_scan_kernel, dim3((P + 255) / 256), dim3(256), s, P, w.sorted, w.boxes, w.sub, mean_d2, idx3);
    return DQO_OK;
}

size_t dqo_knn3_query_ws_bytes(int Q, int R) { return knn_ws(nullptr, Q).total + knn_ws(nullptr, R).total; }

int dqo_launch_knn3_query(int Q, const float* q_xyz, int R, const float* r_xyz, float* dist2, int32_t* idx3, void* ws, size_t ws_bytes,
                          hipStream_t s, float bound2, const int32_t* q_group, const int32_t* r_group, const float* group_box) {
    (void)ws_bytes;
    KnnWs wr = knn_ws(ws, R);
    int rc = knn_build(R, r_xyz, wr, true, s, true, r_group);
    if (rc) return rc;
    const int ns = (R + KNN_SUB - 1) / KNN_SUB;
    DQO_LAUNCH("sub_minmax_kernel", sub_minmax_kernel, dim3((ns * 64 + 255) / 256), dim3(256), s, R, wr.sorted, wr.sub);
    const int nb = (R + KNN_BOX - 1) / KNN_BOX, ng = (nb + 63) / 64;
    DQO_LAUNCH("group_minmax_kernel", group_minmax_kernel, dim3((ng * 64 + 255) / 256), dim3(256), s, nb, wr.boxes, wr.groups);
    if (q_group != nullptr)
        DQO_LAUNCH("knn_query_wave_kernel", knn_query_wave_kernel<true>, dim3(((size_t)Q * 64 + 255) / 256), dim3(256), s, Q, q_xyz, R, wr.sorted,
                   wr.boxes, wr.sub, wr.groups, dist2, idx3, bound2, q_group, group_box);
    else
        DQO_LAUNCH("knn_query_wave_kernel", knn_query_wave_kernel<false>, dim3(((size_t)Q * 64 + 255) / 256), dim3(256), s, Q, q_xyz, R, wr.sorted,
                   wr.boxes, wr.sub, wr.groups, dist2, idx3, bound2, q_group, group_box);
    return DQO_OK;
}
