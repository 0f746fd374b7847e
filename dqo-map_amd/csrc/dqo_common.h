// Shared device/host definitions for libdqoraster.so (gfx950 only: wave64, 160 KB LDS, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dqo_raster.h"

#define DQO_TILE 16  // reference BLOCK_X = BLOCK_Y = 16 (cuda_rasterizer/config.h:15-16); part of the op's semantics
#define DQO_WAVE 64
#define DQO_SPREAD 64  // lines the per-block statistics atomics of K1 are spread over
#define DQO_GATE_OBJECTS 64  // object ids of the per-object loss tap lie in [0, 64) (DqoLossTap.per_object)
#define DQO_OBJ_SPREAD 16    // copies of the per-object loss counters the forward's atomics are spread over
// The per-tile atomic counters (histogram, emit cursors) are spread one per DQO_TSTRIDE words: device-scope atomics execute
// memory-side on MI355X, and counters sharing a line / channel serialise there.
#ifndef DQO_TSTRIDE
#define DQO_TSTRIDE 64
#endif

// ---- error plumbing -------------------------------------------------------------------------------------------
void dqo_set_error(const char* fmt, ...);
#define DQO_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            dqo_set_error(__VA_ARGS__);   \
            return DQO_ERR_INVALID_ARG;   \
        }                                 \
    } while (0)
#define DQO_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            dqo_set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
            return DQO_ERR_LAUNCH;                                                   \
        }                                                                            \
    } while (0)
#define DQO_CHECK_LAUNCH() DQO_CHECK_HIP(hipGetLastError())

static inline size_t dqo_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- optional per-kernel timing (dqo_profile_enable): HIP events recorded on the launch stream around every kernel ----
extern int g_dqo_profile_on;
void dqo_profile_before(const char* name, hipStream_t s);
void dqo_profile_after(hipStream_t s);
#define DQO_LAUNCH(name, kernel, grid, block, stream, ...)                       \
    do {                                                                         \
        if (g_dqo_profile_on) dqo_profile_before(name, stream);                  \
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);         \
        if (g_dqo_profile_on) dqo_profile_after(stream);                         \
        DQO_CHECK_LAUNCH();                                                      \
    } while (0)

// ... with dynamic LDS
#define DQO_LAUNCH_SMEM(name, kernel, grid, block, smem_bytes, stream, ...)       \
    do {                                                                         \
        if (g_dqo_profile_on) dqo_profile_before(name, stream);                  \
        hipLaunchKernelGGL(kernel, grid, block, smem_bytes, stream, __VA_ARGS__); \
        if (g_dqo_profile_on) dqo_profile_after(stream);                         \
        DQO_CHECK_LAUNCH();                                                      \
    } while (0)

// ---- context buffer layouts (private) -------------------------------------------------------------------------
// geom buffer: header + per-Gaussian SoA tables, every table 256-B aligned.
struct DqoGeomLayout {
    DqoRastHeader* header;   // 256 B reserved
    uint32_t* counters;      // [16] device scalars: [0] instance total (slot allocator of the packed-list mode), [1] long-list queue,
                             //      [3..5] split-list queue / tickets, [6] the forward's list_split, [7] bucket mode: a slot region ran out,
                             //      [8] frame_prezeroed was promised but the per-frame scalars were not zero (preprocess_kernel, bin_count_kernel<true>),
                             //      [9] DQO_CLEARED_STAMP: the previous frame's tail has cleared scalars, tile histogram and flags (taken by bin_count_kernel<true>)
    uint32_t* spread;        // [DQO_SPREAD][64] statistics counters spread over DQO_SPREAD lines (same-address atomics serialise
                             //       memory-side): word 0 = visible Gaussians, word 1 = (Gaussian, tile) pairs in the tile rects,
                             //       words 2, 3 = longest list / non-empty tiles (keep_order frames), word 4 = bucket mode: the slot
                             //       allocator of region `line` (bin_count_kernel), words 8..15 = loss tap
    unsigned long long* obj_tap;  // [DQO_OBJ_SPREAD][DQO_GATE_OBJECTS][4] per-object loss tap: colour error sum, mask pixels, depth error
                                  //   sum, valid depth pixels (2^-32 fixed point / counts), zeroed with the header when the tap is per object
    float4* conic_opacity;   // [P] (conic.x, conic.y, conic.z, opacity)           forward.cu:343
    float4* xy_depth;        // [P] (pix.x, pix.y, depth = p_view.z, bits(radius) — or, with a DqoObjectGate, bits(object id))  forward.cu:339-341
    float4* rgb_smax;        // [P] (r, g, b, max(scale)*scale_mod)                 forward.cu:333-335, 73
    float4* normal_c;        // [P] (n_c.xyz, n_c . p_c)   surfel normal in camera space, hoisted out of the blend loop
    float4* point_c;         // [P] (p_c.xyz, max raw scale)   forward.cu:782-783, backward.cu:1009
    float4* drgb_dir;        // [P][3] d(SH colour)/d(view direction) (dqo_sh_dir_grad, backward.cu:168-258): (dRGBdx[3], -), (dRGBdy[3], -),
                             //        (dRGBdz[3], -) — evaluated by the forward, read by the backward's per-Gaussian chain
    uint2* rect16;           // [P] (minx | maxx<<16, miny | maxy<<16)
    uint32_t* tiles_touched; // [P]
    uint32_t* slot_base;     // [P] first gaussian-major instance slot of this Gaussian
    uint8_t* clamped;        // [P] bit0..2 = SH colour channel clamped at 0 (forward.cu:151-153)
    float4* grad_sum;        // [P][4] backward only: summed gradient record (DqoGradRec) of each Gaussian with instances
    size_t total;
};

static inline DqoGeomLayout dqo_geom_layout(void* base, int64_t P) {
    DqoGeomLayout L;
    char* p = (char*)base;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += dqo_align_up(bytes, 256);
        return r;
    };
    L.header = (DqoRastHeader*)take(256);
    L.counters = (uint32_t*)take(256);
    L.spread = (uint32_t*)take(256 * DQO_SPREAD);
    L.obj_tap = (unsigned long long*)take(sizeof(unsigned long long) * 4 * DQO_GATE_OBJECTS * DQO_OBJ_SPREAD);  // (directly after spread)
    L.conic_opacity = (float4*)take(sizeof(float4) * P);
    L.xy_depth = (float4*)take(sizeof(float4) * P);
    L.rgb_smax = (float4*)take(sizeof(float4) * P);
    L.normal_c = (float4*)take(sizeof(float4) * P);
    L.point_c = (float4*)take(sizeof(float4) * P);
    L.drgb_dir = (float4*)take(sizeof(float4) * 3 * P);
    L.rect16 = (uint2*)take(sizeof(uint2) * P);
    L.tiles_touched = (uint32_t*)take(sizeof(uint32_t) * P);
    L.slot_base = (uint32_t*)take(sizeof(uint32_t) * P);
    L.clamped = (uint8_t*)take(P);
    L.grad_sum = (float4*)take(sizeof(float4) * 4 * P);
    L.total = (size_t)(p - (char*)base);
    return L;
}

// DqoLossTap as the kernels get it (by value); scale == nullptr: no tap.  The per-frame sums live in the geometry buffer's spread
// lines (words 8..15 of each of the DQO_SPREAD lines, as four 64-bit counters: colour error sum and depth error sum in 2^-32 fixed
// point, the two pixel counts) — all zeroed with the header by the forward's zero fill.
struct DqoTapDev {
    const float* gt_color;
    const float* gt_depth;
    const uint8_t* mask;
    const float* out_color;
    const float* out_depth;
    float color_weight, depth_weight, add_depth_thres;
    float* loss_out;
    float* scale;
    int per_object;  // DqoLossTap.per_object
};
// DqoObjectGate as the kernels get it (by value); gobj == nullptr: no gate
struct DqoGateDev {
    const int32_t* gobj;  // [P]
    const int32_t* pobj;  // [H*W]
};
static inline DqoGateDev dqo_gate_dev(const DqoObjectGate* g) {
    DqoGateDev d;
    d.gobj = g ? g->gaussian_object : nullptr, d.pobj = g ? g->pixel_object : nullptr;
    return d;
}
static inline DqoTapDev dqo_tap_dev(const DqoLossTap* t) {
    DqoTapDev d;
    d.per_object = 0;
    if (t == nullptr) {
        d.gt_color = d.gt_depth = d.out_color = d.out_depth = nullptr, d.mask = nullptr, d.loss_out = d.scale = nullptr;
        d.color_weight = d.depth_weight = d.add_depth_thres = 0.f;
        return d;
    }
    d.per_object = t->per_object;
    d.gt_color = t->gt_color, d.gt_depth = t->gt_depth, d.mask = t->render_mask, d.out_color = t->out_color, d.out_depth = t->out_depth;
    d.color_weight = t->color_weight, d.depth_weight = t->depth_weight, d.add_depth_thres = t->add_depth_thres;
    d.loss_out = t->loss_out, d.scale = t->grad_scale;
    return d;
}
constexpr double DQO_TAP_FIXED = 4294967296.0;  // 2^32
#ifdef __HIPCC__
// The value of lane ^ D (D a power of two) without the LDS crossbar: a ds_bpermute costs 20-60 cycles per dependent use and shares the
// CU's LDS pipe; DPP moves 3-5, the gfx950 row / half swaps 4-9 (tools/ubench_valu.hip).  D = 1, 2: DPP quad_perm; D = 4: row_shl:4 /
// row_shr:4 on alternate banks; D = 8: row_ror:8; D = 16, 32: v_permlane16/32_swap of the register with a copy of itself.
typedef unsigned dqo_swap_uint2 __attribute__((ext_vector_type(2)));
template <int D>
__device__ __forceinline__ uint32_t dqo_lane_xor(uint32_t x, int lane) {
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);
    else if constexpr (D == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0x5, false);  // banks 0, 2 read lane + 4
        return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);  // banks 1, 3 read lane - 4
    } else if constexpr (D == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);
    else if constexpr (D == 16) {
        const dqo_swap_uint2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // r.x: even rows' values, r.y: odd rows'
        return (lane & 16) ? r.x : r.y;
    } else {
        static_assert(D == 32, "power of two up to 32");
        const dqo_swap_uint2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // r.x: lower half's values, r.y: upper half's
        return (lane & 32) ? r.x : r.y;
    }
}
// wave64 float sum in the order of the xor butterfly 32, 16, 8, 4, 2, 1 (every lane ends with the same total, bit for bit what the
// __shfl_xor loop of that order gives), and integer maximum — vector instructions only
__device__ __forceinline__ float dqo_wave_sum_xor(float x, int lane) {
    x += __uint_as_float(dqo_lane_xor<32>(__float_as_uint(x), lane));
    x += __uint_as_float(dqo_lane_xor<16>(__float_as_uint(x), lane));
    x += __uint_as_float(dqo_lane_xor<8>(__float_as_uint(x), lane));
    x += __uint_as_float(dqo_lane_xor<4>(__float_as_uint(x), lane));
    x += __uint_as_float(dqo_lane_xor<2>(__float_as_uint(x), lane));
    x += __uint_as_float(dqo_lane_xor<1>(__float_as_uint(x), lane));
    return x;
}
__device__ __forceinline__ uint32_t dqo_wave_max_u32(uint32_t x, int lane) {
    x = max(x, dqo_lane_xor<32>(x, lane)), x = max(x, dqo_lane_xor<16>(x, lane)), x = max(x, dqo_lane_xor<8>(x, lane));
    x = max(x, dqo_lane_xor<4>(x, lane)), x = max(x, dqo_lane_xor<2>(x, lane)), x = max(x, dqo_lane_xor<1>(x, lane));
    return x;
}
__device__ __forceinline__ uint32_t dqo_wave_sum_u32(uint32_t x, int lane) {
    x += dqo_lane_xor<32>(x, lane), x += dqo_lane_xor<16>(x, lane), x += dqo_lane_xor<8>(x, lane);
    x += dqo_lane_xor<4>(x, lane), x += dqo_lane_xor<2>(x, lane), x += dqo_lane_xor<1>(x, lane);
    return x;
}
// The two pixel counts of the frame's loss tap (mask pixels, valid depth pixels: at most W x H < 2^32 each) — all a wave of the
// backward needs for its gradient scales; the sums of dqo_tap_totals below are for the report
__device__ __forceinline__ void dqo_tap_counts(const uint32_t* spread, int lane, uint32_t& n_col, uint32_t& n_dep) {
    uint32_t a = 0u, b = 0u;
    for (int j = lane; j < DQO_SPREAD; j += 64) {
        const unsigned long long* l = reinterpret_cast<const unsigned long long*>(spread + (size_t)j * 64 + 8);
        a += (uint32_t)l[1], b += (uint32_t)l[3];
    }
    n_col = dqo_wave_sum_u32(a, lane), n_dep = dqo_wave_sum_u32(b, lane);
}
// The frame totals of the loss tap, by one wave: lane j reads line j, the wave adds up (fixed order).  tot[0..3] = colour error sum,
// mask pixels, depth error sum, valid depth pixels.  Called a kernel boundary after the forward blend kernel wrote them.
__device__ __forceinline__ void dqo_tap_totals(const uint32_t* spread, int lane, double tot[4]) {
    unsigned long long t[4] = {0ull, 0ull, 0ull, 0ull};
    for (int j = lane; j < DQO_SPREAD; j += 64) {  // (any DQO_SPREAD: the forward scatters to blockIdx.x % DQO_SPREAD)
        const unsigned long long* l = reinterpret_cast<const unsigned long long*>(spread + (size_t)j * 64 + 8);
#pragma unroll
        for (int c = 0; c < 4; c++) t[c] += l[c];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int c = 0; c < 4; c++) t[c] += (unsigned long long)__shfl_xor((long long)t[c], off);
    tot[0] = (double)t[0] / DQO_TAP_FIXED, tot[1] = (double)t[1], tot[2] = (double)t[2] / DQO_TAP_FIXED, tot[3] = (double)t[3];
}
// A wave's float sum as a 2^-32 fixed-point addend.  A sum the format cannot hold — NaN, an infinity, anything beyond 2^31 — becomes
// the largest addend (it poisons the loss by design, as a NaN poisons the float version); the float -> integer conversion of such a
// value would be undefined.
__device__ __forceinline__ unsigned long long dqo_tap_fixed(float s) {
    return (s >= 0.f && s < 2147483648.f) ? (unsigned long long)((double)s * DQO_TAP_FIXED + 0.5) : ~0ull >> 1;
}
#endif

// image buffer: per-tile tables + per-pixel forward->backward state.
// DqoRastCtx.list_split as the kernels get it: 0 = off, otherwise the list length above which a list is shared between eight waves
// (at least one chunk per wave makes no sense below 64 entries)
static inline int dqo_list_split(const DqoRastCtx* ctx) { return ctx->list_split <= 0 ? 0 : (ctx->list_split < 64 ? 64 : ctx->list_split); }
struct DqoImageLayout {
    uint32_t* tile_count;   // [T] instances per tile (atomic histogram, K1)
    uint32_t* tile_flag;    // [T] 1 = some Gaussian's reference rect covers the tile but all such instances were culled as dead
    uint32_t* tile_cursor;  // [T] emit cursor
    uint2* ranges;          // [T] [start, end) into the sorted list (rasterizer_impl.cu:120-142)
    uint32_t* walk4;        // [4T] per (tile, 8x8 quadrant): entries the backward must walk = max over the quadrant's pixels of
                            //      max(n_contrib, hit position)
    uint32_t* tile_order;   // [8][ceil(T/8)] tile ids per XCD group (block b serves group b % 8), longest list first; ~0 = unused
    uint4* slot_info;       // [8][ceil(T/8)] per tile_order slot, written by tile_sort_wave_kernel of THIS frame: (tile id or ~0, list
                            //      start, list end, 0) — the blend kernels' waves read their slot's record in ONE round instead of
                            //      tile_order -> ranges in two (a wave's head is a chain of dependent rounds)
    float* final_T;         // [HW] end_T  (forward.cu:849)
    uint32_t* n_contrib;    // [HW] last contributor, 1-based (forward.cu:850)
    uint32_t* hit_pos;      // [HW] bits 0..30: 1-based list position of the Gaussian that fixed the depth, 0 if none;
                            //      bit 31: the backward's ray/plane-depth branch applies to it (backward.cu:1016)
    uint32_t* long_tiles;   // [T] queue of the tiles whose lists are too long for tile_sort_wave_kernel (count: geom counters[1])
    uint32_t* split_tiles;  // [T] DqoRastCtx.list_split: queue of the tiles whose lists the forward blend shares between eight waves (longer
                            //     than list_split entries; count: geom counters[4], the blend's work ticket: counters[3])
    size_t total;
};

static inline DqoImageLayout dqo_image_layout(void* base, int W, int H) {
    DqoImageLayout L;
    const size_t gx = (W + DQO_TILE - 1) / DQO_TILE, gy = (H + DQO_TILE - 1) / DQO_TILE;
    const size_t T = gx * gy, HW = (size_t)W * H;
    char* p = (char*)base;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += dqo_align_up(bytes, 256);
        return r;
    };
    L.tile_count = (uint32_t*)take(4 * T * DQO_TSTRIDE);
    L.tile_flag = (uint32_t*)take(4 * T);  // directly after tile_count: both are zeroed by one fill
    L.tile_cursor = (uint32_t*)take(4 * T * DQO_TSTRIDE);
    L.ranges = (uint2*)take(8 * T);
    L.walk4 = (uint32_t*)take(16 * T);
    L.tile_order = (uint32_t*)take(4 * 8 * ((T + 7) / 8));
    L.slot_info = (uint4*)take(16 * 8 * ((T + 7) / 8));
    L.final_T = (float*)take(4 * HW);
    L.n_contrib = (uint32_t*)take(4 * HW);
    L.hit_pos = (uint32_t*)take(4 * HW);
    L.long_tiles = (uint32_t*)take(4 * T);
    L.split_tiles = (uint32_t*)take(4 * T);
    L.total = (size_t)(p - (char*)base);
    return L;
}

// binning buffer: per-instance tables.
struct DqoBinLayout {
    uint4* recs;          // [cap] unsorted list entries, grouped per tile segment: (gaussian id, depth bits, gaussian-major slot, 0) —
                          //       the sort key is (y << 32 | x), the payload z.  ONE 16-byte record per entry: the binning pass
                          //       writes it to a random list position — one partial-line write instead of the two that separate key /
                          //       slot arrays cost (bin_count_kernel 62.4 -> 59.8 us on cfg 3, -2 % per iteration on cfg 5)
    uint32_t* point_list; // [cap] sorted gaussian ids   (binningState.point_list)
    uint32_t* slot_list;  // [cap] sorted slots (where the backward stores this instance's gradient record)
    uint8_t* live_q;      // [4][cap] per (tile quadrant, sorted instance): 1 iff the forward acted on the instance in that quadrant
    uint2* slot_info;     // [cap] per gaussian-major slot: (tile, rank inside the tile's segment), bin_count -> bin_place
    uint32_t* slot_gid;   // [cap] per gaussian-major slot: Gaussian id
    uint32_t* rec_valid;  // [cap] backward: byte q of word `slot` = 1 iff the partial gradient record (slot, quadrant q) was written;
                          //       zeroed by bin_place_kernel / bin_count_kernel (the forward), so the backward needs no memset
    int64_t list_cap;     // entries of the list-indexed tables above (keys .. live_q): inst capacity, or tiles x bucket
    int32_t bucket;       // DqoRastCtx::tile_bucket_capacity
    size_t total;
};
static inline int64_t dqo_list_cap(int64_t cap, int W, int H, int bucket) {
    const int64_t T = (int64_t)((W + DQO_TILE - 1) / DQO_TILE) * ((H + DQO_TILE - 1) / DQO_TILE);
    return bucket > 0 ? T * bucket : cap;
}

static inline DqoBinLayout dqo_bin_layout(void* base, int64_t cap, int64_t list_cap, int bucket) {
    DqoBinLayout L;
    L.list_cap = list_cap, L.bucket = bucket;
    char* p = (char*)base;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += dqo_align_up(bytes, 256);
        return r;
    };
    L.recs = (uint4*)take(16 * (size_t)list_cap);
    L.point_list = (uint32_t*)take(4 * (size_t)list_cap);
    L.slot_list = (uint32_t*)take(4 * (size_t)list_cap);
    L.live_q = (uint8_t*)take(4 * (size_t)list_cap);
    L.slot_info = (uint2*)take(8 * (size_t)cap);
    L.slot_gid = (uint32_t*)take(4 * (size_t)cap);
    L.rec_valid = (uint32_t*)take(4 * (size_t)cap);
    L.total = (size_t)(p - (char*)base);
    return L;
}

// Per-instance gradient record written by the backward blend kernel (one per (tile, Gaussian) instance), summed per
// Gaussian in a fixed order by the per-Gaussian backward kernel: bitwise reproducible, no float atomics.
struct __attribute__((aligned(16))) DqoGradRec {
    float dcolor[3];   // dL/d rgb
    // pixel moments of q = G * dL/dalpha over the pixels that blended the instance; the opacity / conic / viewport factors that
    // turn them into dL/d(2D mean), dL/d(conic), dL/d(opacity) (backward.cu:964-994) are per-Gaussian constants and are
    // applied by gaussian_backward_kernel
    float m1[2];       // sum q dx, sum q dy
    float m2[3];       // sum q dx^2, sum q dx dy, sum q dy^2
    float m0;          // sum q
    // depth-hit sums over the pixels whose depth this instance fixed (backward.cu:997-1065); the per-Gaussian factors of
    // that gradient that are LINEAR (normal, view matrix, quaternion Jacobian) are applied once per Gaussian by
    // gaussian_backward_kernel:  hit[0] = sum dL/ddepth over pixels in the centre-depth branch,  hit[1] = sum of
    // dL/ddepth * ray.z / (n.ray) over pixels in the ray/plane branch,  hit[2..4] = sum of dL/ddepth * d(depth)/d(n_c) =
    // dL/ddepth * ray.z (nr p_c - np ray) / nr^2, evaluated per pixel with the reference's statements (backward.cu:1018-1041)
    float hit[5];
    float pad[2];
};
static_assert(sizeof(DqoGradRec) == 64, "record must be one 64-byte line");

// backward workspace: [cap][4] partial gradient records, one per (instance slot, tile quadrant); which of them exist is
// recorded in DqoBinLayout::rec_valid.
static inline size_t dqo_bwd_recs_bytes(int64_t cap) { return dqo_align_up(sizeof(DqoGradRec) * 4 * (size_t)(cap < 0 ? 0 : cap), 256); }
static inline size_t dqo_bwd_ws_bytes(int64_t cap) { return dqo_bwd_recs_bytes(cap) + 256; }

// Block -> Gaussian assignment of the two kernels whose cost depends on WHICH Gaussians share a block (bin_count_kernel: same-
// tile atomics; record_sum_kernel: slots per block).  A block of 256 threads owns 16 groups of 16 consecutive Gaussians taken
// from 16 distant parts of the index range (a 16 x G/16 transpose of the group index), so a spatially coherent storage order
// (an incrementally built map) is spread over the blocks the way a random order is, while every load still covers 16
// consecutive records.  Both kernels use the same mapping: a block's Gaussians get one contiguous run of instance slots.
// Returns >= P for the padding positions.
__host__ __device__ static inline int dqo_spread_index(int logical, int P) {
    const int G = (P + 15) >> 4;            // groups of 16
    const int rows = (G + 15) >> 4;         // groups per region
    const int L = logical >> 4;             // logical group
    const int phys = (L & 15) * rows + (L >> 4);
    return phys < G ? phys * 16 + (logical & 15) : P;
}

// number of 256-thread blocks that cover every logical position of dqo_spread_index
static inline int dqo_spread_blocks(int P) {
    const int G = (P + 15) >> 4, rows = (G + 15) >> 4;
    return (rows * 16 * 16 + 255) / 256;
}

// Per-view constants.  Scalars travel by value (kernarg -> SGPRs); the matrices stay device pointers because the
// reference API hands them over as device tensors (no host read, no sync) — kernels fetch them with scalar loads.
struct DqoView {
    const float* view;    // [16]
    const float* proj;    // [16]
    const float* campos;  // [3]
    const float* bg;      // [3]
    float tanfovx, tanfovy, focal_x, focal_y, cx, cy;
    float scale_mod, color_sigma, opaque_thr, depth_thr, normal_thr, T_thr;
    int W, H, gx, gy;
    int P, D, M;
    const uint8_t* row_flags;  // DqoRastInputs.row_flags (NULL = none): DQO_ROW_HIDDEN rows are culled by the per-Gaussian forward
};

// The inputs of the late part of the per-Gaussian forward (dqo_k1_late.h) for the extra blocks of a sort launch
struct DqoK1Late {
    DqoView v;
    const float *means3D, *scales, *rotations, *shs, *colors_precomp;
    int first_block;  // blocks [first_block, gridDim.x) of the launch do the late part, the launch's block size of Gaussians each
};

static inline DqoView dqo_make_view(const DqoRastParams* p, const DqoRastInputs* in) {
    DqoView v;
    v.view = in->viewmatrix;
    v.row_flags = in->row_flags;
    v.proj = in->projmatrix;
    v.campos = in->campos;
    v.bg = in->bg;
    v.tanfovx = p->tanfovx;
    v.tanfovy = p->tanfovy;
    v.focal_y = p->H / (2.0f * p->tanfovy);  // rasterizer_impl.cu:245-246
    v.focal_x = p->W / (2.0f * p->tanfovx);
    v.cx = p->cx;
    v.cy = p->cy;
    v.scale_mod = p->scale_modifier;
    v.color_sigma = p->color_sigma;
    v.opaque_thr = p->opaque_threshold;
    v.depth_thr = p->depth_threshold;
    v.normal_thr = p->normal_threshold;
    v.T_thr = p->T_threshold;
    v.W = p->W;
    v.H = p->H;
    v.gx = (p->W + DQO_TILE - 1) / DQO_TILE;
    v.gy = (p->H + DQO_TILE - 1) / DQO_TILE;
    v.P = p->P;
    v.D = p->D;
    v.M = p->M;
    return v;
}

// The per-frame scalars of the geometry buffer that must be zero when a frame starts: counters | statistics lines | (with a per-object
// loss tap) the per-object loss counters — one contiguous range behind the header.  The forward's zero fill clears header + range;
// dqo_rast_backward_adam's per-Gaussian kernel clears the range for the next frame (DqoRastCtx.frame_prezeroed).
static inline size_t dqo_frame_scalar_words(const DqoRastCtx* ctx) {
    const size_t obj_words = (ctx->loss_tap != nullptr && ctx->loss_tap->per_object)
                                 ? sizeof(unsigned long long) * 4 * DQO_GATE_OBJECTS * DQO_OBJ_SPREAD / 4 : 0;
    return (256 + 256 * DQO_SPREAD) / 4 + obj_words;
}

// counters[9] after dqo_rast_backward_adam's per-Gaussian kernel (which also clears the tile histogram + flags of the context's image
// buffer): the next frame on the context may start in bin_count_kernel<true> — no preprocess_kernel, no zero fill (dqo_fuse_k1).
#define DQO_CLEARED_STAMP 0xC1EA5EDu
// The early part of the per-Gaussian forward runs at the head of the binning kernel (dqo_k1_early.h) when the frame is pre-zeroed (a
// replayed iteration of the fused mapping step), the lists live in per-tile buckets (that caller's mode) and the late part has a home
// behind the sort kernels (dqo_k1_where != 0; 0 keeps everything in preprocess_kernel: the A/B baseline).  DQO_K1_FUSE=0 switches it off.
bool dqo_fuse_k1(const DqoRastParams* p, const DqoRastCtx* ctx);

// keep_order frames (DqoRastCtx.keep_tile_order: no tile_scan_kernel): the header that kernel would have written, from the spread statistics
// lines — one wave, behind the per-tile sort launch: the first block of tile_sort_kernel, or (no long-list launch: dqo_skip_long_sort) an
// extra block of blend_forward_kernel
__device__ __forceinline__ void dqo_header_from_spread(const DqoGeomLayout& g, int64_t capacity, int bucket, int lane) {
    uint32_t nv = 0, nc = 0, mx = 0, nt = 0, total = 0;
    for (int j = lane; j < DQO_SPREAD; j += 64) {
        const uint32_t* line = g.spread + (size_t)j * 64;
        nv += line[0], nc += line[1], mx = max(mx, line[2]), nt += line[3], total += line[4] + line[5];  // (5: list entries without a gradient slot)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nv += __shfl_xor((int)nv, off), nc += __shfl_xor((int)nc, off), nt += __shfl_xor((int)nt, off);
        total += __shfl_xor((int)total, off);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    }
    if (lane == 0) {
        // instances counted by bin_count_kernel's regional slot allocators (line word 4) = the sum of the list lengths
        DqoRastHeader h;
        h.num_rendered = total;
        h.num_tiles = nt;
        (void)capacity;
        h.overflow = (g.counters[7] != 0u || g.counters[8] != 0u || mx > (uint32_t)bucket) ? 1u : 0u;
        h.max_tile_count = mx;
        h.num_visible = nv;
        h.num_candidates = nc;
        h.stage = 2u, h.reserved = 0u;
        *g.header = h;
    }
}

// The long-list sort launch is dropped when no list can outgrow the per-tile sort (buckets of at most DQO_SORTW_CAP entries, tile order kept,
// the late part of the per-Gaussian forward not riding that launch, no list splitting); DQO_SKIP_LONG_SORT=0 keeps it.
bool dqo_skip_long_sort(const DqoRastParams* p, const DqoRastCtx* ctx);

// Zero fill of caller memory on the launch stream.  The library never uses hipMemsetAsync for this: as a memset NODE of a captured
// hipGraph the fill was observed (ROCm 7.2, gfx950) to write garbage on replay once other runtime activity had happened since the
// capture; a kernel node carries its arguments by value.  Defined in rast_forward.hip.
int dqo_launch_zero_words(uint32_t* p, size_t n_words, hipStream_t s);

// launchers (defined in the .hip files)
int dqo_launch_bin_count_k1(const DqoView& v, const DqoRastInputs* in, const DqoRastOutputs* out, const int32_t* gobj, const DqoGeomLayout& g,
                            const DqoImageLayout& img, const DqoBinLayout& bin, int64_t capacity, const unsigned long long* tile_objects,
                            hipStream_t s);
int dqo_launch_forward_prepare(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, hipStream_t s);
int dqo_launch_mark_header_stage0(const DqoRastParams* p, DqoRastCtx* ctx, hipStream_t s);
int dqo_launch_forward_render(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, hipStream_t s,
                              DqoRastHeader* header_host = nullptr, hipEvent_t header_event = nullptr);
int dqo_launch_backward(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx, const float* dL_dcolor,
                        const float* dL_ddepth, const int32_t* hit_image, DqoRastGrads* g, void* ws, size_t ws_bytes, hipStream_t s);
