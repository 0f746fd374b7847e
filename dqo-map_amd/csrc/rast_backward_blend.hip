// Backward blend kernel (K7) for gfx950 — replaces /root/reference/submodules/diff-gaussian-rasterizer-depth/
// cuda_rasterizer/backward.cu:808-1066 (renderCUDA_flat) and :100-148 (propagateRotationGrad).
//
// Structure: ONE wave64 per (tile, 8x8 quadrant) — four single-wave workgroups per 16x16 tile, no __syncthreads, no global
// float atomics (the reference: ~10 per (pixel, Gaussian) pair + 3-7 per hit pixel).
//   * Depth-hit sums first, once per pixel: every pixel has at most one entry that fixed its depth; the pixels of the quadrant
//     that share it are added up by their lowest lane through wave-private LDS and written as floats 9..13 of that
//     (quadrant, instance) record.
//   * The wave then walks its tile's list back to front in chunks of 64 positions, but only over the entries the forward marked
//     live for this quadrant: the live ones of a chunk are compacted with one ballot and their 16-byte records gathered into
//     wave-private LDS; the next chunk's live bytes are loaded while the current chunk is processed.
//   * Per live entry every lane evaluates its pixel with predicated (branch-free) arithmetic and produces nine sums-to-be (three
//     colour terms, six moments of q = G dL/dalpha).  Seven entries at a time, their 63 values go through ONE 64-value
//     reduce-scatter butterfly (v_permlane32/16_swap, bank-masked v_add_f32_dpp, quad_perm), after which lane 9 b + f holds float
//     f of entry b: those lanes store the 64-byte partial records at recs[slot * 4 + quadrant] and lanes 0..6 mark them valid
//     (1 = colour-path floats, 3 = depth-hit floats as well).
// record_sum_kernel adds a Gaussian's valid partial records in a fixed order: bitwise reproducible.  Instruction costs behind the
// choices (swap = 8 cycles, DPP = 4-5, plain = 3-4): tools/ubench_valu.hip, profiles/r01_ubench_valu.txt.
#include <cstdlib>
#include <type_traits>

#include "dqo_common.h"
#include "dqo_cull.h"

namespace {

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// wave64 sum, result broadcast as a wave-uniform value
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);       // row_half_mirror
    v += dpp_mov<0x140>(v);       // row_mirror          -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
    v += dpp_mov<0x143, 0xC>(v);  // row_bcast31 into rows 2,3 -> lanes 48..63 hold the wave total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum of a 32-bit word over the lanes of DPP row 0 (lanes 0..15), as a wave-uniform value
__device__ __forceinline__ uint32_t row0_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror: every lane of a row holds its row's sum
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
}

// Reduce-scatter butterfly over the wave for EIGHT values at once: afterwards every lane l holds the 64-lane total of
// v[l & 7].  Each of the first three stages halves the number of live values per lane (the lane keeps the half selected by
// one bit of its id and ships the other half to its partner), so the whole thing costs ~35 VALU instead of 8 x 11 for eight
// independent reductions.  Partners: xor 1 / xor 2 = DPP quad_perm, xor 4 = row_shl:4 | row_shr:4 split by bank mask,
// xor 8 = row_ror:8, xor 16 / 32 = ds_bpermute.
typedef unsigned dqo_uint2v __attribute__((ext_vector_type(2)));
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_mov_bank(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, BANK_MASK, false));
}
__device__ __forceinline__ float wave_reduce8(const float (&v)[8], int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
    float w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        w[i] = keep + dpp_mov<0xB1>(send);  // partner lane ^ 1
    }
    float x[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const float keep = b1 ? w[2 * m + 1] : w[2 * m], send = b1 ? w[2 * m] : w[2 * m + 1];
        x[m] = keep + dpp_mov<0x4E>(send);  // partner lane ^ 2
    }
    const float keep = b2 ? x[1] : x[0], send = b2 ? x[0] : x[1];
    float t = dpp_mov_bank<0x104, 0x5>(0.f, send);  // row_shl:4 -> lanes with bit2 = 0 read lane + 4
    t = dpp_mov_bank<0x114, 0xA>(t, send);          // row_shr:4 -> lanes with bit2 = 1 read lane - 4
    float y = keep + t;                              // partner lane ^ 4
    y += dpp_mov<0x128>(y);                          // row_ror:8  == lane ^ 8 inside a row of 16
    // lane ^ 16 and lane ^ 32: a swap of a register with its own copy leaves (this half, partner half) in the two results —
    // one VALU op instead of a ds_bpermute round trip through the LDS queue
    const dqo_uint2v r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    y = __uint_as_float(r16.x) + __uint_as_float(r16.y);
    const dqo_uint2v r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(r32.x) + __uint_as_float(r32.y);
}

// Reduce-scatter butterfly for SIXTY-FOUR values at once: afterwards lane l holds the 64-lane total of v[l].  Stage order
// = cheapest exchanges first, while the value count per lane is still high:
//   lane ^ 32, lane ^ 16 : v_permlane32_swap / v_permlane16_swap (gfx950) — the swap itself routes the two halves, so a pair
//                          costs one swap + one add, no select
//   lane ^ 8,  lane ^ 4  : DPP row_ror:8 / row_shl:4 | row_shr:4, the two lane classes are whole DPP banks (bank_mask)
//   lane ^ 2,  lane ^ 1  : DPP quad_perm with explicit selects
// 141 VALU for 64 sums (2.2 per sum; eight separate DPP reductions of one value each would cost ~11 per sum).
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_pick(float old, float v) {  // lanes of the enabled banks read v through CTRL, the others keep old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xF, BANK_MASK, false));
}
// stages lane ^ 8, ^ 4, ^ 2, ^ 1 of the reduce-scatter: 16 values per lane -> 1 (lane bits 3..0 select which)
__device__ __forceinline__ float wave_reduce_tail16(const float (&b)[16], int lane) {
    // lane bit 3 = DPP banks 2, 3 of every row, partner = row_ror:8: banks 0,1 take b[i] + partner's b[i], banks 2,3 take
    // b[i + 8] + partner's b[i + 8] — two bank-masked v_add_f32_dpp writing the two halves of one register (the builtin only
    // yields v_mov_dpp + select + add: 4 instructions).  Hand-written DPP: the leading s_nop covers the 2 wait states a DPP read
    // needs after the VALU write of its source, which the compiler cannot see inside the asm.
    float c[8];
#define DQO_S3(i, j) "v_add_f32_dpp %[c" #i "], %[b" #i "], %[b" #i "] row_ror:8 row_mask:0xf bank_mask:0x3\n" \
                     "v_add_f32_dpp %[c" #i "], %[b" #j "], %[b" #j "] row_ror:8 row_mask:0xf bank_mask:0xc\n"
    asm volatile("s_nop 1\n" DQO_S3(0, 8) DQO_S3(1, 9) DQO_S3(2, 10) DQO_S3(3, 11) DQO_S3(4, 12) DQO_S3(5, 13) DQO_S3(6, 14) DQO_S3(7, 15)
                 : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]),
                   [c6] "=&v"(c[6]), [c7] "=&v"(c[7])
                 : [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]), [b5] "v"(b[5]), [b6] "v"(b[6]),
                   [b7] "v"(b[7]), [b8] "v"(b[8]), [b9] "v"(b[9]), [b10] "v"(b[10]), [b11] "v"(b[11]), [b12] "v"(b[12]),
                   [b13] "v"(b[13]), [b14] "v"(b[14]), [b15] "v"(b[15]));
#undef DQO_S3
    // lane bit 2 = odd DPP banks; partner = lane + 4 (row_shl:4, even banks) / lane - 4 (row_shr:4, odd banks)
    float d[4];
#define DQO_S4(i, j) "v_add_f32_dpp %[d" #i "], %[c" #i "], %[c" #i "] row_shl:4 row_mask:0xf bank_mask:0x5\n" \
                     "v_add_f32_dpp %[d" #i "], %[c" #j "], %[c" #j "] row_shr:4 row_mask:0xf bank_mask:0xa\n"
    asm volatile("s_nop 1\n" DQO_S4(0, 4) DQO_S4(1, 5) DQO_S4(2, 6) DQO_S4(3, 7) "s_nop 1\n"
                 : [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [d2] "=&v"(d[2]), [d3] "=&v"(d[3])
                 : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]),
                   [c7] "v"(c[7]));
#undef DQO_S4
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    float e[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float keep = b1 ? d[i + 2] : d[i], send = b1 ? d[i] : d[i + 2];
        e[i] = keep + dpp_mov<0x4E>(send);  // partner lane ^ 2
    }
    const float keep = b0 ? e[1] : e[0], send = b0 ? e[0] : e[1];
    return keep + dpp_mov<0xB1>(send);  // partner lane ^ 1
}

__device__ __forceinline__ float wave_reduce64(const float (&v)[64], int lane) {
    float a[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {  // lanes 0..31 end up with the sum of v[i], lanes 32..63 with that of v[i + 32]
        const dqo_uint2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 32]), false, false);
        a[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {  // rows 0, 2 (lane bit 4 clear): a[i];  rows 1, 3: a[i + 16]
        const dqo_uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 16]), false, false);
        b[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    return wave_reduce_tail16(b, lane);
}

// The same butterfly for THIRTY-TWO values: lane l (and lane l ^ 32) ends up with the 64-lane total of v[l & 31].  The lane ^ 32
// exchange comes last, on the one value that is left (a swap of a register with its own copy + one add), instead of first on 32
// pairs: 73 VALU for 32 sums — more per sum than the 64-value form, but half the registers, which is what decides how many waves
// share a SIMD (the kernel is bound by per-wave instruction latency, not by VALU throughput: tools/ubench_valu.hip).
__device__ __forceinline__ float wave_reduce32(const float (&v)[32], int lane) {
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const dqo_uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
        b[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    const float y = wave_reduce_tail16(b, lane);
    const dqo_uint2v r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(r32.x) + __uint_as_float(r32.y);
}

// ---- per-row (16-lane) reduce-scatter for the row walk ------------------------------------------------------------------
// SIXTY-FOUR values per lane, reduced over the 16 lanes of each DPP row separately: afterwards lane s of a row holds the row's totals of
// v[4 s .. 4 s + 3].  Four halving stages (lane ^ 8, ^ 4: two bank-masked v_add_f32_dpp per output; lane ^ 2, ^ 1: quad_perm with
// selects), no permlane swap: 64 + 32 + 24 + 12 = 132 VALU for 4 x 63 sums — per row and entry what the 64-lane butterfly costs per
// wave and entry, but the four rows work on four DIFFERENT entries.
__device__ __forceinline__ void row_stage8(const float (&lo)[8], const float (&hi)[8], float (&c)[8]) {
    // lanes with bit 3 clear (DPP banks 0, 1) take lo[i] + the lo[i] of lane ^ 8, the others hi[i] + the partner's hi[i]
#define DQO_R8(i) "v_add_f32_dpp %[c" #i "], %[l" #i "], %[l" #i "] row_ror:8 row_mask:0xf bank_mask:0x3\n" \
                  "v_add_f32_dpp %[c" #i "], %[h" #i "], %[h" #i "] row_ror:8 row_mask:0xf bank_mask:0xc\n"
    asm volatile("s_nop 1\n" DQO_R8(0) DQO_R8(1) DQO_R8(2) DQO_R8(3) DQO_R8(4) DQO_R8(5) DQO_R8(6) DQO_R8(7)
                 : [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3]), [c4] "=&v"(c[4]), [c5] "=&v"(c[5]),
                   [c6] "=&v"(c[6]), [c7] "=&v"(c[7])
                 : [l0] "v"(lo[0]), [l1] "v"(lo[1]), [l2] "v"(lo[2]), [l3] "v"(lo[3]), [l4] "v"(lo[4]), [l5] "v"(lo[5]), [l6] "v"(lo[6]),
                   [l7] "v"(lo[7]), [h0] "v"(hi[0]), [h1] "v"(hi[1]), [h2] "v"(hi[2]), [h3] "v"(hi[3]), [h4] "v"(hi[4]), [h5] "v"(hi[5]),
                   [h6] "v"(hi[6]), [h7] "v"(hi[7]));
#undef DQO_R8
}
__device__ __forceinline__ void row_stage4(const float (&lo)[8], const float (&hi)[8], float (&d)[8]) {
    // lanes with bit 2 clear (even DPP banks) take lo[i] + the lo[i] of lane + 4, the others hi[i] + the hi[i] of lane - 4
#define DQO_R4(i) "v_add_f32_dpp %[d" #i "], %[l" #i "], %[l" #i "] row_shl:4 row_mask:0xf bank_mask:0x5\n" \
                  "v_add_f32_dpp %[d" #i "], %[h" #i "], %[h" #i "] row_shr:4 row_mask:0xf bank_mask:0xa\n"
    asm volatile("s_nop 1\n" DQO_R4(0) DQO_R4(1) DQO_R4(2) DQO_R4(3) DQO_R4(4) DQO_R4(5) DQO_R4(6) DQO_R4(7) "s_nop 1\n"
                 : [d0] "=&v"(d[0]), [d1] "=&v"(d[1]), [d2] "=&v"(d[2]), [d3] "=&v"(d[3]), [d4] "=&v"(d[4]), [d5] "=&v"(d[5]),
                   [d6] "=&v"(d[6]), [d7] "=&v"(d[7])
                 : [l0] "v"(lo[0]), [l1] "v"(lo[1]), [l2] "v"(lo[2]), [l3] "v"(lo[3]), [l4] "v"(lo[4]), [l5] "v"(lo[5]), [l6] "v"(lo[6]),
                   [l7] "v"(lo[7]), [h0] "v"(hi[0]), [h1] "v"(hi[1]), [h2] "v"(hi[2]), [h3] "v"(hi[3]), [h4] "v"(hi[4]), [h5] "v"(hi[5]),
                   [h6] "v"(hi[6]), [h7] "v"(hi[7]));
#undef DQO_R4
}
// ... and for THIRTY-TWO values (three entries per batch): lane s of a row ends up with the row's totals of v[2 s], v[2 s + 1]; 66 VALU
// for 27 sums — more per sum than the 64-value form, but half the registers (what decides how many waves share a SIMD)
__device__ __forceinline__ void row_reduce32(const float (&v)[32], float (&out)[2], int lane) {
    float c[16];
#pragma unroll
    for (int q = 0; q < 2; q++) {  // c[i] = v[i] (+) v[i + 16]
        float lo[8], hi[8], r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) lo[i] = v[8 * q + i], hi[i] = v[16 + 8 * q + i];
        row_stage8(lo, hi, r);
#pragma unroll
        for (int i = 0; i < 8; i++) c[8 * q + i] = r[i];
    }
    float d[8];
    {
        float lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 8; i++) lo[i] = c[i], hi[i] = c[8 + i];
        row_stage4(lo, hi, d);
    }
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    float e[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float keep = b1 ? d[i + 4] : d[i], send = b1 ? d[i] : d[i + 4];
        e[i] = keep + dpp_mov<0x4E>(send);  // partner lane ^ 2
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float keep = b0 ? e[i + 2] : e[i], send = b0 ? e[i] : e[i + 2];
        out[i] = keep + dpp_mov<0xB1>(send);  // partner lane ^ 1
    }
}
__device__ __forceinline__ void row_reduce64(const float (&v)[64], float (&out)[4], int lane) {
    float c[32];
#pragma unroll
    for (int q = 0; q < 4; q++) {  // c[i] = v[i] (+) v[i + 32], i = 8 q .. 8 q + 7
        float lo[8], hi[8], r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) lo[i] = v[8 * q + i], hi[i] = v[32 + 8 * q + i];
        row_stage8(lo, hi, r);
#pragma unroll
        for (int i = 0; i < 8; i++) c[8 * q + i] = r[i];
    }
    float d[16];
#pragma unroll
    for (int q = 0; q < 2; q++) {  // d[i] = c[i] (+) c[i + 16]
        float lo[8], hi[8], r[8];
#pragma unroll
        for (int i = 0; i < 8; i++) lo[i] = c[8 * q + i], hi[i] = c[16 + 8 * q + i];
        row_stage4(lo, hi, r);
#pragma unroll
        for (int i = 0; i < 8; i++) d[8 * q + i] = r[i];
    }
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float keep = b1 ? d[i + 8] : d[i], send = b1 ? d[i] : d[i + 8];
        e[i] = keep + dpp_mov<0x4E>(send);  // partner lane ^ 2
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float keep = b0 ? e[i + 4] : e[i], send = b0 ? e[i] : e[i + 4];
        out[i] = keep + dpp_mov<0xB1>(send);  // partner lane ^ 1
    }
}

__device__ __forceinline__ float3 pixel_ray_b(uint32_t px, uint32_t py, float fx, float fy, float cx, float cy) {
#pragma clang fp contract(off)
    float rx = ((float)px - cx) / fx, ry = ((float)py - cy) / fy, rz = 1.0f;
    const float n = 1.0f / sqrtf(rx * rx + ry * ry + rz * rz);
    return make_float3(rx * n, ry * n, rz * n);
}

// DqoLossTap.per_object: the loss is the sum of the objects' own masked losses, so the report adds up one term per object (lane o =
// object o, its counters summed over the spread copies); every pixel's gradient scale is its OWNER's (blend_backward_kernel).  One
// whole wave, all lanes active.
__device__ __forceinline__ void tap_report_per_object(const DqoGeomLayout& g, const DqoTapDev& tap) {
    const int o = (int)threadIdx.x;  // DQO_GATE_OBJECTS == 64 == the wave
    static_assert(DQO_GATE_OBJECTS == 64, "one lane per object id");
    unsigned long long t[4] = {0ull, 0ull, 0ull, 0ull};
    for (int j = 0; j < DQO_OBJ_SPREAD; j++) {
        const unsigned long long* l = g.obj_tap + ((size_t)j * DQO_GATE_OBJECTS + (size_t)o) * 4;
#pragma unroll
        for (int c = 0; c < 4; c++) t[c] += l[c];
    }
    const double s_c = (double)t[0] / DQO_TAP_FIXED, s_d = (double)t[2] / DQO_TAP_FIXED;
    const float n_col = fmaxf((float)t[1], 1.f), n_dep = fmaxf((float)t[3], 1.f);
    const float color_o = (float)(s_c / (3.0 * (double)n_col)), depth_o = (float)(s_d / (double)n_dep);
    const float color_loss = wave_sum(color_o), depth_loss = wave_sum(depth_o);
    const float total = wave_sum(tap.depth_weight * depth_o + tap.color_weight * color_o);
    const float r4 = wave_sum((float)s_c), r5 = wave_sum((float)t[1]), r6 = wave_sum((float)s_d), r7 = wave_sum((float)t[3]);
    if (threadIdx.x == 0) {
        tap.loss_out[0] = total, tap.loss_out[1] = color_loss, tap.loss_out[2] = depth_loss, tap.loss_out[3] = 0.f;
        tap.loss_out[4] = r4, tap.loss_out[5] = r5, tap.loss_out[6] = r6, tap.loss_out[7] = r7;
    }
}

constexpr int BWD_THREADS = 64;
// LDS of one wave: the gathered records of its chunk's live entries (three float4 tables, three word tables) and the depth-hit sums
constexpr int BWD_BLK = 3 * BWD_THREADS * 4 + 3 * BWD_THREADS + 5 * BWD_THREADS;  // words: 1280 = 5120 B
constexpr int BWD_XCH = 5 * BWD_THREADS;                                            // words of one wave's pass-1 result (SEGS > 1)
__device__ __forceinline__ void row_reduce(const float (&v)[64], float (&out)[4], int lane) { row_reduce64(v, out, lane); }
__device__ __forceinline__ void row_reduce(const float (&v)[32], float (&out)[2], int lane) { row_reduce32(v, out, lane); }
__device__ __forceinline__ void row_store(float* p, const float (&o)[4]) { *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
__device__ __forceinline__ void row_store(float* p, const float (&o)[2]) { *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]); }

// ---- the row walk (ROWS): every 16-lane DPP row (4x4 pixel block, dqo_lane_x / _y) of the quadrant wave walks ITS OWN sub-list ----
// The forward's live byte of a (quadrant, list position) is a 4-bit row code: which rows have a pixel with work for the entry.  The
// wave collects RG live entries at a time (walk order, whatever list positions they sit at), gathers their records into LDS once, and
// builds four sub-lists from the codes; step k of the group hands row r the k-th entry of ITS list — four different entries per wave
// step, each evaluated by the 16 pixels that may have work for it (a wave stepping through the union evaluates every entry on all 64:
// 37 % useful lanes on cfg 3; the rows' own lists: 1.008 M -> 0.78 M wave steps at RG = 56, tests/diag_row_model.py).  Seven steps at
// a time, each row's 63 sums go through a ROW-LOCAL reduce-scatter (row_reduce64) and land in part[row][entry][9]; a (row, entry)
// pair is produced exactly once, so these are plain stores; at the end of the group lane j adds entry j's four row partials in the
// fixed order 0..3 and writes the 64-byte (quadrant, entry) record.  No atomics; the sum order depends on nothing but the pixels and
// the entry, so the records are bitwise reproducible AND bitwise equal between a shard and the unsharded job (a shard's lists hold
// fewer entries, which moves group and batch boundaries, but neither a row's partial nor the order 0..3 sees that).
#ifndef DQO_BWD_RG
#define DQO_BWD_RG 42  // live entries per group: a multiple of 7, at most 56 (a row's list has 8 batches of 7 steps); 42: 9 KB of LDS per
                       // wave = four waves per SIMD (56: 11.7 KB, 3.5 waves, and 159 instead of 145 us on cfg 3 although it saves 3.6 % more steps)
#endif
constexpr int RG = DQO_BWD_RG;
#ifndef DQO_BWD_ROW_NB
#define DQO_BWD_ROW_NB 7  // steps per batch: 7 (63 of 64 values through row_reduce64) or 3 (27 of 32 through row_reduce32: fewer registers)
#endif
constexpr int RNB = DQO_BWD_ROW_NB;
constexpr int RNV = RNB == 7 ? 64 : 32;     // values per lane and batch
constexpr int RLS = RNB == 7 ? 8 : 4;       // list slots per batch (the last one is padding)
static_assert(RNB == 7 || RNB == 3, "batches of 7 or 3 steps");
static_assert(RG % RNB == 0 && RG >= RNB && (RG / RNB) * RLS <= 64 && (RG / RNB) * RLS - 2 < 63, "RG: whole batches, 64 list slots per row, slot 63 free");
constexpr int R_ENT_W = 12;                          // words of an entry record: conic + opacity | x, y, object id, position | r, g, b, slot
constexpr int R_ENT = 0;                             // [RG + 1] records (the last one: the dummy a finished row keeps stepping on)
constexpr int R_LIST = R_ENT + (RG + 1) * R_ENT_W;   // [4 rows][64] u16: byte offset of the entry record, 8 slots per batch (7 used)
constexpr int R_NB = RG / RNB;                       // batches of a group at most
constexpr int R_PART = R_LIST + 4 * 64 / 2;          // [4 rows][R_NB batches][RNV] floats: a row's reduced sums of a batch exactly as the
constexpr int R_PART_W = 4 * R_NB * RNV + 12;        //   reduce-scatter leaves them (value 9 b + f at [9 b + f]), + 12 zeros (the prologue's
                                                     //   depth-hit sums use this space first)
constexpr int R_MPOS = R_PART + R_PART_W;            // [64] ints + [64] bytes: list positions and row codes of the NEXT group (collected while
constexpr int R_MCODE = R_MPOS + 64;                 //        the current one is walked: its ids are in flight during the walk)
constexpr int ROWS_BLK = ((R_MCODE + 16 + 3) / 4) * 4;   // words per wave
static_assert(R_LIST % 4 == 0 && R_PART % 4 == 0, "16-byte aligned tables");
static_assert(5 * 64 <= R_PART_W, "the depth-hit staging fits the partial table");

struct BwdTap {  // DqoLossTap, backward half: the two gradient scales of the frame
    float gc, gdw;
};
// ... from the frame totals in the spread lines (one whole wave; report: lane 0 also writes the loss and the scales out)
__device__ __forceinline__ BwdTap tap_frame_scales(const DqoGeomLayout& g, const DqoTapDev& tap, int lane, bool report) {
    BwdTap t;
    if (!report) {  // every wave with work: the two pixel counts alone (32-bit sums, no LDS crossbar)
        uint32_t c1, c3;
        dqo_tap_counts(g.spread, lane, c1, c3);
        const float n_col = fmaxf((float)c1, 1.f), n_dep = fmaxf((float)c3, 1.f);
        t.gc = tap.color_weight / (3.f * n_col), t.gdw = tap.depth_weight / n_dep;  // = loss_grad_kernel's gc / gdw
        return t;
    }
    double tot[4];
    dqo_tap_totals(g.spread, lane, tot);
    const float n_col = fmaxf((float)tot[1], 1.f), n_dep = fmaxf((float)tot[3], 1.f);
    t.gc = tap.color_weight / (3.f * n_col), t.gdw = tap.depth_weight / n_dep;  // = loss_grad_kernel's gc / gdw
    if (report && lane == 0) {
        const float color_loss = (float)(tot[0] / (3.0 * (double)n_col)), depth_loss = (float)(tot[2] / (double)n_dep);
        tap.loss_out[0] = tap.depth_weight * depth_loss + tap.color_weight * color_loss;  // mapper.py:870-875
        tap.loss_out[1] = color_loss, tap.loss_out[2] = depth_loss, tap.loss_out[3] = 0.f;
        tap.loss_out[4] = (float)tot[0], tap.loss_out[5] = (float)tot[1], tap.loss_out[6] = (float)tot[2], tap.loss_out[7] = (float)tot[3];
        tap.scale[0] = t.gc, tap.scale[1] = t.gdw;
    }
    return t;
}
// live entries per reduction batch: BWD_NB x 9 values go through one butterfly — 7 x 9 = 63 of 64 values (wave_reduce64), or
// 3 x 9 = 27 of 32 (wave_reduce32: fewer registers -> more waves per SIMD)
// GATE: DqoObjectGate (an entry acts on a pixel only if the Gaussian's object id equals the pixel's owner id) — a template parameter,
// so that the ungated kernel keeps its instruction stream.
// SEGS: waves per quadrant.  1 = the wave described at the top of the file.  8 = DqoRastCtx.list_split, the counterpart of the
// forward's split (rast_forward_blend.hip): the walk goes in rounds of SEGS chunks, chunk r * SEGS + w of round r by wave w.  Walking
// an entry maps the pixel's state (T, S) to (T / (1 - alpha), S + alpha (c - S)) — T is scaled, S goes through an affine map — so
//   pass 1  every wave composes its chunk's maps per pixel: Q = prod 1 / (1 - alpha), and (A, B) with S_out = A + B S_in;
//   scan    the state its chunk starts from = the round's start state sent through the chunks before it (one block barrier per round);
//   pass 2  the walk itself on the chunk from that state: the gradient sums and records exactly as the single wave forms them.
// The backward has no early exit, so unlike the forward nothing is evaluated that the single wave would have skipped; pass 1 costs
// a quarter of pass 2.  T and S are grouped by chunk instead of strictly back to front: last-bit differences, like the forward's.
template <int BWD_NB, bool GATE, int SEGS, bool ROWS = false>
__device__ __forceinline__ void blend_quadrant_bwd(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                                                   const float* __restrict__ dL_dpixels, const float* __restrict__ dL_ddepths,
                                                   float* __restrict__ recs, uint8_t* __restrict__ valid, const int64_t capacity,
                                                   const DqoTapDev& tap, const DqoGateDev& gate, const int tile, const int quad, const int wave, const int lane, uint32_t* const lds, const int skip_over,
                                                   const uint2 range) {
    static_assert(!ROWS || (SEGS == 1 && BWD_NB == 7), "the row walk is the single-wave walk");
    uint32_t* const blk = lds + wave * (ROWS ? ROWS_BLK : BWD_BLK);
    float4* const s_co = reinterpret_cast<float4*>(blk);
    float4* const s_xy = s_co + BWD_THREADS;
    float4* const s_rgb = s_xy + BWD_THREADS;
    int* const s_id = reinterpret_cast<int*>(s_rgb + BWD_THREADS);
    uint32_t* const s_slot = reinterpret_cast<uint32_t*>(s_id + BWD_THREADS);
    int* const s_pos = reinterpret_cast<int*>(s_slot + BWD_THREADS);
    float* const s_hit = ROWS ? reinterpret_cast<float*>(blk + R_PART) : reinterpret_cast<float*>(s_pos + BWD_THREADS);
    const int n = (int)(range.y - range.x);  // (range: the tile's list, img.ranges[tile] — the caller has it with the tile id)
    if (n == 0 || n > skip_over) return;  // (n > skip_over: SEGS == 1 beside the split blocks, which own the long lists)
    const int L = min((int)img.walk4[tile * 4 + quad], n);  // list positions [0, L) matter to this quadrant
    if (L == 0) return;
    const int tile_x = tile % v.gx, tile_y = tile / v.gx;
    const size_t HW = (size_t)v.W * v.H;
    const uint32_t px = tile_x * DQO_TILE + (quad & 1) * 8 + dqo_lane_x(lane);
    const uint32_t py = tile_y * DQO_TILE + (quad >> 1) * 8 + dqo_lane_y(lane);
    const bool inside = px < (uint32_t)v.W && py < (uint32_t)v.H;
    const size_t pid = (size_t)v.W * py + px;
    const float pixfx = (float)px, pixfy = (float)py;
    const uint8_t* const live = bin.live_q + (size_t)quad * (size_t)bin.list_cap + range.x;  // this quadrant's live bytes of this tile's segment
    // ---- ONE round of loads for everything that depends on the pixel alone (a wave of this kernel lives for ~30 us of which it issues
    // arithmetic for ~8: its time is a chain of dependent memory rounds, so independent loads are issued together — unconditionally, the
    // lanes outside the image read pixel 0 and discard it — instead of one round per `if`) ----
    const size_t ps = inside ? pid : (size_t)0;
    const float T_final_ld = img.final_T[ps];
    const uint32_t n_contrib_ld = img.n_contrib[ps], hit_word_ld = img.hit_pos[ps];
    int owner_ld = 0;
    if (GATE) owner_ld = gate.pobj[ps];
    float in0, in1, in2, in3, in4 = 0.f, in5 = 0.f, in6 = 0.f, in7 = 0.f;
    uint8_t mask_ld = 1;
    if (tap.scale == nullptr) {
        in0 = dL_dpixels[ps], in1 = dL_dpixels[HW + ps], in2 = dL_dpixels[2 * HW + ps], in3 = dL_ddepths[ps];
    } else {
        if (tap.mask) mask_ld = tap.mask[ps];
        in0 = tap.out_color[ps], in1 = tap.out_color[HW + ps], in2 = tap.out_color[2 * HW + ps], in3 = tap.out_depth[ps];
        in4 = tap.gt_color[ps], in5 = tap.gt_color[HW + ps], in6 = tap.gt_color[2 * HW + ps], in7 = tap.gt_depth[ps];
    }
    uint8_t lv_first = 0;  // the row walk's first chunk of live bytes
    if (ROWS) lv_first = (L - 1 - lane >= 0) ? live[L - 1 - lane] : (uint8_t)0;
    // DqoLossTap, backward half: every wave that has work derives the two gradient scales from the frame totals the forward left in
    // the spread lines (one load per lane + a wave sum)
    float tap_gc = 0.f, tap_gdw = 0.f;
    if (tap.scale != nullptr && !(GATE && tap.per_object)) {
        const BwdTap t = tap_frame_scales(g, tap, lane, false);
        tap_gc = t.gc, tap_gdw = t.gdw;
    }
    const float T_final = inside ? T_final_ld : 0.f;
    float T = T_final;
    const int last_contrib = inside ? (int)n_contrib_ld : 0;
    const uint32_t hit_word = inside ? hit_word_ld : 0u;
    const int hit_pos = (int)(hit_word & 0x7fffffffu);
    const bool hit_plane = (hit_word >> 31) != 0u;  // the forward decided backward.cu:1016's branch for this pixel
    const int hit_c0 = hit_pos - 1;                 // list position of the entry that fixed this pixel's depth (-1: none)
    const bool has_hit = hit_pos > 0;               // implies inside
    // ---- second round, issued before anything waits: the depth-hit entry's id and slot (a safe position for the pixels without one),
    // and — row walk — the first group's ids and slots ----
    const int gid_h = (int)bin.point_list[range.x + max(hit_c0, 0)];
    const uint32_t slot_h_ld = bin.slot_list[range.x + max(hit_c0, 0)];
    // the row walk's state (declared here: its first group is collected and its loads are started in the shadow of the work below)
    char* const ent_base = reinterpret_cast<char*>(blk + R_ENT);
    int* const s_mpos = reinterpret_cast<int*>(blk + R_MPOS);
    uint8_t* const s_mcode = reinterpret_cast<uint8_t*>(blk + R_MCODE);
    const int chunks = (L + BWD_THREADS - 1) / BWD_THREADS;
    int chunk_i = 0;                  // next chunk of 64 list positions (walk order: from L - 1 down)
    unsigned long long pend = 0ull;   // lanes of the chunk in flight whose live entry has not been taken into a group yet
    uint32_t my_code = 0u;
    uint8_t lv_nx = lv_first;
    // set bits of a wave-uniform mask below this lane (v_mbcnt: no lane-mask constant in registers)
    auto below = [](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };
    int cnt = 0;                      // entries of the group in flight
    // collect: the next RG live entries in walk order (their list positions and row codes into LDS, lane order == walk order)
    auto collect = [&]() -> int {
        int cnt = 0;
        while (cnt < RG) {
            if (pend == 0ull) {
                if (chunk_i >= chunks) break;
                my_code = (uint32_t)lv_nx & 0xfu;
                chunk_i++;
                const int pn = L - 1 - (chunk_i * BWD_THREADS + lane);
                lv_nx = pn >= 0 ? live[pn] : (uint8_t)0;  // the next chunk's live bytes, in flight while this group is walked
                pend = __builtin_amdgcn_ballot_w64(my_code != 0u);
                continue;
            }
            const int rank = below(pend);
            const bool take = ((pend >> lane) & 1ull) != 0ull && rank < RG - cnt;
            // (the chunk in flight is chunk_i - 1: its lane l sits at list position L - 1 - ((chunk_i - 1) * 64 + l))
            if (take) s_mpos[cnt + rank] = L - 1 - ((chunk_i - 1) * BWD_THREADS + lane), s_mcode[cnt + rank] = (uint8_t)my_code;
            const unsigned long long tm = __builtin_amdgcn_ballot_w64(take);
            cnt += (int)__popcll(tm);
            pend &= ~tm;
        }
        return cnt;
    };
    // gather, first hop: lane j starts the load of entry j's Gaussian id (every lane takes part, with a safe position: no value of the
    // previous group stays alive in the lanes beyond the count)
    int g_id = 0;
    auto gather_ids = [&](const int c) { g_id = (int)bin.point_list[range.x + (lane < c ? s_mpos[lane] : 0)]; };
    if constexpr (ROWS) {
        cnt = collect();
        gather_ids(cnt);
    }
    int owner = (int)0x80000000;  // object gate: this pixel's owner ("none" equals no Gaussian's non-negative object id)
    if (GATE) {
        if (inside) owner = owner_ld;
        if (owner < 0) owner = (int)0x80000000;
        if (tap.scale != nullptr && tap.per_object) {
            // per-object gradient scales: one trip per distinct owner among the quadrant's pixels (usually one or two, wave-uniform);
            // lanes 0..15 each read one spread copy of the owner's two pixel counts
            unsigned long long todo = __builtin_amdgcn_ballot_w64(owner >= 0);
            while (todo != 0ull) {
                const int f = (int)__builtin_ctzll(todo);
                const int o = __builtin_amdgcn_readlane(owner, f);
                const unsigned long long* l = g.obj_tap + ((size_t)(lane & (DQO_OBJ_SPREAD - 1)) * DQO_GATE_OBJECTS + (size_t)(o & (DQO_GATE_OBJECTS - 1))) * 4;
                // (pixel counts: at most W x H < 2^32 — summed as 32-bit words over the first DPP row, four DPP adds each instead of the
                // ten 64-bit trips through the LDS crossbar that __shfl_xor / __shfl on a long long cost: 20 dependent ds_bpermute)
                static_assert(DQO_OBJ_SPREAD <= 16, "the copies sit in one DPP row");
                const uint32_t n1 = row0_sum_u32(lane < DQO_OBJ_SPREAD ? (uint32_t)l[1] : 0u);
                const uint32_t n3 = row0_sum_u32(lane < DQO_OBJ_SPREAD ? (uint32_t)l[3] : 0u);
                const float n_col = fmaxf((float)n1, 1.f), n_dep = fmaxf((float)n3, 1.f);
                const bool mine = owner == o;
                tap_gc = mine ? tap.color_weight / (3.f * n_col) : tap_gc;
                tap_gdw = mine ? tap.depth_weight / n_dep : tap_gdw;
                todo &= ~__builtin_amdgcn_ballot_w64(mine);
            }
        }
    }
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, ddep = 0.f;
    if (tap.scale == nullptr) {
        if (inside) dp0 = in0, dp1 = in1, dp2 = in2, ddep = in3;
    } else if (inside) {
        // DqoLossTap, backward half: the gradient images of the masked L1 loss, formed in place (what loss_grad_kernel writes:
        // sign(error) x weight / count, 0 outside the mask; a tile with a list always has hit id -1 <=> hit_pos 0)
        const float gc = tap_gc, gdw = tap_gdw;
        const bool m = mask_ld != 0 && (!(GATE && tap.per_object) || owner >= 0);
        const float d0 = in0 - in4, d1 = in1 - in5, d2 = in2 - in6;
        const float gd = in7, err = in3 - gd;
        dp0 = m ? (d0 > 0.f ? gc : (d0 < 0.f ? -gc : 0.f)) : 0.f;
        dp1 = m ? (d1 > 0.f ? gc : (d1 < 0.f ? -gc : 0.f)) : 0.f;
        dp2 = m ? (d2 > 0.f ? gc : (d2 < 0.f ? -gc : 0.f)) : 0.f;
        const bool dvalid = m && hit_pos > 0 && gd > 0.f && err < tap.add_depth_thres;
        ddep = dvalid ? (err > 0.f ? gdw : (err < 0.f ? -gdw : 0.f)) : 0.f;
    }
    // No incoming gradient on any pixel of the quadrant (outside the loss mask): every term of every record would be an exact
    // zero, so nothing is written and the records stay invalid (= zero for the per-Gaussian sum).
    // An earlier backward over the same forward context (retain_graph=True, one autograd.grad call per loss term) may have
    // written this quadrant's records and marked them valid: the marks are taken back, so that record_sum_kernel never adds a
    // previous call's records (the workspace is the caller's and changes between calls) — the backward is a function of its
    // arguments only, like the reference's.
    if (__builtin_amdgcn_ballot_w64(dp0 != 0.f || dp1 != 0.f || dp2 != 0.f || ddep != 0.f) == 0ull) {
        if (SEGS > 1 && wave != 0) return;  // (every wave of the quadrant sees the same pixels: a block-uniform branch)
        for (int p0 = 0; p0 < L; p0 += BWD_THREADS) {
            const int pos = p0 + lane;
            if (pos < L && live[pos] != 0) {
                const uint32_t slot = bin.slot_list[range.x + pos];
                if ((int64_t)slot < capacity) valid[(size_t)slot * 4 + quad] = (uint8_t)0;
            }
        }
        return;
    }
    const float bgdot = v.bg[0] * dp0 + v.bg[1] * dp1 + v.bg[2] * dp2;
    const float bg_term = -T_final * bgdot;  // d(background term)/d(alpha) = bg_term / (1 - alpha): end_T, not the running T (quirk B2)
    const float3 ray = pixel_ray_b(px, py, v.focal_x, v.focal_y, v.cx, v.cy);
    // ---- third round: the depth-hit entry's surfel normal and — row walk — the first group's records ----
    const float4 n_np_h = g.normal_c[gid_h];
    const float4 pc_h = g.point_c[gid_h];
    // gather, second hop: entry j's records by its id, its instance slot, and its position / row code back from the collect's tables
    float4 g_co, g_xy, g_cs;
    int g_pos = 0;
    uint32_t g_code = 0u, g_slot = 0xffffffffu;
    auto gather_records = [&](const int c) {
        const bool on = lane < c;
        g_pos = on ? s_mpos[lane] : 0;
        g_code = on ? (uint32_t)s_mcode[lane] : 0u;
        g_slot = on ? bin.slot_list[range.x + g_pos] : 0xffffffffu;
        g_co = g.conic_opacity[g_id], g_xy = g.xy_depth[g_id], g_cs = g.rgb_smax[g_id];
    };
    if constexpr (ROWS) gather_records(cnt);
    else g_co = g_xy = g_cs = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- depth-hit sums (backward.cu:997-1065): ONCE per pixel, outside the entry loop ----
    // Every pixel has at most one entry that fixed its depth (hit_pos) and the gradient it sends to that Gaussian depends on
    // nothing the walk below computes.  Only the pixel-dependent factors are summed here (DqoGradRec::hit).  Pixels of the
    // quadrant that share the hit entry are added up through wave-private LDS: the lowest such lane (the leader) stores its own
    // terms, the others ds_add_f32 theirs onto them — one instruction's lanes are served in a fixed order, and the table is
    // private to the wave, so the sums are reproducible — and the leader writes floats 9..13 of the (quadrant, instance) record.
    // The walk marks those records with validity 3 when it writes their colour part.
    {
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(has_hit);
        if (hm != 0ull && (SEGS == 1 || wave == 0)) {
            float h[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            const uint32_t slot_h = has_hit ? slot_h_ld : 0xffffffffu;
            if (has_hit) {
#pragma clang fp contract(off)
                // The reference's statements per pixel (backward.cu:1018-1041), IEEE division included: d(depth)/d(n_c) =
                // ray.z (nr p_c - np ray) / nr^2 cancels INSIDE the pixel (nr p_c - np ray = nr (p_c - hit point): the surfel's
                // extent against its distance from the camera).  Rounds 2-5 summed ray.z / nr and ray.z ray / nr^2 over the pixels
                // and cancelled the two SUMS per Gaussian, with v_rcp_f32: rotation rows 5x further from fp64 than the reference's
                // own arithmetic is.  What is left to the per-Gaussian chain is linear: V^T and d(normal)/d(quaternion).
                const float4 n_np = n_np_h, pc = pc_h;
                const float nr_f = n_np.x * ray.x + n_np.y * ray.y + n_np.z * ray.z;
                const float nr = (float)((double)nr_f + 1e-8);  // backward.cu:1018
                const float inv_nr = 1.0f / nr, inv_nr2 = inv_nr * inv_nr;
                const float np = n_np.x * pc.x + n_np.y * pc.y + n_np.z * pc.z;
                h[0] = hit_plane ? 0.f : ddep;  // the forward decided backward.cu:1016's branch for this pixel
                h[1] = hit_plane ? ddep * (ray.z * inv_nr) : 0.f;
                h[2] = hit_plane ? ddep * (ray.z * (nr * pc.x - np * ray.x) * inv_nr2) : 0.f;
                h[3] = hit_plane ? ddep * (ray.z * (nr * pc.y - np * ray.y) * inv_nr2) : 0.f;
                h[4] = hit_plane ? ddep * (ray.z * (nr * pc.z - np * ray.z) * inv_nr2) : 0.f;
            }
            int leader = lane;  // lowest lane with the same hit entry
            unsigned long long todo = hm;
            while (todo != 0ull) {  // one trip per distinct hit entry of the quadrant (wave-uniform)
                const int f = (int)__builtin_ctzll(todo);
                const int key = __builtin_amdgcn_readlane(hit_pos, f);
                const bool same = has_hit && hit_pos == key;
                leader = same ? f : leader;
                todo &= ~__builtin_amdgcn_ballot_w64(same);
            }
            const bool is_leader = has_hit && leader == lane;
            if (is_leader) {
#pragma unroll
                for (int i = 0; i < 5; i++) s_hit[lane * 5 + i] = h[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (has_hit && !is_leader) {
#pragma unroll
                for (int i = 0; i < 5; i++) atomicAdd(&s_hit[leader * 5 + i], h[i]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (is_leader && (int64_t)slot_h < capacity) {
                float* rec = recs + ((size_t)slot_h * 4 + quad) * 16 + 9;
#pragma unroll
                for (int i = 0; i < 5; i++) rec[i] = s_hit[lane * 5 + i];
            }
        }
    }
    if constexpr (ROWS) {
        // (the depth-hit staging shares LDS with the partial table: its last reads happened-before in program order)
        float S0 = 0.f, S1 = 0.f, S2 = 0.f;  // colour blended behind the current entry (see below)
        uint16_t* const s_list = reinterpret_cast<uint16_t*>(blk + R_LIST);
        float* const s_part = reinterpret_cast<float*>(blk + R_PART);
        const int row = lane >> 4, sl = lane & 15;
        const char* const list_row = reinterpret_cast<const char*>(s_list) + row * 128;  // this row's sub-list (64 u16)
        // after row_reduce64 lane sl of a row holds the totals of values 4 sl .. 4 sl + 3 (value 9 b + f = float f of the batch's b-th
        // entry): one 16-byte store per lane and batch puts them at part[row][batch][4 sl ..]
        float* const part_lane = s_part + row * (R_NB * RNV) + sl * (RNV / 16);
        float* const s_zero = s_part + 4 * R_NB * RNV;  // nine zeros: what a row that has nothing for an entry contributes
        constexpr uint32_t DUMMY = (uint32_t)(RG * R_ENT_W * 4);  // byte offset of the dummy record
        {   // the dummy record: opacity 0 -> alpha 0 -> every term an exact zero and the pixel state untouched; position -2 matches no hit
            if (lane < R_ENT_W) reinterpret_cast<uint32_t*>(ent_base + DUMMY)[lane] = lane == 7 ? 0xfffffffeu : 0u;
            if (lane < 12) s_zero[lane] = 0.f;
        }
        for (;;) {
            if (cnt == 0) break;
            // ---- commit: entry j's records into LDS (the loads were started a round ago) ----
            const uint32_t code = g_code;
            if (lane < cnt) {
                const uint32_t slot_j = g_slot;
                float4* e = reinterpret_cast<float4*>(ent_base + lane * (R_ENT_W * 4));
                e[0] = make_float4(-0.5f * g_co.x, g_co.y, -0.5f * g_co.z, g_co.w);  // (dqo_power_pre)
                e[1] = make_float4(g_xy.x, g_xy.y, g_xy.w, __int_as_float(g_pos));
                e[2] = make_float4(g_cs.x, g_cs.y, g_cs.z, __uint_as_float(slot_j));
            }
            // does entry j also carry depth-hit sums?  The group's positions descend from lane 0 to lane cnt - 1: a pixel whose hit
            // position lies in that range names one of the group's entries (a hit entry is live for its pixel's row); one trip per
            // distinct such position, and every pixel's turn comes exactly once in the whole walk
            bool hit_j = false;
            {
                const int p_hi = __builtin_amdgcn_readlane(g_pos, 0), p_lo = __builtin_amdgcn_readlane(g_pos, cnt - 1);
                const bool in_grp = has_hit && hit_c0 <= p_hi && hit_c0 >= p_lo;
                unsigned long long todo = __builtin_amdgcn_ballot_w64(in_grp);
                while (todo != 0ull) {
                    const int key = __builtin_amdgcn_readlane(hit_c0, (int)__builtin_ctzll(todo));
                    hit_j = hit_j || (lane < cnt && g_pos == key);
                    todo &= ~__builtin_amdgcn_ballot_w64(in_grp && hit_c0 == key);
                }
            }
            // ---- the four sub-lists: rank r_k of entry j in row r's list -> slot (r_k / 7) * 8 + r_k % 7 (8 u16 = one 16-byte read per batch) ----
            reinterpret_cast<uint2*>(s_list)[lane] = make_uint2(DUMMY | (DUMMY << 16), DUMMY | (DUMMY << 16));  // 4 x 64 u16 = 512 B
            int nsteps = 0;
            uint32_t where = 0x3f3f3f3fu;  // byte r: entry j's slot in row r's list (batch * 8 + step), 63 = not in that list
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const bool in = ((code >> r) & 1u) != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                const int rk = below(m);
                const int sidx = RNB == 7 ? rk + ((rk * 37) >> 8) : rk + ((rk * 43) >> 7);  // rk + rk / RNB for rk < 64: batch * RLS + step
                if (in) s_list[r * 64 + sidx] = (uint16_t)(lane * (R_ENT_W * 4));
                where = in ? ((where & ~(0xffu << (8 * r))) | ((uint32_t)sidx << (8 * r))) : where;
                nsteps = max(nsteps, (int)__popcll(m));
            }
            // ---- the NEXT group's entries are collected now and their ids fetched while this group is walked (one memory round of the
            // two a group's gather takes leaves the critical path) ----
            const int cnt_next = collect();
            gather_ids(cnt_next);
            // ---- walk: seven steps per batch ----
            auto batch = [&](const int t, auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                uint32_t lw[4];
                if constexpr (RNB == 7) {
                    const uint4 li = *reinterpret_cast<const uint4*>(list_row + t * 16);
                    lw[0] = li.x, lw[1] = li.y, lw[2] = li.z, lw[3] = li.w;
                } else {
                    const uint2 li = *reinterpret_cast<const uint2*>(list_row + t * 8);
                    lw[0] = li.x, lw[1] = li.y, lw[2] = lw[3] = 0u;
                }
                float v64[RNV];
#pragma unroll
                for (int i = 9 * RNB; i < RNV; i++) v64[i] = 0.f;
#pragma unroll
                for (int b = 0; b < RNB; b++) {
                    float r_c0 = 0.f, r_c1 = 0.f, r_c2 = 0.f, r_mx = 0.f, r_my = 0.f, r_ka = 0.f, r_kb = 0.f, r_kc = 0.f, r_op = 0.f;
                    if (FULL || t * RNB + b < nsteps) {  // wave-uniform
                        const uint32_t off = (b & 1) ? (lw[b >> 1] >> 16) : (lw[b >> 1] & 0xffffu);
                        const float4* e = reinterpret_cast<const float4*>(ent_base + off);
                        const float4 co = e[0], xy = e[1], cs = e[2];
                        const int c0 = __float_as_int(xy.w);  // 0-based list position == the reference's `contributor` after its --
                        // ---- predicated per-pixel gradient terms (backward.cu:932-994): the statements of the union walk below ----
                        const float dx = xy.x - pixfx, dy = xy.y - pixfy;
                        const float power = dqo_power_pre(co.x, co.y, co.z, dx, dy);
                        const float Gx = dqo_gauss(power);
                        const float alpha_x = fminf(0.99f, co.w * Gx);
                        const bool did_color = (c0 < last_contrib) & (power <= 0.0f) & (alpha_x >= 1.0f / 255.0f) & (!GATE || __float_as_int(xy.z) == owner);
                        const float alpha = did_color ? alpha_x : 0.f;
                        const float G = did_color ? Gx : 0.f;
                        const float inv_1ma = dqo_rcp(1.f - alpha);
                        T = T * inv_1ma;
                        const float e0 = cs.x - S0, e1 = cs.y - S1, e2 = cs.z - S2;
                        float dL_dalpha = (e0 * dp0 + e1 * dp1 + e2 * dp2) * T;
                        dL_dalpha += bg_term * inv_1ma;
                        const float dchannel_dcolor = alpha * T;
                        S0 += alpha * e0;
                        S1 += alpha * e1;
                        S2 += alpha * e2;
                        const float q = G * dL_dalpha;
                        const float qx = q * dx, qy = q * dy;
                        r_c0 = dchannel_dcolor * dp0;
                        r_c1 = dchannel_dcolor * dp1;
                        r_c2 = dchannel_dcolor * dp2;
                        r_mx = qx, r_my = qy;
                        r_ka = qx * dx;
                        r_kb = qx * dy;
                        r_kc = qy * dy;
                        r_op = q;
                    }
                    v64[9 * b + 0] = r_c0, v64[9 * b + 1] = r_c1, v64[9 * b + 2] = r_c2, v64[9 * b + 3] = r_mx, v64[9 * b + 4] = r_my;
                    v64[9 * b + 5] = r_ka, v64[9 * b + 6] = r_kb, v64[9 * b + 7] = r_kc, v64[9 * b + 8] = r_op;
                }
                // the row's sums of the batch's entries, as they are: part[row][batch][9 b + f] (every (row, entry) pair exists once)
                float out[RNV / 16];
                row_reduce(v64, out, lane);
                row_store(part_lane + t * RNV, out);
            };
            {
                int t = 0;
                for (; (t + 1) * RNB <= nsteps; t++) batch(t, std::true_type{});
                if (t * RNB < nsteps) batch(t, std::false_type{});
            }
            // ---- write-out: lane j adds entry j's row partials in the fixed order 0..3 and stores the record's nine floats ----
            const uint32_t slot_j = lane < cnt ? reinterpret_cast<const uint32_t*>(ent_base)[lane * R_ENT_W + 11] : 0xffffffffu;
            if (lane < cnt && (int64_t)slot_j < capacity) {
                const float* pr[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t sidx = (where >> (8 * r)) & 0xffu;
                    pr[r] = sidx == 63u ? s_zero : s_part + (r * R_NB + (int)(sidx / RLS)) * RNV + (int)(sidx % RLS) * 9;
                }
                float acc[9];
#pragma unroll
                for (int f = 0; f < 9; f++) acc[f] = ((pr[0][f] + pr[1][f]) + pr[2][f]) + pr[3][f];
                float* rec = recs + ((size_t)slot_j * 4 + quad) * 16;
                reinterpret_cast<float4*>(rec)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                reinterpret_cast<float4*>(rec)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
                rec[8] = acc[8];
                valid[(size_t)slot_j * 4 + quad] = hit_j ? (uint8_t)3 : (uint8_t)1;  // 3: the depth-hit floats 9..13 are there as well
            }
            // ---- the next group ----
            cnt = cnt_next;
            if (cnt == 0) break;
            gather_records(cnt);
        }
        return;
    }
    // colour blended behind the current entry (the reference's accum_rec after folding in last_alpha / last_color,
    // backward.cu:957-962, evaluated one step earlier: same operands, same rounding)
    float S0 = 0.f, S1 = 0.f, S2 = 0.f;
    const int lane_b = lane / 9, lane_f = lane - 9 * lane_b;  // this lane's (entry of the batch, record float) after the butterfly
    cnt = 0;  // live entries of the chunk in flight (compacted in LDS)
    // The live entries of the chunk are processed BWD_NB at a time: their 9 colour-path sums each go through ONE reduce-scatter
    // butterfly, after which lane l = 9 b + f holds float f of entry b's record.  A batch whose BWD_NB entries all exist runs as
    // one straight-line block (FULL): no wave-uniform branch sits between two entries, so the scheduler interleaves the parts of
    // neighbouring entries that do not depend on each other (footprint, exp, alpha — everything but the T / S recurrences).  A
    // wave issues a dependent VALU instruction only every ~10 cycles (tools/ubench_valu.hip), so the loop is paced by the length
    // of its dependency chains, not by the number of instructions: with a branch per entry the chain was one whole entry long.
    auto batch = [&](const int k0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int NV = BWD_NB == 7 ? 64 : 32;
        float v64[NV];
#pragma unroll
        for (int i = 9 * BWD_NB; i < NV; i++) v64[i] = 0.f;
        uint32_t hitmask = 0u;  // entries of this batch that also carry depth-hit sums (wave-uniform)
#pragma unroll
        for (int b = 0; b < BWD_NB; b++) {
            const int k = k0 + b;
            float r_c0 = 0.f, r_c1 = 0.f, r_c2 = 0.f, r_mx = 0.f, r_my = 0.f, r_ka = 0.f, r_kb = 0.f, r_kc = 0.f, r_op = 0.f;
            if (FULL || k < cnt) {  // wave-uniform
                const float4 co = s_co[k], xy = s_xy[k], cs = s_rgb[k];
                const int c0 = s_pos[k];  // 0-based list position == the reference's `contributor` after its --
                // ---- predicated per-pixel gradient terms (backward.cu:932-994) ----
                // A pixel that did not blend this entry runs the same arithmetic with alpha = 0 and G = 0: T / (1 - 0) = T
                // and S + 0 * (c - S) = S leave its state untouched bit for bit, and all its gradient terms are exact zeros.
                const float dx = xy.x - pixfx, dy = xy.y - pixfy;
                const float power = dqo_power_pre(co.x, co.y, co.z, dx, dy);
                const float Gx = dqo_gauss(power);
                const float alpha_x = fminf(0.99f, co.w * Gx);
                const bool did_color = c0 < last_contrib && power <= 0.0f && alpha_x >= 1.0f / 255.0f && (!GATE || __float_as_int(xy.w) == owner);
                const float alpha = did_color ? alpha_x : 0.f;
                const float G = did_color ? Gx : 0.f;
                const float inv_1ma = dqo_rcp(1.f - alpha);
                T = T * inv_1ma;  // T / (1 - alpha), backward.cu:948
                const float e0 = cs.x - S0, e1 = cs.y - S1, e2 = cs.z - S2;
                float dL_dalpha = (e0 * dp0 + e1 * dp1 + e2 * dp2) * T;
                dL_dalpha += bg_term * inv_1ma;
                const float dchannel_dcolor = alpha * T;
                S0 += alpha * e0;  // = alpha c + (1 - alpha) S: one fma on the difference that is needed anyway
                S1 += alpha * e1;
                S2 += alpha * e2;
                // Everything downstream of dL/dalpha * G is linear in per-Gaussian constants (opacity, conic, W/2, H/2):
                // the wave only sums the pixel moments of q = G * dL/dalpha — q, q dx, q dy, q dx^2, q dx dy, q dy^2 — and
                // gaussian_backward_kernel applies those constants once per Gaussian (backward.cu:964-994 does it per pair).
                const float q = G * dL_dalpha;
                const float qx = q * dx, qy = q * dy;
                r_c0 = dchannel_dcolor * dp0;
                r_c1 = dchannel_dcolor * dp1;
                r_c2 = dchannel_dcolor * dp2;
                r_mx = qx, r_my = qy;
                r_ka = qx * dx;
                r_kb = qx * dy;
                r_kc = qy * dy;
                r_op = q;

                // the entry fixed some pixel's depth: its record also carries the depth-hit sums written before the loop
                if (__builtin_amdgcn_ballot_w64(hit_c0 == c0) != 0ull) hitmask |= 1u << b;
            }
            v64[9 * b + 0] = r_c0, v64[9 * b + 1] = r_c1, v64[9 * b + 2] = r_c2, v64[9 * b + 3] = r_mx, v64[9 * b + 4] = r_my;
            v64[9 * b + 5] = r_ka, v64[9 * b + 6] = r_kb, v64[9 * b + 7] = r_kc, v64[9 * b + 8] = r_op;
        }
        float tot;
        if constexpr (BWD_NB == 7) tot = wave_reduce64(v64, lane);
        else tot = wave_reduce32(v64, lane);
        // lane 9 b + f stores float f (0..8) of entry b's 64-byte partial record; lanes 0..BWD_NB-1 mark the records valid
        // (1 = colour-path floats, 3 = depth-hit floats 9..13 present as well)
        const int kb = k0 + lane_b;
        if (lane < 9 * BWD_NB && (FULL || kb < cnt)) {
            const uint32_t slot = s_slot[kb];
            if ((int64_t)slot < capacity) recs[((size_t)slot * 4 + quad) * 16 + lane_f] = tot;
        }
        if (lane < BWD_NB && (FULL || k0 + lane < cnt)) {
            const uint32_t slot = s_slot[k0 + lane];
            if ((int64_t)slot < capacity) valid[(size_t)slot * 4 + quad] = ((hitmask >> lane) & 1u) ? (uint8_t)3 : (uint8_t)1;
        }
    };
    // the chunk's live entries gathered into LDS (lane order == walk order); sets cnt
    auto gather = [&](int pos, bool is_live) {
        const unsigned long long lm = __builtin_amdgcn_ballot_w64(is_live);
        cnt = (int)__popcll(lm);
        const int myk = (int)__popcll(lm & ((1ull << lane) - 1ull));
        if (is_live) {
            // only live entries are gathered at all
            const int id = (int)bin.point_list[range.x + pos];
            s_id[myk] = id;
            s_pos[myk] = pos;
            s_slot[myk] = bin.slot_list[range.x + pos];
            const float4 co_g = g.conic_opacity[id];
            s_co[myk] = make_float4(-0.5f * co_g.x, co_g.y, -0.5f * co_g.z, co_g.w);  // (dqo_power_pre)
            s_xy[myk] = g.xy_depth[id];
            s_rgb[myk] = g.rgb_smax[id];
        }
    };
    auto walk_chunk = [&]() {
        int k0 = 0;
        for (; k0 + BWD_NB <= cnt; k0 += BWD_NB) batch(k0, std::true_type{});
        if (k0 < cnt) batch(k0, std::false_type{});
    };
    if (SEGS == 1) {
        // chunk c covers list positions L-1-c*64 ... descending; lane l looks at position L-1-(c*64+l): lane order == walk order
        lv_nx = (L - 1 - lane >= 0) ? live[L - 1 - lane] : (uint8_t)0;
        for (int c = 0; c < chunks; c++) {
            const int pos = L - 1 - (c * BWD_THREADS + lane);
            const bool is_live = lv_nx != 0;
            {
                const int pn = pos - BWD_THREADS;
                lv_nx = pn >= 0 ? live[pn] : (uint8_t)0;  // next chunk's live bytes, in flight while this chunk is processed
            }
            if (__builtin_amdgcn_ballot_w64(is_live) == 0ull) continue;
            gather(pos, is_live);
            walk_chunk();
        }
    } else {
        // ---- rounds of SEGS chunks, chunk r * SEGS + w (in walk order) to wave w ----
        float* const s_x = reinterpret_cast<float*>(lds + SEGS * BWD_BLK);  // [round parity][wave][Q, B, A0, A1, A2][lane]
        float T_round = T_final, R0 = 0.f, R1 = 0.f, R2 = 0.f;              // the state at the start of the round (same bits in every wave)
        int p_nx = L - 1 - (wave * BWD_THREADS + lane);
        lv_nx = p_nx >= 0 ? live[p_nx] : (uint8_t)0;
        for (int r = 0; r * SEGS < chunks; r++) {
            const int pos = L - 1 - ((r * SEGS + wave) * BWD_THREADS + lane);
            const bool is_live = lv_nx != 0;
            {
                const int pn = pos - SEGS * BWD_THREADS;
                lv_nx = pn >= 0 ? live[pn] : (uint8_t)0;
            }
            gather(pos, is_live);  // (no live entry, or a chunk past the front of the list: cnt = 0, the identity map)
            // pass 1: the chunk's map of the pixel's state
            float Q = 1.f, B = 1.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
#pragma unroll 2
            for (int k = 0; k < cnt; k++) {
                const float4 co = s_co[k], xy = s_xy[k], cs = s_rgb[k];
                const int c0 = s_pos[k];
                const float dx = xy.x - pixfx, dy = xy.y - pixfy;
                const float power = dqo_power_pre(co.x, co.y, co.z, dx, dy);
                const float alpha_x = fminf(0.99f, co.w * dqo_gauss(power));
                const bool did_color = c0 < last_contrib && power <= 0.0f && alpha_x >= 1.0f / 255.0f && (!GATE || __float_as_int(xy.w) == owner);
                const float alpha = did_color ? alpha_x : 0.f;
                Q *= dqo_rcp(1.f - alpha);
                A0 += alpha * (cs.x - A0), A1 += alpha * (cs.y - A1), A2 += alpha * (cs.z - A2);
                B *= 1.f - alpha;
            }
            float* const xw = s_x + ((r & 1) * SEGS + wave) * BWD_XCH + lane;
            xw[0 * BWD_THREADS] = Q, xw[1 * BWD_THREADS] = B, xw[2 * BWD_THREADS] = A0, xw[3 * BWD_THREADS] = A1, xw[4 * BWD_THREADS] = A2;
            // (one barrier per round: the round after next writes this parity again, and the next round's barrier lies between)
            __syncthreads();
            // scan: the state this chunk starts from, and the state the next round starts from
            float Tn = T_round, N0 = R0, N1 = R1, N2 = R2;
#pragma unroll
            for (int j = 0; j < SEGS; j++) {
                if (j == wave) T = Tn, S0 = N0, S1 = N1, S2 = N2;
                const float* xj = s_x + ((r & 1) * SEGS + j) * BWD_XCH + lane;
                const float Bj = xj[1 * BWD_THREADS];
                Tn *= xj[0 * BWD_THREADS];
                N0 = xj[2 * BWD_THREADS] + Bj * N0, N1 = xj[3 * BWD_THREADS] + Bj * N1, N2 = xj[4 * BWD_THREADS] + Bj * N2;
            }
            T_round = Tn, R0 = N0, R1 = N1, R2 = N2;
            // pass 2: the walk of the chunk from that state
            walk_chunk();
        }
    }
}

// block b: XCD group x = b % 8 (blocks b and b + 8 share an XCD), within the group item j = b / 8 = (tile slot, quadrant)
#ifndef BWD_WPB
#define BWD_WPB 1  // waves (= quadrants of ONE tile) per workgroup; independent of each other either way
#endif
#ifndef DQO_BWD_ROWS_WAVES
#define DQO_BWD_ROWS_WAVES 4  // the row walk's LDS (ROWS_BLK words per wave) leaves room for 4 waves per SIMD at RG = 42
#endif
template <int BWD_NB, bool GATE, bool ROWS = false>
__global__ __launch_bounds__(BWD_THREADS * BWD_WPB, ROWS ? DQO_BWD_ROWS_WAVES : (BWD_NB == 7 ? 5 : 8)) void blend_backward_kernel(const DqoView v, DqoGeomLayout g, DqoImageLayout img,
                                                                     DqoBinLayout bin, const float* __restrict__ dL_dpixels,
                                                                     const float* __restrict__ dL_ddepths,
                                                                     float* __restrict__ recs, uint8_t* __restrict__ valid,
                                                                     int64_t capacity, const DqoTapDev tap, const DqoGateDev gate) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[(ROWS ? ROWS_BLK : BWD_BLK) * BWD_WPB];
    const int wave = BWD_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int lane = (int)(threadIdx.x & 63);
    // DqoLossTap: the first block also reports the loss (before the early exits below: block 0 may have no list)
    if (tap.scale != nullptr && blockIdx.x == 0 && wave == 0) {
        if (GATE && tap.per_object) tap_report_per_object(g, tap);
        else tap_frame_scales(g, tap, lane, true);
    }
    const int xg = blockIdx.x & 7, jg = BWD_WPB > 1 ? ((int)(blockIdx.x >> 3) * BWD_WPB + wave) : (int)(blockIdx.x >> 3);
    const int T8 = (v.gx * v.gy + 7) / 8;
    const uint4 si = img.slot_info[xg * T8 + (jg >> 2)];  // (tile, list start, list end) of the slot: one round
    const uint32_t tile_u = si.x;
    if (tile_u == 0xffffffffu) return;  // unused slot
    blend_quadrant_bwd<BWD_NB, GATE, 1, ROWS>(v, g, img, bin, dL_dpixels, dL_ddepths, recs, valid, capacity, tap, gate, (int)tile_u, jg & 3, wave,
                                               lane, lds, 0x7fffffff, make_uint2(si.y, si.z));
}

// DqoRastCtx.list_split, the backward's half (the layout of blend_forward_split_kernel): blocks of eight waves; the first BSPLIT_GRID and
// the last BSPLIT_GRID blocks take the long lists from img.split_tiles one (tile, quadrant) at a time by ticket, every wave one chunk per
// round; the blocks in between are eight independent single-wave walks, two tiles of one XCD band, skipping the long lists.
constexpr int BSPLIT_RUNS = 8;
constexpr int BSPLIT_GRID = 256;
// LDS of a block: the eight waves' blocks + the pass-1 results of two round parities (long lists), or — ROWS: the short-list blocks walk
// their lists like blend_backward_kernel<7, GATE, true>, the row walk — eight row-walk blocks: 72.7 KB, more than a kernel may declare
// statically, so the array is the launch's dynamic LDS (two blocks per CU either way)
constexpr int BSPLIT_LDS_SPLIT = BSPLIT_RUNS * BWD_BLK + 2 * BSPLIT_RUNS * BWD_XCH;
constexpr int BSPLIT_LDS_ROWS = BSPLIT_RUNS * ROWS_BLK;
template <bool ROWS>
constexpr int bsplit_lds_words() { return (ROWS && BSPLIT_LDS_ROWS > BSPLIT_LDS_SPLIT) ? BSPLIT_LDS_ROWS : BSPLIT_LDS_SPLIT; }
template <bool GATE, bool ROWS>
__global__ __launch_bounds__(BWD_THREADS * BSPLIT_RUNS, 4) void blend_backward_split_kernel(const DqoView v, DqoGeomLayout g, DqoImageLayout img,
                                                                     DqoBinLayout bin, const float* __restrict__ dL_dpixels,
                                                                     const float* __restrict__ dL_ddepths,
                                                                     float* __restrict__ recs, uint8_t* __restrict__ valid,
                                                                     int64_t capacity, const DqoTapDev tap, const DqoGateDev gate) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // bsplit_lds_words<ROWS>() words
    __shared__ uint32_t s_item;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    if (tap.scale != nullptr && blockIdx.x == 0 && wave == 0) {
        if (GATE && tap.per_object) tap_report_per_object(g, tap);
        else tap_frame_scales(g, tap, lane, true);
    }
    const int T = v.gx * v.gy, T8 = (T + 7) / 8;
    const int short_blocks = 8 * ((T8 + 1) / 2);
    if (blockIdx.x < BSPLIT_GRID || (int)blockIdx.x >= BSPLIT_GRID + short_blocks) {
        const uint32_t items = 4u * min(g.counters[4], (uint32_t)T);
        for (;;) {  // (ends for every wave of every block: the ticket only grows and `items` is fixed)
            if (threadIdx.x == 0) s_item = atomicAdd(&g.counters[5], 1u);
            __syncthreads();
            const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);
            if (it >= items) {
                // every long-list block draws exactly one ticket beyond the queue: the block that draws the last of them puts the ticket
                // back to zero, for a second backward over the same forward (the forward's zero fill only runs once per frame)
                if (threadIdx.x == 0 && it == items + 2u * BSPLIT_GRID - 1u) g.counters[5] = 0u;
                return;
            }
            const int tile_s = (int)img.split_tiles[it >> 2];
            blend_quadrant_bwd<7, GATE, BSPLIT_RUNS>(v, g, img, bin, dL_dpixels, dL_ddepths, recs, valid, capacity, tap, gate, tile_s,
                                                     (int)(it & 3u), wave, lane, lds, 0x7fffffff, img.ranges[tile_s]);
            __syncthreads();  // (everyone has read this trip's ticket and the last round's pass-1 results)
        }
    }
    // the lists in the queue are the ones longer than the FORWARD's threshold (0: it built no queue, every list is walked here)
    const int fwd_split = (int)g.counters[6];
    const int b = (int)blockIdx.x - BSPLIT_GRID;
    const int slot = (b >> 3) * 2 + (wave >> 2);  // two tile slots of band b % 8 per block
    if (slot >= T8) return;
    const uint32_t tile_u = img.tile_order[(b & 7) * T8 + slot];
    if (tile_u == 0xffffffffu) return;
    blend_quadrant_bwd<7, GATE, 1, ROWS>(v, g, img, bin, dL_dpixels, dL_ddepths, recs, valid, capacity, tap, gate, (int)tile_u, wave & 3, wave, lane,
                                         lds, fwd_split > 0 ? fwd_split : 0x7fffffff, img.ranges[tile_u]);
}

}  // namespace

namespace {
// DqoLossTap on an empty map (P = 0: no blend kernel runs in the backward): the loss of the background-only frame is still reported.
__global__ void tap_report_kernel(DqoGeomLayout g, const DqoTapDev tap) {
    if (tap.per_object) {
        tap_report_per_object(g, tap);
        return;
    }
    double tot[4];
    dqo_tap_totals(g.spread, (int)threadIdx.x, tot);
    if (threadIdx.x != 0) return;
    const float n_col = fmaxf((float)tot[1], 1.f), n_dep = fmaxf((float)tot[3], 1.f);
    const float color_loss = (float)(tot[0] / (3.0 * (double)n_col)), depth_loss = (float)(tot[2] / (double)n_dep);
    tap.loss_out[0] = tap.depth_weight * depth_loss + tap.color_weight * color_loss;
    tap.loss_out[1] = color_loss, tap.loss_out[2] = depth_loss, tap.loss_out[3] = 0.f;
    tap.loss_out[4] = (float)tot[0], tap.loss_out[5] = (float)tot[1], tap.loss_out[6] = (float)tot[2], tap.loss_out[7] = (float)tot[3];
    tap.scale[0] = tap.color_weight / (3.f * n_col), tap.scale[1] = tap.depth_weight / n_dep;
}

}  // namespace

int dqo_launch_tap_report(const DqoGeomLayout& g, const DqoTapDev& tap, hipStream_t s) {
    DQO_LAUNCH("tap_report_kernel", tap_report_kernel, dim3(1), dim3(64), s, g, tap);
    return DQO_OK;
}

int dqo_launch_blend_backward(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin, int T,
                              const float* dL_dcolor, const float* dL_ddepth, DqoGradRec* recs, uint8_t* valid, int64_t capacity,
                              const DqoTapDev& tap, const DqoGateDev& gate, int list_split, hipStream_t s) {
    // DQO_BWD_NB=3 (measurement only) selects the 32-value butterfly: 55 instead of 81 VGPRs, 5.4 instead of 3.7 waves resident
    // per SIMD — and 5 % SLOWER (round 2, profiles/README.md): the kernel is bound by VALU execution, not by latency
    static const int nb = [] {
        const char* e = getenv("DQO_BWD_NB");
        return e ? atoi(e) : 7;
    }();
    float* r = reinterpret_cast<float*>(recs);
    // DQO_BWD_ROWS=0 (measurement / A-B only): the union walk (every entry on all 64 pixels of the quadrant) instead of the row walk
    static const bool rows = [] {
        const char* e = getenv("DQO_BWD_ROWS");
        return e ? atoi(e) != 0 : true;
    }();
    if (list_split > 0) {
        const dim3 grid(2 * BSPLIT_GRID + 8 * (((T + 7) / 8 + 1) / 2)), block(BWD_THREADS * BSPLIT_RUNS);
        // (the row-walk variant's dynamic LDS is above the 64 KB a launch gets without asking: raised once per process and kernel)
        static const int raised = [] {
            const int bytes = bsplit_lds_words<true>() * 4;
            hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&blend_backward_split_kernel<true, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&blend_backward_split_kernel<false, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            return (e1 == hipSuccess && e2 == hipSuccess) ? 1 : 0;
        }();
        const bool use_rows = rows && nb == 7 && raised != 0;
#define DQO_BSPLIT(GT, RW)                                                                                                                 \
    DQO_LAUNCH_SMEM("blend_backward_kernel", (blend_backward_split_kernel<GT, RW>), grid, block, (size_t)bsplit_lds_words<RW>() * 4, s, v, g, img, \
                    bin, dL_dcolor, dL_ddepth, r, valid, capacity, tap, gate)
        if (gate.gobj != nullptr) {
            if (use_rows) DQO_BSPLIT(true, true);
            else DQO_BSPLIT(true, false);
        } else {
            if (use_rows) DQO_BSPLIT(false, true);
            else DQO_BSPLIT(false, false);
        }
#undef DQO_BSPLIT
        return DQO_OK;
    }
    const dim3 grid(8 * ((T + 7) / 8) * 4 / BWD_WPB);
    if (rows && nb == 7) {
        if (gate.gobj != nullptr)
            DQO_LAUNCH("blend_backward_kernel", (blend_backward_kernel<7, true, true>), grid, dim3(BWD_THREADS * BWD_WPB), s, v, g, img, bin, dL_dcolor,
                       dL_ddepth, r, valid, capacity, tap, gate);
        else
            DQO_LAUNCH("blend_backward_kernel", (blend_backward_kernel<7, false, true>), grid, dim3(BWD_THREADS * BWD_WPB), s, v, g, img, bin, dL_dcolor,
                       dL_ddepth, r, valid, capacity, tap, gate);
        return DQO_OK;
    }
    if (gate.gobj != nullptr)
        DQO_LAUNCH("blend_backward_kernel", (blend_backward_kernel<7, true>), grid, dim3(BWD_THREADS * BWD_WPB), s, v, g, img, bin, dL_dcolor, dL_ddepth, r,
                   valid, capacity, tap, gate);
    else if (nb == 7)
        DQO_LAUNCH("blend_backward_kernel", (blend_backward_kernel<7, false>), grid, dim3(BWD_THREADS * BWD_WPB), s, v, g, img, bin, dL_dcolor, dL_ddepth, r,
                   valid, capacity, tap, gate);
    else
        DQO_LAUNCH("blend_backward_kernel", (blend_backward_kernel<3, false>), grid, dim3(BWD_THREADS * BWD_WPB), s, v, g, img, bin, dL_dcolor, dL_ddepth, r,
                   valid, capacity, tap, gate);
    return DQO_OK;
}
