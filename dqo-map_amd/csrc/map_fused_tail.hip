// Fused per-Gaussian tail of one mapping iteration for gfx950: record sum -> per-Gaussian backward chain -> Adam in ONE pass over
// the Gaussians (dqo_rast_backward_adam, include/dqo_raster.h).
//
// Replaces, for the fused mapping iteration only (the drop-in backward keeps its kernels: its gradient rows are outputs),
//   record_sum_kernel + gaussian_backward_kernel (rast_backward.hip)  <- cuda_rasterizer/backward.cu:273-548 (K8, K9)
//   adam_kernel (map_fused.hip)                                       <- SLAM/gaussian_pointcloud.py:331-378, mapper.py:548, 812-829
// As three kernels the 59-float gradient row of every visible Gaussian made a round trip through HBM between two kernels that
// visit the same Gaussians (284 B written + 236 B re-read), and so did its 64-byte summed record.  Here a workgroup owns 128
// Gaussians (the thread -> Gaussian assignment of bin_count_kernel / record_sum_kernel, so their instance slots are one contiguous
// range):
//   A  lists the rows Adam has to touch (LDS: visible Gaussians and, in the exact sparse mode, those with non-zero moments),
//   B  sums its Gaussians' partial gradient records in the fixed (slot, quadrant) order of record_sum_kernel,
//   C  runs the per-Gaussian chain (dqo_gauss_chain.h: the same statements as gaussian_backward_kernel) and leaves each gradient row
//      in LDS — factored: dL/dsh[k][c] = w[k] * dRGB[c], so a row is 30 floats instead of 59 (15.5 KB per workgroup),
//   D  runs Adam's passes over the list with the gradient read from LDS (dqo_adam.h: the same statements as adam_kernel).
// Every float that reaches a parameter or a moment is produced by the same operations on the same operands as in the three-kernel
// path: bit-identical results (tests/test_gpu_fused_mapping.py::test_fused_tail_is_bitwise_the_three_kernels).
//
// What bounds it (round 3 measurements, cfg 3): the latency of a workgroup's dependent memory rounds — SQ counters: 60 % of
// the wave cycles parked at s_waitcnt, 20 % issue stalls, 20 % active; in-kernel stamps: list 9 %, chain inputs 14 %, record gather
// 27 %, chain 20 %, Adam 29 % of a wave's lifetime.  Not bandwidth (475 MB in 150 us), not the number of memory transactions (without
// any moment store — 20 % of all requests — the time is the same), not VALU or address-unit work (compacting the visible Gaussians,
// 16-byte SH loads: no change).  Register pressure decides the rest: the chain needs ~160 VGPRs, and a spill of a loaded value
// WAITS for the load (-DDQO_TAIL_WAVES=4: 372 bytes of scratch per lane, +25 % time).
#include "dqo_adam.h"
#include "dqo_common.h"
#include "dqo_gauss_chain.h"

namespace {

#ifndef DQO_TAIL_THREADS
#define DQO_TAIL_THREADS 128
#endif
constexpr int TAIL_THREADS = DQO_TAIL_THREADS;  // Gaussians (= threads) per workgroup; measured on cfg 3: 64 -> 160 us, 128 -> 150, 256 -> 152
// gradient row in LDS: [0..2] dL/dmean, [3..18] SH basis weights w, [19..21] dRGB, [22] dL/dopacity, [23..25] dL/dscales,
// [26..29] dL/drotation; stride 31 (odd: the per-thread row writes of phase C fall on distinct banks)
constexpr int ROW_MEAN = 0, ROW_W = 3, ROW_RGB = 19, ROW_OP = 22, ROW_SC = 23, ROW_ROT = 26, ROW_STRIDE = 31;

// Gradient source of Adam's passes: the rows phase C left in LDS (k = list row).  The SH gradient is formed here from its two factors
// — one IEEE multiply, the reference's per-coefficient statement (backward.cu:152-268) and what gaussian_backward_kernel stores.
struct AdamGradLds {
    const float* s_g;
    int used3;  // 3 x (coefficients of the active SH degree): elements beyond it have a zero gradient (rasterize_points.cu:204)
    __device__ __forceinline__ float xyz(int k, uint32_t j, size_t, bool) const { return s_g[k * ROW_STRIDE + ROW_MEAN + (int)j]; }
    __device__ __forceinline__ float scales(int k, uint32_t j, size_t, bool) const { return s_g[k * ROW_STRIDE + ROW_SC + (int)j]; }
    __device__ __forceinline__ float sh(int k, uint32_t j, uint32_t, uint32_t, bool) const {
#pragma clang fp contract(off)
        const uint32_t jc = j < (uint32_t)used3 ? j : 0u;
        const uint32_t c = (jc * 171u) >> 9, ch = jc - 3u * c;  // jc / 3, jc % 3 for jc < 256
        const float g = s_g[k * ROW_STRIDE + ROW_W + (int)c] * s_g[k * ROW_STRIDE + ROW_RGB + (int)ch];
        return j < (uint32_t)used3 ? g : 0.f;
    }
    // elements j .. j + 3 of the row (j a multiple of 4)
    __device__ __forceinline__ float4 sh4(int k, uint32_t j) const {
        return make_float4(sh(k, j, 0u, 0u, true), sh(k, j + 1u, 0u, 0u, true), sh(k, j + 2u, 0u, 0u, true), sh(k, j + 3u, 0u, 0u, true));
    }
    __device__ __forceinline__ float opacity(int k, uint32_t, bool) const { return s_g[k * ROW_STRIDE + ROW_OP]; }
    __device__ __forceinline__ float4 rot(int k, uint32_t, bool) const {
        const float* r = s_g + k * ROW_STRIDE + ROW_ROT;
        return make_float4(r[0], r[1], r[2], r[3]);
    }
    // the f_dc gradient = elements 0..2 of the SH row (a row without a gradient holds stale LDS: has_g decides first)
    __device__ __forceinline__ bool dc_nonzero(int k, uint32_t, bool has_g) const {
        return has_g && (sh(k, 0u, 0u, 0u, true) != 0.f || sh(k, 1u, 0u, 0u, true) != 0.f || sh(k, 2u, 0u, 0u, true) != 0.f);
    }
};

#ifndef DQO_TAIL_SH_U
#define DQO_TAIL_SH_U 4  // float4s per lane and trip of the SH pass
#endif
#ifndef DQO_TAIL_WAVES
#define DQO_TAIL_WAVES 4  // waves per SIMD the register allocation leaves room for (128 VGPRs since the SH row left the chain)
#endif
template <bool SPARSE, bool ATTACH>
__global__ __launch_bounds__(TAIL_THREADS, DQO_TAIL_WAVES) void gaussian_tail_kernel(const DqoView v, DqoGeomLayout g, const float* means3D,
                                                                        const float* scales, const float* rotations, const float* shs,
                                                                        const float4* __restrict__ partial,
                                                                        const uint32_t* __restrict__ valid, int64_t capacity, AdamArgs a,
                                                                        uint8_t* __restrict__ moment_live, const uint32_t frame_words,
                                                                        uint32_t* __restrict__ hist, const uint32_t hist_words) {
#pragma clang fp contract(off)
    // This kernel is the last consumer of the frame's counters (its blocks only read header.overflow): each block clears a slice of
    // counters | statistics lines | loss-tap counters for the NEXT frame (DqoRastCtx.frame_prezeroed: a replayed iteration then has no
    // zero-fill launch).  Before the overflow exit: an invalid frame must leave clean counters too.
    {
        const uint32_t per = (frame_words + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = blockIdx.x * per, z1 = min(frame_words, z0 + per);
        // (word 9 takes the stamp bin_count_kernel<true> asks for instead of a zero: "scalars, histogram and flags have been cleared")
        for (uint32_t i = z0 + threadIdx.x; i < z1; i += TAIL_THREADS) g.counters[i] = i == 9u ? DQO_CLEARED_STAMP : 0u;
    }
    // ... and of the tile histogram + tile flags (nothing reads them behind the sort / blend kernels of this frame): the next frame may
    // then start in bin_count_kernel<true>, without the preprocess_kernel launch that zeroes them (dqo_fuse_k1)
    {
        const uint32_t per = (hist_words + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = blockIdx.x * per, z1 = min(hist_words, z0 + per);
        for (uint32_t i = z0 + threadIdx.x; i < z1; i += TAIL_THREADS) hist[i] = 0u;
    }
    // Everything the block's head needs from memory goes out as ONE round (stamps of round 6, profiles/r06_tail_stamps.txt: the head took
    // 6.8 us of a block's 48 — the overflow word, then the step count, then the bias table's tag, then its values, then the per-Gaussian
    // loads: four dependent round trips): the frame's overflow word, the step count and the whole bias table unconditionally, the attach
    // gains, and the first per-Gaussian loads; the decisions follow below.
    const uint32_t ovf_ld = a.frame_header != nullptr ? a.frame_header->overflow : 0u;
    int step_ld = 0;
    float tb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.step_dev != nullptr) {
        step_ld = *a.step_dev;
        if (a.bias_table != nullptr) {
#pragma unroll
            for (int i = 0; i < 8; i++) tb[i] = a.bias_table[i];
        }
    }
    if (ATTACH) adam_attach_gains(a);
    // one buffer: phase B's slot staging (4 KB), then phase C / D's gradient rows (7.75 KB)
    __shared__ float4 s_buf[(TAIL_THREADS * ROW_STRIDE * 4 + 15) / 16];
    float4* const s_rec = s_buf;
    float* const s_g = reinterpret_cast<float*>(s_buf);
    static_assert(sizeof(float4) * TAIL_THREADS * 4 <= sizeof(s_buf), "the slot staging must fit the buffer");
    __shared__ uint32_t s_rows[TAIL_THREADS];  // Gaussian index | has-gradient << 31 | attach-loss member << 30
    __shared__ uint32_t s_lohi[2 * (TAIL_THREADS / 64)];  // per wave: lowest slot, end of the highest
    __shared__ int s_wave_n[TAIL_THREADS / 64];
    __shared__ float s_att[TAIL_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int idx = dqo_spread_index(blockIdx.x * TAIL_THREADS + tid, v.P);  // bin_count_kernel's thread -> Gaussian assignment
    const bool in_range = idx < v.P;

    // ---- A: first round of loads (rect, instance count, slot base, list flags); Adam's row list ----
    uint2 rc = make_uint2(0u, 0u);
    uint32_t base = 0, cnt = 0;
    bool live_m = false, att = false, trained = in_range;
    if (in_range) {
        rc = g.rect16[idx];
        cnt = g.tiles_touched[idx];
        base = g.slot_base[idx];
        if (SPARSE) live_m = moment_live[idx] != 0;
        if (ATTACH) att = a.attach_mask[idx] != 0;
        // DqoAdamStep.row_flags: a frozen row is rendered and back-propagated THROUGH (its entries shape the pixels' T), but it is no
        // parameter of this mapping call: no record sum, no chain, no Adam, no confidence (kernel-uniform branch, one byte per row)
        if (a.row_flags != nullptr) trained = (a.row_flags[idx] & DQO_ROW_FROZEN) == 0u;
    }
    // (the chain's camera matrices come from the kernel-argument segment: fetched in the same round)
    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        view[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.view[i])));
        proj[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.proj[i])));
    }
    // A frame flagged invalid by the forward must not train: nothing is read or written (adam_kernel's rule; block-uniform, no barrier yet)
    if (ovf_ld != 0u) return;
    if (a.step_dev != nullptr) {  // bias corrections of this step: DqoAdamStep.bias_table when it holds this step, computed otherwise
        float ss[7];              // (the same function either way: adam_bias_to_lds's rule, every thread, uniform)
        if (a.bias_table != nullptr && __float_as_int(tb[7]) == step_ld) {
#pragma unroll
            for (int i = 0; i < 7; i++) ss[i] = tb[i];
        } else {
            adam_bias_compute(a, step_ld, ss);
        }
        adam_bias_from_lds(a, ss);
    }
    if (!trained) cnt = 0u, live_m = false, att = false;
    // radii > 0 (backward.cu:285, 513; DqoAdamStep.radii)  <=>  the forward kept a non-empty tile rect for this Gaussian
    const bool visible = trained && ((rc.x >> 16) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.y & 0xffffu));
    const bool act = trained && (!SPARSE || visible || live_m);
    if (SPARSE && visible) moment_live[idx] = 1;  // only this thread ever looks at this byte
    const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
    // the workgroup's slot range (its Gaussians' slots are one contiguous range, rast_binning.hip): per wave the minimum of the bases and
    // the maximum of the ends through the crossbar-free butterflies of dqo_common.h, handed over with the row counts in ONE barrier —
    // 128 LDS atomics on two words and a second barrier until round 6 (2 us of the block's head)
    const uint32_t w_lo = ~dqo_wave_max_u32(cnt ? ~base : 0u, lane), w_hi = dqo_wave_max_u32(cnt ? base + cnt : 0u, lane);
    if (lane == 0) s_wave_n[wave] = (int)__popcll(am), s_lohi[2 * wave] = w_lo, s_lohi[2 * wave + 1] = w_hi;
    __syncthreads();  // (a one-wave workgroup's __syncthreads is a wave-level fence, not an s_barrier)
    int before = 0, n_rows = 0;
    uint32_t lo = 0xffffffffu, hi_all = 0u;
#pragma unroll
    for (int w = 0; w < TAIL_THREADS / 64; w++) {
        const int c = s_wave_n[w];
        before += w < wave ? c : 0;
        n_rows += c;
        lo = min(lo, s_lohi[2 * w]), hi_all = max(hi_all, s_lohi[2 * w + 1]);
    }
    const int my_row = before + (int)__popcll(am & ((1ull << lane) - 1ull));  // this Gaussian's list row (if act)
    // (s_rows is first read in phase D, behind the barrier that closes phase C: no barrier of its own)
    if (act) s_rows[my_row] = (uint32_t)idx | (visible ? 0x80000000u : 0u) | (att ? 0x40000000u : 0u);
    const uint32_t hi = (uint32_t)min((int64_t)hi_all, capacity);  // (an overflowed forward never gets here; belt and braces)

    // ---- second round of loads: everything the chain needs of this lane's Gaussian, issued as a whole before the record gather (the
    //      same round structure as gaussian_backward_kernel; in flight while phase B runs).  (Compacting the visible Gaussians of a
    //      256-thread block into as few waves as they fill — cfg 3: 39 % are visible — was built and measured: no gain; the chain is a
    //      long dependent instruction sequence whose duration does not depend on how many lanes run it.) ----
    DqoChainIn ci;
#pragma unroll
    for (int i = 0; i < 9; i++) ci.dd[i] = 0.f;
    ci.cop = make_float4(0.f, 0.f, 0.f, 0.f);
    ci.mx = ci.my = ci.mz = ci.sx = ci.sy = ci.sz = 0.f;
    ci.qt = make_float4(1.f, 0.f, 0.f, 0.f);
    ci.n_np = ci.pc = make_float4(0.f, 0.f, 0.f, 0.f);
    ci.cl = 0;
    if (visible) {
        ci.cop = g.conic_opacity[idx];
        ci.mx = means3D[3 * idx], ci.my = means3D[3 * idx + 1], ci.mz = means3D[3 * idx + 2];
        ci.sx = scales[3 * idx], ci.sy = scales[3 * idx + 1], ci.sz = scales[3 * idx + 2];
        ci.qt = reinterpret_cast<const float4*>(rotations)[idx];
        ci.n_np = g.normal_c[idx], ci.pc = g.point_c[idx], ci.cl = g.clamped[idx];
        // d(SH colour)/d(direction), left by the forward's preprocess_kernel (dqo_sh_dir_grad): 9 floats in place of the 48-float SH
        // row the chain held in registers across its whole length until round 4 (168 VGPRs -> three waves per SIMD)
        const float4* ddp = g.drgb_dir + 3 * (size_t)idx;
        const float4 d0 = ddp[0], d1 = ddp[1], d2 = ddp[2];
        ci.dd[0] = d0.x, ci.dd[1] = d0.y, ci.dd[2] = d0.z, ci.dd[3] = d1.x, ci.dd[4] = d1.y, ci.dd[5] = d1.z;
        ci.dd[6] = d2.x, ci.dd[7] = d2.y, ci.dd[8] = d2.z;
    }

    // ---- B: fixed-order sum of this lane's Gaussian's partial gradient records (record_sum_kernel's statements; one slot per thread and trip) ----
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (lo < hi) {  // (wave-uniform)
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t c0 = lo; c0 < hi; c0 += TAIL_THREADS) {
            const uint32_t slot = c0 + tid;
            if (slot < hi) {
                const uint32_t vw = valid[slot];
                const float4* p = partial + (size_t)slot * 16;
                // all sixteen loads unconditionally and back to back; the lanes of an invalid quadrant read one shared dummy record
                float4 r[4][4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4* src = ((vw >> (8 * q)) & 0xffu) ? p + 4 * q : partial;
#pragma unroll
                    for (int i = 0; i < 4; i++) r[q][i] = src[i];
                }
                float4 m0 = z, m1 = z, m2 = z, m3 = z;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t bq = (vw >> (8 * q)) & 0xffu;  // 1: floats 0..8 written, 3: depth-hit floats 9..13 as well
                    if (bq) {
                        const float4 r0 = r[q][0], r1 = r[q][1], r2 = r[q][2], r3 = r[q][3];
                        m0.x += r0.x, m0.y += r0.y, m0.z += r0.z, m0.w += r0.w;
                        m1.x += r1.x, m1.y += r1.y, m1.z += r1.z, m1.w += r1.w;
                        m2.x += r2.x;
                        if (bq & 2u) {
                            m2.y += r2.y, m2.z += r2.z, m2.w += r2.w;
                            m3.x += r3.x, m3.y += r3.y;
                        }
                    }
                }
                s_rec[tid * 4] = m0, s_rec[tid * 4 + 1] = m1, s_rec[tid * 4 + 2] = m2, s_rec[tid * 4 + 3] = m3;
            }
            __syncthreads();
            const uint32_t k0 = max(base, c0), k1 = min(base + cnt, min(c0 + (uint32_t)TAIL_THREADS, hi));
            for (uint32_t k = k0; k < k1; k++) {
                const float4 r0 = s_rec[(k - c0) * 4], r1 = s_rec[(k - c0) * 4 + 1], r2 = s_rec[(k - c0) * 4 + 2], r3 = s_rec[(k - c0) * 4 + 3];
                a0.x += r0.x, a0.y += r0.y, a0.z += r0.z, a0.w += r0.w;
                a1.x += r1.x, a1.y += r1.y, a1.z += r1.z, a1.w += r1.w;
                a2.x += r2.x, a2.y += r2.y, a2.z += r2.z, a2.w += r2.w;
                a3.x += r3.x, a3.y += r3.y, a3.z += r3.z, a3.w += r3.w;
            }
            __syncthreads();  // (the staging area is overwritten — by the next trip or by phase C's rows)
        }
    }
    if (n_rows == 0) {  // (wave-uniform) nothing to update: no visible Gaussian, no live moment
        if (ATTACH && a.attach_partial != nullptr && tid == 0) a.attach_partial[blockIdx.x] = 0.f;
        adam_take_ticket(a);
        return;
    }

    // ---- C: the per-Gaussian chain; the gradient row goes to LDS at the Gaussian's list row ----
    if (visible) {
        ci.a[0] = a0.x, ci.a[1] = a0.y, ci.a[2] = a0.z, ci.a[3] = a0.w;
        ci.a[4] = a1.x, ci.a[5] = a1.y, ci.a[6] = a1.z, ci.a[7] = a1.w;
        ci.a[8] = a2.x, ci.a[9] = a2.y, ci.a[10] = a2.z, ci.a[11] = a2.w;
        ci.a[12] = a3.x, ci.a[13] = a3.y, ci.a[14] = a3.z, ci.a[15] = a3.w;
        if (cnt == 0u) ci.cop = make_float4(0.f, 0.f, 0.f, 0.f);  // (a Gaussian without instances: gaussian_backward_kernel's rule)
        DqoChainOut co;
        dqo_gauss_chain(v, view, proj, ci, true, co);
        float* row = s_g + my_row * ROW_STRIDE;
        row[ROW_MEAN] = co.mean_g[0], row[ROW_MEAN + 1] = co.mean_g[1], row[ROW_MEAN + 2] = co.mean_g[2];
#pragma unroll
        for (int k = 0; k < 16; k++) row[ROW_W + k] = co.w[k];
        row[ROW_RGB] = co.dRGB[0], row[ROW_RGB + 1] = co.dRGB[1], row[ROW_RGB + 2] = co.dRGB[2];
        row[ROW_OP] = co.dop;
        row[ROW_SC] = co.dsc[0], row[ROW_SC + 1] = co.dsc[1], row[ROW_SC + 2] = co.dsc[2];
        row[ROW_ROT] = co.rot_g[0], row[ROW_ROT + 1] = co.rot_g[1], row[ROW_ROT + 2] = co.rot_g[2], row[ROW_ROT + 3] = co.rot_g[3];
    }
    __syncthreads();
    // ---- D: Adam over the wave's list, gradient from LDS (adam_kernel's statements) ----
    const int used = (v.D + 1) * (v.D + 1);
    // SH pass in float4s (rows of 48 floats, 16-byte aligned tensors), four per lane and trip: 1024 floats per wave and trip instead of
    // 256 — a wave's ~25 rows in 1-2 dependent rounds instead of 5 (-3 % on the kernel).  (Also issuing the loads of the two small
    // passes together with the first SH trip measured 4 % SLOWER, and twice that when it pushed the kernel into register spills.)
    const bool vec4 = v.M == 16 && ((reinterpret_cast<uintptr_t>(a.shs) | reinterpret_cast<uintptr_t>(a.m_shs) |
                                     reinterpret_cast<uintptr_t>(a.v_shs)) & 15u) == 0u;
    float att_sum;
    if (vec4) att_sum = adam_passes_tail<ATTACH, TAIL_THREADS, true, DQO_TAIL_SH_U>(a, s_rows, n_rows, AdamGradLds{s_g, 3 * used});
    else att_sum = adam_passes<ATTACH, TAIL_THREADS>(a, s_rows, n_rows, AdamGradLds{s_g, 3 * used});
    if (ATTACH && a.attach_partial != nullptr) {  // fixed-order sum of the attach loss (the reported "scale_loss")
        att_sum = adam_wave_red(att_sum);
        if (TAIL_THREADS == 64) {
            if (lane == 0) a.attach_partial[blockIdx.x] = att_sum;
        } else {
            if (lane == 0) s_att[wave] = att_sum;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
                for (int w = 0; w < TAIL_THREADS / 64; w++) t += s_att[w];
                a.attach_partial[blockIdx.x] = t;
            }
        }
    }
    adam_take_ticket(a);
}

// ------------------------------------------------------------------------------------------------------------------
// The drop-in backward's per-Gaussian half as ONE kernel: record sum -> per-Gaussian chain -> gradient ROWS, written to the caller's
// tensors in coalesced pieces.  Replaces record_sum_kernel + gaussian_backward_kernel (rast_backward.hip; kept behind
// DQO_ROWS_KERNEL=0 for A/B): those two sent every Gaussian's 64-byte summed record through HBM, and gaussian_backward_kernel wrote a
// Gaussian's 59 + 12 gradient floats from ONE thread — 71 store instructions per wave, each touching 64 different cache lines, for
// visible and culled rows alike (the culled rows, 61 % of cfg 3, are plain zeros the reference's API contract still wants written).
// Here phases A - C are gaussian_tail_kernel's (same block -> Gaussian mapping, same fixed-order sums, same chain: same bits), the
// gradient row of every Gaussian of the block waits in LDS (41 floats, factored SH), and phase D streams the block's rows out tensor
// by tensor: consecutive threads write consecutive floats, 16 consecutive Gaussians of a spread group at a time.
// ------------------------------------------------------------------------------------------------------------------
constexpr int RROW_MEAN = 0, RROW_W = 3, RROW_RGB = 19, RROW_OP = 22, RROW_SC = 23, RROW_ROT = 26, RROW_COL = 30, RROW_G2 = 33, RROW_CV = 35,
              RROW_STRIDE = 43;

__global__ __launch_bounds__(TAIL_THREADS, DQO_TAIL_WAVES) void gaussian_rows_kernel(const DqoView v, DqoGeomLayout g, const float* means3D,
                                                                    const float* scales, const float* rotations, const float* shs,
                                                                    const float4* __restrict__ partial,
                                                                    const uint32_t* __restrict__ valid, int64_t capacity, DqoRastGrads gr,
                                                                    const uint32_t frame_words, uint32_t* __restrict__ hist,
                                                                    const uint32_t hist_words) {
#pragma clang fp contract(off)
    // dqo_rast_backward on a context with frame_prezeroed (the drop-in op's pooled contexts): like gaussian_tail_kernel, the last
    // consumer of the frame's counters clears them, the tile histogram and the tile flags for the NEXT frame on the context and leaves
    // the stamp (frame_words == 0: nothing is cleared — the plain call)
    if (frame_words != 0u) {
        const uint32_t per = (frame_words + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = blockIdx.x * per, z1 = min(frame_words, z0 + per);
        for (uint32_t i = z0 + threadIdx.x; i < z1; i += TAIL_THREADS) g.counters[i] = i == 9u ? DQO_CLEARED_STAMP : 0u;
        const uint32_t perh = (hist_words + gridDim.x - 1) / gridDim.x;
        const uint32_t h0 = blockIdx.x * perh, h1 = min(hist_words, h0 + perh);
        for (uint32_t i = h0 + threadIdx.x; i < h1; i += TAIL_THREADS) hist[i] = 0u;
    }
    __shared__ float4 s_buf[(TAIL_THREADS * RROW_STRIDE * 4 + 15) / 16];
    float4* const s_rec = s_buf;
    float* const s_g = reinterpret_cast<float*>(s_buf);
    static_assert(sizeof(float4) * TAIL_THREADS * 4 <= sizeof(s_buf), "the slot staging must fit the buffer");
    __shared__ uint32_t s_lohi[2];
    __shared__ uint8_t s_vis[TAIL_THREADS];
    const int tid = threadIdx.x;
    if (tid == 0) s_lohi[0] = 0xffffffffu, s_lohi[1] = 0u;
    const int idx = dqo_spread_index(blockIdx.x * TAIL_THREADS + tid, v.P);
    const bool in_range = idx < v.P;
    uint2 rc = make_uint2(0u, 0u);
    uint32_t base = 0, cnt = 0;
    if (in_range) {
        rc = g.rect16[idx];
        cnt = g.tiles_touched[idx];
        base = g.slot_base[idx];
    }
    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        view[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.view[i])));
        proj[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.proj[i])));
    }
    const bool visible = in_range && ((rc.x >> 16) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.y & 0xffffu));
    s_vis[tid] = visible ? (uint8_t)1 : (uint8_t)0;
    __syncthreads();
    if (cnt) {
        atomicMin(&s_lohi[0], base);
        atomicMax(&s_lohi[1], base + cnt);
    }
    __syncthreads();
    const uint32_t lo = s_lohi[0];
    const uint32_t hi = (uint32_t)min((int64_t)s_lohi[1], capacity);
    const bool with_sh = shs != nullptr && gr.dL_dsh != nullptr;
    DqoChainIn ci;
#pragma unroll
    for (int i = 0; i < 9; i++) ci.dd[i] = 0.f;
    ci.cop = make_float4(0.f, 0.f, 0.f, 0.f);
    ci.mx = ci.my = ci.mz = ci.sx = ci.sy = ci.sz = 0.f;
    ci.qt = make_float4(1.f, 0.f, 0.f, 0.f);
    ci.n_np = ci.pc = make_float4(0.f, 0.f, 0.f, 0.f);
    ci.cl = 0;
    if (visible) {
        ci.cop = g.conic_opacity[idx];
        ci.mx = means3D[3 * idx], ci.my = means3D[3 * idx + 1], ci.mz = means3D[3 * idx + 2];
        ci.sx = scales[3 * idx], ci.sy = scales[3 * idx + 1], ci.sz = scales[3 * idx + 2];
        ci.qt = reinterpret_cast<const float4*>(rotations)[idx];
        ci.n_np = g.normal_c[idx], ci.pc = g.point_c[idx], ci.cl = g.clamped[idx];
        if (with_sh) {
            const float4* ddp = g.drgb_dir + 3 * (size_t)idx;
            const float4 d0 = ddp[0], d1 = ddp[1], d2 = ddp[2];
            ci.dd[0] = d0.x, ci.dd[1] = d0.y, ci.dd[2] = d0.z, ci.dd[3] = d1.x, ci.dd[4] = d1.y, ci.dd[5] = d1.z;
            ci.dd[6] = d2.x, ci.dd[7] = d2.y, ci.dd[8] = d2.z;
        }
    }
    // ---- B: fixed-order sum of the partial gradient records (gaussian_tail_kernel's statements) ----
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (lo < hi) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t c0 = lo; c0 < hi; c0 += TAIL_THREADS) {
            const uint32_t slot = c0 + tid;
            if (slot < hi) {
                const uint32_t vw = valid[slot];
                const float4* p = partial + (size_t)slot * 16;
                float4 r[4][4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4* src = ((vw >> (8 * q)) & 0xffu) ? p + 4 * q : partial;
#pragma unroll
                    for (int i = 0; i < 4; i++) r[q][i] = src[i];
                }
                float4 m0 = z, m1 = z, m2 = z, m3 = z;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t bq = (vw >> (8 * q)) & 0xffu;
                    if (bq) {
                        const float4 r0 = r[q][0], r1 = r[q][1], r2 = r[q][2], r3 = r[q][3];
                        m0.x += r0.x, m0.y += r0.y, m0.z += r0.z, m0.w += r0.w;
                        m1.x += r1.x, m1.y += r1.y, m1.z += r1.z, m1.w += r1.w;
                        m2.x += r2.x;
                        if (bq & 2u) {
                            m2.y += r2.y, m2.z += r2.z, m2.w += r2.w;
                            m3.x += r3.x, m3.y += r3.y;
                        }
                    }
                }
                s_rec[tid * 4] = m0, s_rec[tid * 4 + 1] = m1, s_rec[tid * 4 + 2] = m2, s_rec[tid * 4 + 3] = m3;
            }
            __syncthreads();
            const uint32_t k0 = max(base, c0), k1 = min(base + cnt, min(c0 + (uint32_t)TAIL_THREADS, hi));
            for (uint32_t k = k0; k < k1; k++) {
                const float4 r0 = s_rec[(k - c0) * 4], r1 = s_rec[(k - c0) * 4 + 1], r2 = s_rec[(k - c0) * 4 + 2], r3 = s_rec[(k - c0) * 4 + 3];
                a0.x += r0.x, a0.y += r0.y, a0.z += r0.z, a0.w += r0.w;
                a1.x += r1.x, a1.y += r1.y, a1.z += r1.z, a1.w += r1.w;
                a2.x += r2.x, a2.y += r2.y, a2.z += r2.z, a2.w += r2.w;
                a3.x += r3.x, a3.y += r3.y, a3.z += r3.z, a3.w += r3.w;
            }
            __syncthreads();
        }
    }
    // ---- C: the chain; the row goes to LDS at the thread's own position ----
    if (visible) {
        ci.a[0] = a0.x, ci.a[1] = a0.y, ci.a[2] = a0.z, ci.a[3] = a0.w;
        ci.a[4] = a1.x, ci.a[5] = a1.y, ci.a[6] = a1.z, ci.a[7] = a1.w;
        ci.a[8] = a2.x, ci.a[9] = a2.y, ci.a[10] = a2.z, ci.a[11] = a2.w;
        ci.a[12] = a3.x, ci.a[13] = a3.y, ci.a[14] = a3.z, ci.a[15] = a3.w;
        if (cnt == 0u) ci.cop = make_float4(0.f, 0.f, 0.f, 0.f);  // (a Gaussian without instances: gaussian_backward_kernel's rule)
        DqoChainOut co;
        dqo_gauss_chain(v, view, proj, ci, with_sh, co);
        float* row = s_g + tid * RROW_STRIDE;
        row[RROW_MEAN] = co.mean_g[0], row[RROW_MEAN + 1] = co.mean_g[1], row[RROW_MEAN + 2] = co.mean_g[2];
#pragma unroll
        for (int k = 0; k < 16; k++) row[RROW_W + k] = co.w[k];
        row[RROW_RGB] = co.dRGB[0], row[RROW_RGB + 1] = co.dRGB[1], row[RROW_RGB + 2] = co.dRGB[2];
        row[RROW_OP] = co.dop;
        row[RROW_SC] = co.dsc[0], row[RROW_SC + 1] = co.dsc[1], row[RROW_SC + 2] = co.dsc[2];
        row[RROW_ROT] = co.rot_g[0], row[RROW_ROT + 1] = co.rot_g[1], row[RROW_ROT + 2] = co.rot_g[2], row[RROW_ROT + 3] = co.rot_g[3];
        row[RROW_COL] = co.dcolr[0], row[RROW_COL + 1] = co.dcolr[1], row[RROW_COL + 2] = co.dcolr[2];
        row[RROW_G2] = co.g2x, row[RROW_G2 + 1] = co.g2y;
#pragma unroll
        for (int i = 0; i < 6; i++) row[RROW_CV + i] = co.dcv[i];
    }
    __syncthreads();
    // ---- D: the block's rows, tensor by tensor: element e of a tensor with rows of `len` floats belongs to block row e / len; the
    //      rows of a spread group of 16 Gaussians are consecutive in memory, so a wave writes runs of 16 x len floats ----
    const bool skip = gr.skip_culled_rows != 0;
    const int logical0 = blockIdx.x * TAIL_THREADS;
    // A spread group's 16 rows of `len` floats are 4 x len whole float4s (16-byte aligned: the group starts at a multiple of 16
    // Gaussians): thread q writes float4 q of the block's 8 groups, its four floats looked up in the LDS rows.  The last (partial)
    // group of the map, and the sparse-row mode (culled rows stay unwritten), go float by float.
    auto stream_rows = [&](float* out, const int len, auto value) {
        if (out == nullptr) return;
        const int per_group = 4 * len;  // float4s of one group
        for (int q = tid; q < (TAIL_THREADS / 16) * per_group; q += TAIL_THREADS) {
            const int grp = q / per_group, n = q - grp * per_group;
            const int gi0 = dqo_spread_index(logical0 + 16 * grp, v.P);
            if (gi0 >= v.P) continue;
            const bool whole = !skip && gi0 + 15 < v.P;
            float f[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = 4 * n + u, r = e / len, j = e - r * len;
                const int t = 16 * grp + r;
                const bool vis = s_vis[t] != 0;
                f[u] = vis ? value(s_g + t * RROW_STRIDE, j) : 0.f;
                if (!whole && gi0 + r < v.P && (vis || !skip)) out[(size_t)(gi0 + r) * len + j] = f[u];
            }
            if (whole) *reinterpret_cast<float4*>(out + (size_t)gi0 * len + 4 * n) = make_float4(f[0], f[1], f[2], f[3]);
        }
    };
    stream_rows(gr.dL_dmeans3D, 3, [](const float* row, int j) { return row[RROW_MEAN + j]; });
    if (gr.dL_dsh != nullptr && v.M > 0) {
        const int used3 = 3 * (v.D + 1) * (v.D + 1);
        const bool sh_on = with_sh;
        stream_rows(gr.dL_dsh, 3 * v.M, [used3, sh_on](const float* row, int j) {
#pragma clang fp contract(off)
            // dL/dsh[k][c] = w[k] * dRGB[c] (backward.cu:152-268); coefficients above the active degree keep the reference's zeros
            const int jc = j < used3 ? j : 0;
            const int k = jc / 3, c = jc - 3 * k;
            const float gsh = row[RROW_W + k] * row[RROW_RGB + c];
            return (sh_on && j < used3) ? gsh : 0.f;
        });
    }
    stream_rows(gr.dL_dcolors, 3, [](const float* row, int j) { return row[RROW_COL + j]; });
    stream_rows(gr.dL_dopacity, 1, [](const float* row, int) { return row[RROW_OP]; });
    stream_rows(gr.dL_dscales, 3, [](const float* row, int j) { return row[RROW_SC + j]; });
    stream_rows(gr.dL_drotations, 4, [](const float* row, int j) { return row[RROW_ROT + j]; });
    stream_rows(gr.dL_dcov3D, 6, [](const float* row, int j) { return row[RROW_CV + j]; });
    stream_rows(gr.dL_dmeans2D, 3, [](const float* row, int j) { return j < 2 ? row[RROW_G2 + j] : 0.f; });
}

}  // namespace
// the per-Gaussian half of dqo_rast_backward (dqo_launch_backward, rast_backward.hip)
int dqo_launch_gaussian_rows(const DqoView& v, const DqoGeomLayout& g, const DqoRastInputs* in, const DqoGradRec* recs, const uint8_t* valid,
                             int64_t cap, const DqoRastGrads& gr, hipStream_t s, uint32_t frame_words, uint32_t* hist, uint32_t hist_words) {
    const int blocks = dqo_spread_blocks(v.P) * (256 / TAIL_THREADS);
    DQO_LAUNCH("gaussian_rows_kernel", gaussian_rows_kernel, dim3(blocks), dim3(TAIL_THREADS), s, v, g, in->means3D, in->scales, in->rotations,
               in->shs, reinterpret_cast<const float4*>(recs), reinterpret_cast<const uint32_t*>(valid), cap, gr, frame_words, hist, hist_words);
    return DQO_OK;
}

int dqo_launch_blend_backward(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin, int T,
                              const float* dL_dcolor, const float* dL_ddepth, DqoGradRec* recs, uint8_t* valid, int64_t capacity,
                              const DqoTapDev& tap, const DqoGateDev& gate, int list_split, hipStream_t s);
int dqo_launch_adam_advance(int32_t* step_dev, const DqoRastHeader* frame_header, hipStream_t s);

// blend backward + the fused per-Gaussian tail.  The caller (dqo_rast_backward_adam) has checked the arguments.
int dqo_launch_backward_adam(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx, const float* dL_dcolor,
                             const float* dL_ddepth, const DqoAdamStep* st, void* ws, hipStream_t s) {
    if (p->P <= 0) return DQO_OK;
    const DqoView v = dqo_make_view(p, in);
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    DqoImageLayout img = dqo_image_layout(ctx->image, p->W, p->H);
    DqoBinLayout bin = dqo_bin_layout(ctx->binning, ctx->inst_capacity,
                                      dqo_list_cap(ctx->inst_capacity, p->W, p->H, ctx->tile_bucket_capacity), ctx->tile_bucket_capacity);
    const int T = v.gx * v.gy;
    const int64_t cap = (int64_t)ctx->inst_capacity;
    DqoGradRec* recs = (DqoGradRec*)ws;
    uint8_t* valid = reinterpret_cast<uint8_t*>(bin.rec_valid);  // zeroed by the forward
    const int blocks = dqo_spread_blocks(p->P) * (256 / TAIL_THREADS);  // TAIL_THREADS logical positions of the spread mapping each
    AdamArgs a;
    bool attach = false;
    int rc = dqo_adam_args(st, blocks, &a, &attach);
    if (rc) return rc;
    a.frame_header = g.header;  // the frame whose gradients this step consumes is this context's
    rc = dqo_launch_blend_backward(v, g, img, bin, T, dL_dcolor, dL_ddepth, recs, valid, cap, dqo_tap_dev(ctx->loss_tap),
                                   dqo_gate_dev(ctx->object_gate), dqo_list_split(ctx), s);
    if (rc) return rc;
    const float4* partial = reinterpret_cast<const float4*>(recs);
    const uint32_t* vw = reinterpret_cast<const uint32_t*>(valid);
    const uint32_t frame_words = (uint32_t)dqo_frame_scalar_words(ctx);  // counters .. (per-object) loss counters, cleared for the next frame
    const uint32_t hist_words = (uint32_t)((img.tile_flag + T) - img.tile_count);  // tile histogram (padded) + flags, likewise
#define DQO_TAIL(SP, AT)                                                                                                              \
    DQO_LAUNCH("gaussian_tail_kernel", (gaussian_tail_kernel<SP, AT>), dim3(blocks), dim3(TAIL_THREADS), s, v, g, in->means3D, in->scales, \
               in->rotations, in->shs, partial, vw, cap, a, st->moment_live, frame_words, img.tile_count, hist_words)
    if (st->moment_live != nullptr) {
        if (attach) DQO_TAIL(true, true);
        else DQO_TAIL(true, false);
    } else {
        if (attach) DQO_TAIL(false, true);
        else DQO_TAIL(false, false);
    }
#undef DQO_TAIL
    // (DqoAdamStep.block_ticket == NULL: the count is advanced by a launch of its own, as dqo_map_adam_step does)
    if (st->step_dev != nullptr && a.step_advance == nullptr) return dqo_launch_adam_advance(st->step_dev, g.header, s);
    return DQO_OK;
}
