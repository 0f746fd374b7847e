// Backward pass of the depth-aware Gaussian rasteriser for gfx950 (MI355X).
//
// Replaces (semantics, not structure) /root/reference/submodules/diff-gaussian-rasterizer-depth/
//   cuda_rasterizer/backward.cu:808-1066  renderCUDA_flat (+propagateRotationGrad :100-148)  -> blend_backward_kernel (rast_backward_blend.hip)
//   cuda_rasterizer/backward.cu:273-422   computeCov2DCUDA                                    \
//   cuda_rasterizer/backward.cu:492-548   preprocessCUDA (+SH bwd :152-268, cov3D bwd :426-487)/ -> gaussian_backward_kernel
//
// MI355X design: the reference issues ~10 scattered global float atomics per (pixel, Gaussian) pair plus 3-7 per hit
// pixel.  On gfx950 global float atomics execute memory-side and a wave instruction whose 64 lanes hit 64 different rows
// runs ~17x below the streaming rate (MI355X_MICROARCH.md, Global float atomics), so none are used here:
//   * one wave64 owns one 8x8 quadrant of a 16x16 tile (1 pixel per lane); every lane looks at the same list entry in
//     the same trip, so the per-entry gradient is a wave reduction, shared by seven entries (rast_backward_blend.hip);
//   * the wave stores ONE 64-byte partial record per live (quadrant, instance) pair at the instance's gaussian-major
//     slot (4 partial records per slot + a validity word);
//   * record_sum_kernel sums each Gaussian's valid partial records in a fixed order (bitwise reproducible);
//   * gaussian_backward_kernel finishes the chain rule (cov2D, projection, SH, cov3D, depth-hit Jacobians) per Gaussian.
#include <cstdlib>

#include "dqo_common.h"
#include "dqo_cull.h"
#include "dqo_gauss_chain.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Fixed-order sum of every Gaussian's gradient records: sums[idx] = sum over the Gaussian's instance slots (ascending) of
// the sum over the slot's valid per-quadrant partial records (quadrant order) — bitwise reproducible.
// A block owns 256 consecutive Gaussians; their slots are one contiguous, idx-ordered range (rast_binning.hip hands out
// slots in index order).  Per chunk of 256 slots, thread t merges slot c0 + t (validity word, then up to four 64-byte
// partial records, all loads in flight together) into LDS, and each Gaussian's thread then adds its own contiguous records
// out of LDS.  Kept apart from gaussian_backward_kernel on purpose: this part is a sparse gather whose latency is hidden by
// occupancy (few registers, 16 KB LDS), which the register-heavy fp64 chain rule kernel cannot provide.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void record_sum_kernel(int P, DqoGeomLayout g, const float4* __restrict__ partial,
                                                         const uint32_t* __restrict__ valid, float4* __restrict__ sums,
                                                         int64_t capacity) {
    constexpr int GB_CHUNK = 256;
    __shared__ float4 s_rec[GB_CHUNK * 4];
    __shared__ uint32_t s_lohi[2];
    const int tid = threadIdx.x;
    const int idx = dqo_spread_index(blockIdx.x * blockDim.x + tid, P);  // same block -> Gaussian assignment as bin_count_kernel
    uint32_t base = 0, cnt = 0;
    if (idx < P) base = g.slot_base[idx], cnt = g.tiles_touched[idx];
    if (tid == 0) s_lohi[0] = 0xffffffffu, s_lohi[1] = 0u;
    __syncthreads();
    if (cnt) {
        atomicMin(&s_lohi[0], base);
        atomicMax(&s_lohi[1], base + cnt);
    }
    __syncthreads();
    const uint32_t lo = s_lohi[0];
    const uint32_t hi = (uint32_t)min((int64_t)s_lohi[1], capacity);  // an overflowed (invalid) forward must not read out of bounds
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    const float4 z = a0;
    for (uint32_t c0 = lo; c0 < hi; c0 += GB_CHUNK) {
        const uint32_t slot = c0 + tid;
        if (slot < hi) {
            const uint32_t vw = valid[slot];
            const float4* p = partial + (size_t)slot * 16;
            // All sixteen loads are issued unconditionally and back to back: a load under a lane condition makes the compiler wait
            // for every earlier load first, which would put the four quadrant records (and their validity word) in series.  The
            // lanes of an invalid quadrant read one shared dummy record instead (one cached line per instruction, discarded).
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
        const uint32_t k0 = max(base, c0), k1 = min(base + cnt, min(c0 + (uint32_t)GB_CHUNK, hi));
        for (uint32_t k = k0; k < k1; k++) {
            const float4 r0 = s_rec[(k - c0) * 4], r1 = s_rec[(k - c0) * 4 + 1], r2 = s_rec[(k - c0) * 4 + 2], r3 = s_rec[(k - c0) * 4 + 3];
            a0.x += r0.x, a0.y += r0.y, a0.z += r0.z, a0.w += r0.w;
            a1.x += r1.x, a1.y += r1.y, a1.z += r1.z, a1.w += r1.w;
            a2.x += r2.x, a2.y += r2.y, a2.z += r2.z, a2.w += r2.w;
            a3.x += r3.x, a3.y += r3.y, a3.z += r3.z, a3.w += r3.w;
        }
        __syncthreads();
    }
    if (cnt) {
        float4* o = sums + (size_t)idx * 4;
        o[0] = a0, o[1] = a1, o[2] = a2, o[3] = a3;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-Gaussian backward: sum the instance records, then K8 (cov2D) + K9 (projection, SH, cov3D) in one pass.
// Writes every gradient element (zeros for culled Gaussians), so the caller's tensors can be torch.empty.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_backward_kernel(const DqoView v, DqoGeomLayout g,
                                                                const float* __restrict__ means3D, const float* __restrict__ scales,
                                                                const float* __restrict__ rotations, const float* __restrict__ shs,
                                                                const DqoGradRec* __restrict__ recs, int64_t capacity,
                                                                DqoRastGrads gr) {
    // Separate IEEE multiplies and adds throughout, in the reference's statement order (backward.cu:273-548): the cov2D-inverse ->
    // cov3D -> (scale, quaternion) chain is ill-conditioned for thin surfels (denom^2, b^2 by cancellation), so its result depends
    // on where the roundings fall; evaluated like this it rounds exactly where the oracle (and an uncontracted build of the
    // reference) rounds, and the remaining difference is the summation order of the per-pixel terms, which the reference's float
    // atomics do not fix either (quirk B10).
#pragma clang fp contract(off)
    // record_sum_kernel left the summed gradient record of every Gaussian that owns instances
    const int tid = threadIdx.x;
    const int idx = blockIdx.x * blockDim.x + tid;
    const bool in_range = idx < v.P;
    // Two rounds of loads, each issued as a whole: (rect, instance count) decide whether the Gaussian has any work; everything
    // the chain needs follows in one go.  (A load under a lane condition waits for all earlier loads: the original order —
    // count, then record, then rect, then parameters — put four memory latencies in series at the head of every wave.)
    uint2 rc = make_uint2(0u, 0u);
    uint32_t n_inst = 0;
    if (in_range) {
        rc = g.rect16[idx];
        n_inst = g.tiles_touched[idx];
    }
    // The two camera matrices travel with the first round and stay in scalar registers (wave-uniform): left where they are first
    // used — behind the first gradient stores, which may alias them as far as the compiler knows — they were a third memory round
    // trip in the middle of every wave.
    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        view[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.view[i])));
        proj[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.proj[i])));
    }
    // radii > 0 (backward.cu:285, 513)  <=>  the forward kept a non-empty tile rect for this Gaussian
    const bool visible = in_range && ((rc.x >> 16) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.y & 0xffffu));
    DqoChainIn ci;  // the chain's inputs are loaded straight into its argument (dqo_gauss_chain.h)
    float(&a)[16] = ci.a;
    float4& cop = ci.cop;  // (conic, opacity) of the forward
    float &mx = ci.mx, &my = ci.my, &mz = ci.mz, &sx = ci.sx, &sy = ci.sy, &sz = ci.sz;
    float4& qt = ci.qt;
#pragma unroll
    for (int i = 0; i < 9; i++) ci.dd[i] = 0.f;
    float4 &n_np = ci.n_np, &pc = ci.pc;
    uint32_t& cl = ci.cl;
    cop = make_float4(0.f, 0.f, 0.f, 0.f);
    mx = my = mz = sx = sy = sz = 0.f;
    qt = make_float4(1.f, 0.f, 0.f, 0.f);
    n_np = pc = make_float4(0.f, 0.f, 0.f, 0.f);
    cl = 0;
    if (visible) {
        const float4* r = reinterpret_cast<const float4*>(recs) + (size_t)idx * 4;
        const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];  // (stale memory for a Gaussian without instances: zeroed below)
        cop = g.conic_opacity[idx];
        mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
        sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
        qt = reinterpret_cast<const float4*>(rotations)[idx];
        n_np = g.normal_c[idx], pc = g.point_c[idx], cl = g.clamped[idx];  // (depth-hit chain / SH clamp flags: same round)
        if (shs != nullptr && gr.dL_dsh != nullptr) {
            // d(SH colour)/d(direction) of the forward (dqo_sh_dir_grad): 9 floats in place of the 48-float SH row
            const float4* ddp = g.drgb_dir + 3 * (size_t)idx;
            const float4 d0 = ddp[0], d1 = ddp[1], d2 = ddp[2];
            ci.dd[0] = d0.x, ci.dd[1] = d0.y, ci.dd[2] = d0.z, ci.dd[3] = d1.x, ci.dd[4] = d1.y, ci.dd[5] = d1.z;
            ci.dd[6] = d2.x, ci.dd[7] = d2.y, ci.dd[8] = d2.z;
        }
        a[0] = r0.x, a[1] = r0.y, a[2] = r0.z, a[3] = r0.w;
        a[4] = r1.x, a[5] = r1.y, a[6] = r1.z, a[7] = r1.w;
        a[8] = r2.x, a[9] = r2.y, a[10] = r2.z, a[11] = r2.w;
        a[12] = r3.x, a[13] = r3.y, a[14] = r3.z, a[15] = r3.w;
    }
    if (n_inst == 0u) {  // record_sum_kernel only leaves a summed record for Gaussians that own instances
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = 0.f;
        cop = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!in_range) return;
    const int M = v.M, D = v.D;
    float* dm = gr.dL_dmeans3D + 3 * (size_t)idx;
    float* dsh = gr.dL_dsh ? gr.dL_dsh + (size_t)idx * M * 3 : nullptr;
    float* dcol = gr.dL_dcolors + 3 * (size_t)idx;
    float* dsc = gr.dL_dscales + 3 * (size_t)idx;
    float* drot = gr.dL_drotations + 4 * (size_t)idx;
    float* dcov = gr.dL_dcov3D + 6 * (size_t)idx;
    float* dm2 = gr.dL_dmeans2D + 3 * (size_t)idx;
    if (!visible) {
        if (gr.skip_culled_rows) return;  // the consumer knows the row is zero from radii (DqoRastGrads)
        dm[0] = dm[1] = dm[2] = 0.f;
        if (dsh)
            for (int i = 0; i < 3 * M; i++) dsh[i] = 0.f;
        if (gr.dL_dcolors) dcol[0] = dcol[1] = dcol[2] = 0.f;
        gr.dL_dopacity[idx] = 0.f;
        dsc[0] = dsc[1] = dsc[2] = 0.f;
        drot[0] = drot[1] = drot[2] = drot[3] = 0.f;
        if (gr.dL_dcov3D)
            for (int i = 0; i < 6; i++) dcov[i] = 0.f;
        if (gr.dL_dmeans2D) dm2[0] = dm2[1] = dm2[2] = 0.f;
        return;
    }
    DqoChainOut co;
    const bool with_sh = shs != nullptr && dsh != nullptr;
    dqo_gauss_chain(v, view, proj, ci, with_sh, co);
    gr.dL_dopacity[idx] = co.dop;
    if (gr.dL_dcolors) dcol[0] = co.dcolr[0], dcol[1] = co.dcolr[1], dcol[2] = co.dcolr[2];
    if (gr.dL_dmeans2D) dm2[0] = co.g2x, dm2[1] = co.g2y, dm2[2] = 0.f;
    if (gr.dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = co.dcv[i];
    }
    if (with_sh) {
        // dL/dsh[k][c] = w[k] * dRGB[c] (backward.cu:152-268); coefficients above the active degree keep the reference's zero
        // initialisation (rasterize_points.cu:204)
        const int used = (D + 1) * (D + 1);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < used) dsh[3 * k] = co.w[k] * co.dRGB[0], dsh[3 * k + 1] = co.w[k] * co.dRGB[1], dsh[3 * k + 2] = co.w[k] * co.dRGB[2];
        }
        for (int k = used; k < M; k++) dsh[3 * k] = dsh[3 * k + 1] = dsh[3 * k + 2] = 0.f;
    }
    dsc[0] = co.dsc[0], dsc[1] = co.dsc[1], dsc[2] = co.dsc[2];
    dm[0] = co.mean_g[0], dm[1] = co.mean_g[1], dm[2] = co.mean_g[2];
    drot[0] = co.rot_g[0], drot[1] = co.rot_g[1], drot[2] = co.rot_g[2], drot[3] = co.rot_g[3];
}

}  // namespace

int dqo_launch_blend_backward(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin, int T,
                              const float* dL_dcolor, const float* dL_ddepth, DqoGradRec* recs, uint8_t* valid, int64_t capacity,
                              const DqoTapDev& tap, const DqoGateDev& gate, int list_split, hipStream_t s);
int dqo_launch_gaussian_rows(const DqoView& v, const DqoGeomLayout& g, const DqoRastInputs* in, const DqoGradRec* recs, const uint8_t* valid,
                             int64_t cap, const DqoRastGrads& gr, hipStream_t s, uint32_t frame_words, uint32_t* hist, uint32_t hist_words);

int dqo_launch_backward(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx, const float* dL_dcolor,
                        const float* dL_ddepth, const int32_t* hit_image, DqoRastGrads* gr, void* ws, size_t ws_bytes, hipStream_t s) {
    (void)hit_image;  // the hit Gaussian is recovered from its list position kept in the image context
    (void)ws_bytes;
    if (p->P <= 0) return DQO_OK;
    const DqoView v = dqo_make_view(p, in);
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    DqoImageLayout img = dqo_image_layout(ctx->image, p->W, p->H);
    DqoBinLayout bin = dqo_bin_layout(ctx->binning, ctx->inst_capacity,
                                      dqo_list_cap(ctx->inst_capacity, p->W, p->H, ctx->tile_bucket_capacity), ctx->tile_bucket_capacity);
    const int T = v.gx * v.gy;
    const int64_t cap = (int64_t)ctx->inst_capacity;
    DqoGradRec* recs = (DqoGradRec*)ws;
    uint8_t* valid = reinterpret_cast<uint8_t*>(bin.rec_valid);  // zeroed by the forward (bin_place_kernel)
    int rc = dqo_launch_blend_backward(v, g, img, bin, T, dL_dcolor, dL_ddepth, recs, valid, cap, dqo_tap_dev(ctx->loss_tap),
                                       dqo_gate_dev(ctx->object_gate), dqo_list_split(ctx), s);
    if (rc) return rc;
    // the per-Gaussian half: ONE kernel (record sum -> chain -> coalesced gradient rows, map_fused_tail.hip); DQO_ROWS_KERNEL=0
    // (environment, read once) keeps the two kernels below for A/B
    static const bool rows_kernel = [] {
        const char* e = getenv("DQO_ROWS_KERNEL");
        return e == nullptr || atoi(e) != 0;
    }();
    // DqoRastCtx.frame_prezeroed given to THIS call: the per-Gaussian kernel, the last consumer of the frame's counters, clears them
    // (+ tile histogram and flags) for the next forward on the context and leaves the stamp — what dqo_rast_backward_adam always does.
    // Not with list_split: a second backward over the same forward (retain_graph) would find the long-list queue's counters gone.
    const bool clear = ctx->frame_prezeroed != 0 && dqo_list_split(ctx) == 0 && rows_kernel;
    if (rows_kernel)
        return dqo_launch_gaussian_rows(v, g, in, recs, valid, cap, *gr, s, clear ? (uint32_t)dqo_frame_scalar_words(ctx) : 0u, img.tile_count,
                                        clear ? (uint32_t)((img.tile_flag + T) - img.tile_count) : 0u);
    DqoGradRec* sums = reinterpret_cast<DqoGradRec*>(g.grad_sum);  // [P], lives in the forward's geometry buffer
    DQO_LAUNCH("record_sum_kernel", record_sum_kernel, dim3(dqo_spread_blocks(p->P)), dim3(256), s, p->P, g, reinterpret_cast<const float4*>(recs),
               reinterpret_cast<const uint32_t*>(valid), reinterpret_cast<float4*>(sums), cap);
    DQO_LAUNCH("gaussian_backward_kernel", gaussian_backward_kernel, dim3((p->P + 255) / 256), dim3(256), s, v, g, in->means3D, in->scales,
               in->rotations, in->shs, sums, cap, *gr);
    return DQO_OK;
}
