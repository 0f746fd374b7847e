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
#include "dqo_common.h"
#include "dqo_cull.h"

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
__constant__ float bSH_C0 = 0.28209479177387814f;
__constant__ float bSH_C1 = 0.4886025119029199f;
__constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

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
    float a[16];
    float4 cop = make_float4(0.f, 0.f, 0.f, 0.f);  // (conic, opacity) of the forward
    float mx = 0.f, my = 0.f, mz = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    float4 qt = make_float4(1.f, 0.f, 0.f, 0.f);
    float sh[48];  // SH coefficients 1.. of the active degree (coefficient 0 has no direction gradient)
    float4 n_np = make_float4(0.f, 0.f, 0.f, 0.f), pc = n_np;
    uint32_t cl = 0;
    if (visible) {
        const float4* r = reinterpret_cast<const float4*>(recs) + (size_t)idx * 4;
        const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];  // (stale memory for a Gaussian without instances: zeroed below)
        cop = g.conic_opacity[idx];
        mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
        sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
        qt = reinterpret_cast<const float4*>(rotations)[idx];
        n_np = g.normal_c[idx], pc = g.point_c[idx], cl = g.clamped[idx];  // (depth-hit chain / SH clamp flags: same round)
        if (shs != nullptr && gr.dL_dsh != nullptr) {
            // the SH coefficients of the active degree in the same round (they are only needed at the end of the chain; fetched
            // there, inside the per-degree blocks, they would arrive in three waited-for groups)
            const float* shp = shs + (size_t)idx * v.M * 3;
            if (v.D >= 3) {
#pragma unroll
                for (int i = 3; i < 48; i++) sh[i] = shp[i];
            } else if (v.D == 2) {
#pragma unroll
                for (int i = 3; i < 27; i++) sh[i] = shp[i];
            } else if (v.D == 1) {
#pragma unroll
                for (int i = 3; i < 12; i++) sh[i] = shp[i];
            }
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
    const float dcolr[3] = {a[0], a[1], a[2]};
    // pixel moments of q = G * dL/dalpha summed by the blend kernel (DqoGradRec) -> gradients w.r.t. the 2D mean, the conic and
    // the opacity (backward.cu:964-994): the per-Gaussian constants are applied here, once
    const float g2x = -cop.w * (cop.x * a[3] + cop.y * a[4]) * (0.5f * v.W);
    const float g2y = -cop.w * (cop.z * a[4] + cop.y * a[3]) * (0.5f * v.H);
    const float dcx = -0.5f * cop.w * a[5], dcy = -0.5f * cop.w * a[6], dcz = -0.5f * cop.w * a[7];
    float mean_g[3] = {0.f, 0.f, 0.f};
    float rot_g[4] = {0.f, 0.f, 0.f, 0.f};
    gr.dL_dopacity[idx] = a[8];
    if (gr.dL_dcolors) dcol[0] = dcolr[0], dcol[1] = dcolr[1], dcol[2] = dcolr[2];
    if (gr.dL_dmeans2D) dm2[0] = g2x, dm2[1] = g2y, dm2[2] = 0.f;

    // `real` = float reproduces the reference's arithmetic (the parity target).  -DDQO_BWD_CHAIN_FP64 evaluates the chain in double
    // instead: closer to the exact derivative of the same formulas (DESIGN.md §2 quantifies both), but not what the reference
    // computes; per Gaussian, not per pixel, so the cost is invisible either way.
#ifdef DQO_BWD_CHAIN_FP64
    typedef double real;
#else
    typedef float real;
#endif
#define RL(v) ((real)(v))
    const real r = qt.x, x = qt.y, y = qt.z, z = qt.w;
    real Rm[3][3];
    Rm[0][0] = RL(1) - RL(2) * (y * y + z * z), Rm[0][1] = RL(2) * (x * y - r * z), Rm[0][2] = RL(2) * (x * z + r * y);
    Rm[1][0] = RL(2) * (x * y + r * z), Rm[1][1] = RL(1) - RL(2) * (x * x + z * z), Rm[1][2] = RL(2) * (y * z - r * x);
    Rm[2][0] = RL(2) * (x * z - r * y), Rm[2][1] = RL(2) * (y * z + r * x), Rm[2][2] = RL(1) - RL(2) * (x * x + y * y);
    // ---- depth-hit gradient (backward.cu:997-1065 + propagateRotationGrad :100-148) ----
    // The blend kernel delivered the pixel sums hit[0..4] (DqoGradRec); everything that is constant per Gaussian — surfel
    // normal n_c, camera-space point p_c, view matrix, d(normal)/d(quaternion) — is applied here, once:
    //   dL/dmean3D = hit1 * V^T n_c + hit0 * V^T e_z,   dL/dn_c = p_c * hit1 - (n_c . p_c) * hit[2..4],   dL/dq = (dn_w/dq)^T V^T dL/dn_c
    if (a[9] != 0.f || a[10] != 0.f || a[11] != 0.f || a[12] != 0.f || a[13] != 0.f) {
        const real h0 = a[9], h1 = a[10], h2x = a[11], h2y = a[12], h2z = a[13];
        const real nx = n_np.x, ny = n_np.y, nz = n_np.z;
        const real np = nx * pc.x + ny * pc.y + nz * pc.z;
        mean_g[0] = (float)(h1 * (nx * view[0] + ny * view[1] + nz * view[2]) + h0 * view[2]);
        mean_g[1] = (float)(h1 * (nx * view[4] + ny * view[5] + nz * view[6]) + h0 * view[6]);
        mean_g[2] = (float)(h1 * (nx * view[8] + ny * view[9] + nz * view[10]) + h0 * view[10]);
        const real n1c = pc.x * h1 - np * h2x, n2c = pc.y * h1 - np * h2y, n3c = pc.z * h1 - np * h2z;
        const real n1w = n1c * view[0] + n2c * view[1] + n3c * view[2];
        const real n2w = n1c * view[4] + n2c * view[5] + n3c * view[6];
        const real n3w = n1c * view[8] + n2c * view[9] + n3c * view[10];
        // the surfel normal is column `axis` of R(q), axis = the smallest raw scale (forward.cu:54-74)
        const int axis = (sx <= sy && sx <= sz) ? 0 : ((sy <= sx && sy <= sz) ? 1 : 2);
        const real q0 = r, q1 = x, q2 = y, q3 = z;
        real d0[3], d1[3], d2[3], d3[3];
        if (axis == 0) {
            d0[0] = RL(0), d0[1] = RL(2) * q3, d0[2] = -RL(2) * q2;
            d1[0] = RL(0), d1[1] = RL(2) * q2, d1[2] = RL(2) * q3;
            d2[0] = -RL(4) * q2, d2[1] = RL(2) * q1, d2[2] = -RL(2) * q0;
            d3[0] = -RL(4) * q3, d3[1] = RL(2) * q0, d3[2] = RL(2) * q1;
        } else if (axis == 1) {
            d0[0] = -RL(2) * q3, d0[1] = RL(0), d0[2] = RL(2) * q1;
            d1[0] = RL(2) * q2, d1[1] = -RL(4) * q1, d1[2] = RL(2) * q0;
            d2[0] = RL(2) * q1, d2[1] = RL(0), d2[2] = RL(2) * q3;
            d3[0] = -RL(2) * q0, d3[1] = -RL(4) * q3, d3[2] = RL(2) * q2;
        } else {
            d0[0] = RL(2) * q2, d0[1] = -RL(2) * q1, d0[2] = RL(0);
            d1[0] = RL(2) * q3, d1[1] = -RL(2) * q0, d1[2] = -RL(4) * q1;
            d2[0] = RL(2) * q0, d2[1] = RL(2) * q3, d2[2] = -RL(4) * q2;
            d3[0] = RL(2) * q1, d3[1] = RL(2) * q2, d3[2] = RL(0);
        }
        rot_g[0] = (float)(n1w * d0[0] + n2w * d0[1] + n3w * d0[2]);
        rot_g[1] = (float)(n1w * d1[0] + n2w * d1[1] + n3w * d1[2]);
        rot_g[2] = (float)(n1w * d2[0] + n2w * d2[1] + n3w * d2[2]);
        rot_g[3] = (float)(n1w * d3[0] + n2w * d3[1] + n3w * d3[2]);
    }
    const real s[3] = {RL(v.scale_mod) * RL(sx), RL(v.scale_mod) * RL(sy), RL(v.scale_mod) * RL(sz)};
    real Mm[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) Mm[k][i] = s[k] * Rm[i][k];
    real c3[6];
    {
        int o = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i; j < 3; j++) c3[o++] = Mm[0][i] * Mm[0][j] + Mm[1][i] * Mm[1][j] + Mm[2][i] * Mm[2][j];
    }
    // ---- K8 computeCov2DCUDA, backward.cu:273-422 ----
    const real tvx0 = RL(view[0]) * RL(mx) + RL(view[4]) * RL(my) + RL(view[8]) * RL(mz) + RL(view[12]);
    const real tvy0 = RL(view[1]) * RL(mx) + RL(view[5]) * RL(my) + RL(view[9]) * RL(mz) + RL(view[13]);
    const real tvz = RL(view[2]) * RL(mx) + RL(view[6]) * RL(my) + RL(view[10]) * RL(mz) + RL(view[14]);
    const real limx = RL(1.3f) * RL(v.tanfovx), limy = RL(1.3f) * RL(v.tanfovy);
    const real txtz = tvx0 / tvz, tytz = tvy0 / tvz;
    const real tx = (txtz > limx ? limx : (txtz < -limx ? -limx : txtz)) * tvz;  // min(lim, max(-lim, t)), backward.cu:300-301
    const real ty = (tytz > limy ? limy : (tytz < -limy ? -limy : tytz)) * tvz;
    const real x_grad_mul = (txtz < -limx || txtz > limx) ? RL(0) : RL(1);
    const real y_grad_mul = (tytz < -limy || tytz > limy) ? RL(0) : RL(1);
    const real fx = v.focal_x, fy = v.focal_y;
    const real J00 = fx / tvz, J02 = -(fx * tx) / (tvz * tvz);
    const real J11 = fy / tvz, J12 = -(fy * ty) / (tvz * tvz);
    real A0[3], A1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        A0[j] = J00 * RL(view[j * 4 + 0]) + J02 * RL(view[j * 4 + 2]);
        A1[j] = J11 * RL(view[j * 4 + 1]) + J12 * RL(view[j * 4 + 2]);
    }
    const real V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    real A0V[3], A1V[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        A0V[j] = A0[0] * V[j][0] + A0[1] * V[j][1] + A0[2] * V[j][2];
        A1V[j] = A1[0] * V[j][0] + A1[1] * V[j][1] + A1[2] * V[j][2];
    }
    const real ca = A0[0] * A0V[0] + A0[1] * A0V[1] + A0[2] * A0V[2] + RL(0.3f);
    const real cb = A0[0] * A1V[0] + A0[1] * A1V[1] + A0[2] * A1V[2];
    const real cc = A1[0] * A1V[0] + A1[1] * A1V[1] + A1[2] * A1V[2] + RL(0.3f);
    const real denom = ca * cc - cb * cb;
    real dL_da = 0, dL_db = 0, dL_dc = 0;
    const real denom2inv = RL(1) / ((denom * denom) + RL(0.0000001f));
    real dcv[6];
    const real dcx_ = RL(dcx), dcy_ = RL(dcy), dcz_ = RL(dcz);
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dcx_ + RL(2) * cb * cc * dcy_ + (denom - ca * cc) * dcz_);
        dL_dc = denom2inv * (-ca * ca * dcz_ + RL(2) * ca * cb * dcy_ + (denom - ca * cc) * dcx_);
        dL_db = denom2inv * RL(2) * (cb * cc * dcx_ - (denom + RL(2) * cb * cb) * dcy_ + ca * cb * dcz_);
        dcv[0] = A0[0] * A0[0] * dL_da + A0[0] * A1[0] * dL_db + A1[0] * A1[0] * dL_dc;
        dcv[3] = A0[1] * A0[1] * dL_da + A0[1] * A1[1] * dL_db + A1[1] * A1[1] * dL_dc;
        dcv[5] = A0[2] * A0[2] * dL_da + A0[2] * A1[2] * dL_db + A1[2] * A1[2] * dL_dc;
        dcv[1] = RL(2) * A0[0] * A0[1] * dL_da + (A0[0] * A1[1] + A0[1] * A1[0]) * dL_db + RL(2) * A1[0] * A1[1] * dL_dc;
        dcv[2] = RL(2) * A0[0] * A0[2] * dL_da + (A0[0] * A1[2] + A0[2] * A1[0]) * dL_db + RL(2) * A1[0] * A1[2] * dL_dc;
        dcv[4] = RL(2) * A0[2] * A0[1] * dL_da + (A0[1] * A1[2] + A0[2] * A1[1]) * dL_db + RL(2) * A1[1] * A1[2] * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) dcv[i] = RL(0);
    }
    if (gr.dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = (float)dcv[i];
    }
    real dT0[3], dT1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        dT0[j] = RL(2) * A0V[j] * dL_da + A1V[j] * dL_db;
        dT1[j] = RL(2) * A1V[j] * dL_dc + A0V[j] * dL_db;
    }
    const real dJ00 = RL(view[0]) * dT0[0] + RL(view[4]) * dT0[1] + RL(view[8]) * dT0[2];
    const real dJ02 = RL(view[2]) * dT0[0] + RL(view[6]) * dT0[1] + RL(view[10]) * dT0[2];
    const real dJ11 = RL(view[1]) * dT1[0] + RL(view[5]) * dT1[1] + RL(view[9]) * dT1[2];
    const real dJ12 = RL(view[2]) * dT1[0] + RL(view[6]) * dT1[1] + RL(view[10]) * dT1[2];
    const real tzi = RL(1) / tvz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
    const real dL_dtx = x_grad_mul * -fx * tz2 * dJ02;
    const real dL_dty = y_grad_mul * -fy * tz2 * dJ12;
    const real dL_dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (RL(2) * fx * tx) * tz3 * dJ02 + (RL(2) * fy * ty) * tz3 * dJ12;
    mean_g[0] += (float)(RL(view[0]) * dL_dtx + RL(view[1]) * dL_dty + RL(view[2]) * dL_dtz);
    mean_g[1] += (float)(RL(view[4]) * dL_dtx + RL(view[5]) * dL_dty + RL(view[6]) * dL_dtz);
    mean_g[2] += (float)(RL(view[8]) * dL_dtx + RL(view[9]) * dL_dty + RL(view[10]) * dL_dtz);

    // ---- K9 preprocessCUDA backward, backward.cu:492-548 ----
    const float hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
    const float m_w = 1.0f / (hw + 0.0000001f);
    const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
    mean_g[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    mean_g[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    mean_g[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

    if (shs != nullptr && dsh != nullptr) {
        // SH backward, backward.cu:152-268
        const float dox = mx - v.campos[0], doy = my - v.campos[1], doz = mz - v.campos[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float dx = dox / len, dy = doy / len, dz = doz / len;
        const float dRGB[3] = {dcolr[0] * ((cl & 1u) ? 0.f : 1.f), dcolr[1] * ((cl & 2u) ? 0.f : 1.f), dcolr[2] * ((cl & 4u) ? 0.f : 1.f)};
        float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
        auto setd = [&](int k, float w) {
            dsh[3 * k] = w * dRGB[0];
            dsh[3 * k + 1] = w * dRGB[1];
            dsh[3 * k + 2] = w * dRGB[2];
        };
        const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
        setd(0, bSH_C0);
        if (D > 0) {
            setd(1, -bSH_C1 * dy);
            setd(2, bSH_C1 * dz);
            setd(3, -bSH_C1 * dx);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] = -bSH_C1 * sh[9 + ch];
                dRGBdy[ch] = -bSH_C1 * sh[3 + ch];
                dRGBdz[ch] = bSH_C1 * sh[6 + ch];
            }
            if (D > 1) {
                setd(4, bSH_C2[0] * xy);
                setd(5, bSH_C2[1] * yz);
                setd(6, bSH_C2[2] * (2.f * zz - xx - yy));
                setd(7, bSH_C2[3] * xz);
                setd(8, bSH_C2[4] * (xx - yy));
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += bSH_C2[0] * dy * sh[12 + ch] + bSH_C2[2] * 2.f * -dx * sh[18 + ch] + bSH_C2[3] * dz * sh[21 + ch] +
                                  bSH_C2[4] * 2.f * dx * sh[24 + ch];
                    dRGBdy[ch] += bSH_C2[0] * dx * sh[12 + ch] + bSH_C2[1] * dz * sh[15 + ch] + bSH_C2[2] * 2.f * -dy * sh[18 + ch] +
                                  bSH_C2[4] * 2.f * -dy * sh[24 + ch];
                    dRGBdz[ch] += bSH_C2[1] * dy * sh[15 + ch] + bSH_C2[2] * 2.f * 2.f * dz * sh[18 + ch] + bSH_C2[3] * dx * sh[21 + ch];
                }
                if (D > 2) {
                    setd(9, bSH_C3[0] * dy * (3.f * xx - yy));
                    setd(10, bSH_C3[1] * xy * dz);
                    setd(11, bSH_C3[2] * dy * (4.f * zz - xx - yy));
                    setd(12, bSH_C3[3] * dz * (2.f * zz - 3.f * xx - 3.f * yy));
                    setd(13, bSH_C3[4] * dx * (4.f * zz - xx - yy));
                    setd(14, bSH_C3[5] * dz * (xx - yy));
                    setd(15, bSH_C3[6] * dx * (xx - 3.f * yy));
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        dRGBdx[ch] += (bSH_C3[0] * sh[27 + ch] * 3.f * 2.f * xy + bSH_C3[1] * sh[30 + ch] * yz +
                                       bSH_C3[2] * sh[33 + ch] * -2.f * xy + bSH_C3[3] * sh[36 + ch] * -3.f * 2.f * xz +
                                       bSH_C3[4] * sh[39 + ch] * (-3.f * xx + 4.f * zz - yy) + bSH_C3[5] * sh[42 + ch] * 2.f * xz +
                                       bSH_C3[6] * sh[45 + ch] * 3.f * (xx - yy));
                        dRGBdy[ch] += (bSH_C3[0] * sh[27 + ch] * 3.f * (xx - yy) + bSH_C3[1] * sh[30 + ch] * xz +
                                       bSH_C3[2] * sh[33 + ch] * (-3.f * yy + 4.f * zz - xx) + bSH_C3[3] * sh[36 + ch] * -3.f * 2.f * yz +
                                       bSH_C3[4] * sh[39 + ch] * -2.f * xy + bSH_C3[5] * sh[42 + ch] * -2.f * yz +
                                       bSH_C3[6] * sh[45 + ch] * -3.f * 2.f * xy);
                        dRGBdz[ch] += (bSH_C3[1] * sh[30 + ch] * xy + bSH_C3[2] * sh[33 + ch] * 4.f * 2.f * yz +
                                       bSH_C3[3] * sh[36 + ch] * 3.f * (2.f * zz - xx - yy) + bSH_C3[4] * sh[39 + ch] * 4.f * 2.f * xz +
                                       bSH_C3[5] * sh[42 + ch] * (xx - yy));
                    }
                }
            }
        }
        // coefficients above the active degree keep the reference's zero initialisation (rasterize_points.cu:204)
        const int used = (D + 1) * (D + 1);
        for (int k = used; k < M; k++) dsh[3 * k] = dsh[3 * k + 1] = dsh[3 * k + 2] = 0.f;
        const float ddx = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
        const float ddy = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
        const float ddz = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        mean_g[0] += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
        mean_g[1] += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
        mean_g[2] += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
    }
    // cov3D backward, backward.cu:426-487 (no quaternion-norm Jacobian, B1; ADDS onto the depth-hit rotation grads)
    {
        const real dS[3][3] = {{dcv[0], RL(0.5f) * dcv[1], RL(0.5f) * dcv[2]},
                               {RL(0.5f) * dcv[1], dcv[3], RL(0.5f) * dcv[4]},
                               {RL(0.5f) * dcv[2], RL(0.5f) * dcv[4], dcv[5]}};
        real dM[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                dM[k][j] = RL(2) * (s[k] * Rm[0][k] * dS[0][j] + s[k] * Rm[1][k] * dS[1][j] + s[k] * Rm[2][k] * dS[2][j]);
#pragma unroll
        for (int k = 0; k < 3; k++) dsc[k] = (float)(Rm[0][k] * dM[k][0] + Rm[1][k] * dM[k][1] + Rm[2][k] * dM[k][2]);
        real Mt[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 3; j++) Mt[k][j] = dM[k][j] * s[k];
        const real c2 = RL(2), c4 = RL(4);
        rot_g[0] += (float)(c2 * z * (Mt[0][1] - Mt[1][0]) + c2 * y * (Mt[2][0] - Mt[0][2]) + c2 * x * (Mt[1][2] - Mt[2][1]));
        rot_g[1] += (float)(c2 * y * (Mt[1][0] + Mt[0][1]) + c2 * z * (Mt[2][0] + Mt[0][2]) + c2 * r * (Mt[1][2] - Mt[2][1]) - c4 * x * (Mt[2][2] + Mt[1][1]));
        rot_g[2] += (float)(c2 * x * (Mt[1][0] + Mt[0][1]) + c2 * r * (Mt[2][0] - Mt[0][2]) + c2 * z * (Mt[1][2] + Mt[2][1]) - c4 * y * (Mt[2][2] + Mt[0][0]));
        rot_g[3] += (float)(c2 * r * (Mt[0][1] - Mt[1][0]) + c2 * x * (Mt[2][0] + Mt[0][2]) + c2 * y * (Mt[1][2] + Mt[2][1]) - c4 * z * (Mt[1][1] + Mt[0][0]));
    }
    dm[0] = mean_g[0], dm[1] = mean_g[1], dm[2] = mean_g[2];
    drot[0] = rot_g[0], drot[1] = rot_g[1], drot[2] = rot_g[2], drot[3] = rot_g[3];
#undef RL
}

}  // namespace

int dqo_launch_blend_backward(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin, int T,
                              const float* scales, const float* rotations, const float* dL_dcolor, const float* dL_ddepth,
                              DqoGradRec* recs, uint8_t* valid, int64_t capacity, const DqoTapDev& tap, hipStream_t s);

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
    int rc = dqo_launch_blend_backward(v, g, img, bin, T, in->scales, in->rotations, dL_dcolor, dL_ddepth, recs, valid, cap,
                                       dqo_tap_dev(ctx->loss_tap), s);
    if (rc) return rc;
    DqoGradRec* sums = reinterpret_cast<DqoGradRec*>(g.grad_sum);  // [P], lives in the forward's geometry buffer
    DQO_LAUNCH("record_sum_kernel", record_sum_kernel, dim3(dqo_spread_blocks(p->P)), dim3(256), s, p->P, g, reinterpret_cast<const float4*>(recs),
               reinterpret_cast<const uint32_t*>(valid), reinterpret_cast<float4*>(sums), cap);
    DQO_LAUNCH("gaussian_backward_kernel", gaussian_backward_kernel, dim3((p->P + 255) / 256), dim3(256), s, v, g, in->means3D, in->scales,
               in->rotations, in->shs, sums, cap, *gr);
    return DQO_OK;
}
