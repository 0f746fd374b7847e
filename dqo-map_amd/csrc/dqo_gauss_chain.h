// Per-Gaussian backward chain (K8 + K9) as one device function, shared by gaussian_backward_kernel (rast_backward.hip: the drop-in
// backward, gradient rows to HBM) and gaussian_tail_kernel (map_fused_tail.hip: the fused mapping iteration, gradient rows stay in
// LDS and feed Adam) — ONE statement sequence, so both produce the same bits.
//
// Restates /root/reference/submodules/diff-gaussian-rasterizer-depth/cuda_rasterizer/
//   backward.cu:273-422   computeCov2DCUDA
//   backward.cu:492-548   preprocessCUDA (+ SH backward :152-268, cov3D backward :426-487)
//   backward.cu:997-1065  the per-Gaussian factors of the depth-hit gradient (+ propagateRotationGrad :100-148)
// in the reference's statement order with separate IEEE multiplies and adds (no FMA contraction): the cov2D-inverse -> cov3D ->
// (scale, quaternion) chain is ill-conditioned for thin surfels (denom^2, b^2 by cancellation), so its result depends on where the
// roundings fall; evaluated like this it rounds where the oracle (and an uncontracted build of the reference) rounds — with ONE
// exception since round 6: the ten statements of the 2x2 inverse's derivative run in double (see there), which is where those
// cancellations sit.
#pragma once
#include "dqo_common.h"

// d(SH colour)/d(view direction) of one Gaussian, before the clamp mask and the incoming gradient are applied: out[0..2] = dRGBdx,
// out[3..5] = dRGBdy, out[6..8] = dRGBdz (per channel) — the sums of backward.cu:168-258 in the reference's statement order.  They
// depend on the SH coefficients and the direction only, not on any gradient, so the FORWARD's preprocess_kernel — which holds the 48
// coefficients and the direction anyway — evaluates them and leaves 9 floats per visible Gaussian (DqoGeomLayout::drgb_dir); the
// backward chain reads those instead of the 48-float SH row (45 registers less across its whole length, 192 B less to gather per
// Gaussian).  Same statements, same operands, contraction off on both sides: the later dot products of backward.cu:259-261 round where
// they rounded when the chain evaluated this itself.
__device__ __forceinline__ void dqo_sh_dir_grad(const int D, const float* sh, const float dx, const float dy, const float dz, float (&out)[9]) {
#pragma clang fp contract(off)
    constexpr float bSH_C1 = 0.4886025119029199f;
    constexpr float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                 0.5462742152960396f};
    constexpr float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                 -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};
    float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
    if (D > 0) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            dRGBdx[ch] = -bSH_C1 * sh[9 + ch];
            dRGBdy[ch] = -bSH_C1 * sh[3 + ch];
            dRGBdz[ch] = bSH_C1 * sh[6 + ch];
        }
        if (D > 1) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] += bSH_C2[0] * dy * sh[12 + ch] + bSH_C2[2] * 2.f * -dx * sh[18 + ch] + bSH_C2[3] * dz * sh[21 + ch] +
                              bSH_C2[4] * 2.f * dx * sh[24 + ch];
                dRGBdy[ch] += bSH_C2[0] * dx * sh[12 + ch] + bSH_C2[1] * dz * sh[15 + ch] + bSH_C2[2] * 2.f * -dy * sh[18 + ch] +
                              bSH_C2[4] * 2.f * -dy * sh[24 + ch];
                dRGBdz[ch] += bSH_C2[1] * dy * sh[15 + ch] + bSH_C2[2] * 2.f * 2.f * dz * sh[18 + ch] + bSH_C2[3] * dx * sh[21 + ch];
            }
            if (D > 2) {
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
#pragma unroll
    for (int ch = 0; ch < 3; ch++) out[ch] = dRGBdx[ch], out[3 + ch] = dRGBdy[ch], out[6 + ch] = dRGBdz[ch];
}

// what the chain needs of one visible Gaussian: its summed gradient record and its parameters / forward tables
struct DqoChainIn {
    float a[16];           // DqoGradRec, summed over the Gaussian's instances
    float4 cop;            // (conic, opacity) of the forward
    float mx, my, mz;      // mean
    float sx, sy, sz;      // activated scales
    float4 qt;             // activated rotation (r, x, y, z)
    float dd[9];           // dqo_sh_dir_grad of the forward (DqoGeomLayout::drgb_dir): dRGBdx[3], dRGBdy[3], dRGBdz[3]
    float4 n_np, pc;       // surfel normal in camera space (+ n . p_c), camera-space point (+ max raw scale)
    uint32_t cl;           // SH colour clamp bits
};

// gradient row of the Gaussian w.r.t. the ACTIVATED parameters.  dL/dsh[k][c] = w[k] * dRGB[c] (backward.cu:152-268 writes exactly
// this product per coefficient): kept factored — 19 floats instead of 48.
struct DqoChainOut {
    float mean_g[3];
    float rot_g[4];
    float dsc[3];
    float dop;
    float dcolr[3];        // dL/d(precomputed colour)
    float g2x, g2y;        // dL/d(2D mean)
    float dcv[6];          // dL/d(cov3D)
    float dRGB[3];         // dL/dcolour with the clamp mask applied
    float w[16];           // SH basis weights of the view direction (0 above the active degree)
};

// with_sh (wave-uniform): SH colours (the SH backward runs) or precomputed colours (it does not: w = dRGB = 0).
// DEG: the active SH degree when the caller knows it at compile time (no per-degree branches), -1 = v.D.
template <int DEG = -1>
__device__ __forceinline__ void dqo_gauss_chain(const DqoView& v, const float (&view)[16], const float (&proj)[16], const DqoChainIn& in,
                                                const bool with_sh, DqoChainOut& o) {
#pragma clang fp contract(off)
    constexpr float bSH_C0 = 0.28209479177387814f;
    constexpr float bSH_C1 = 0.4886025119029199f;
    constexpr float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                 0.5462742152960396f};
    constexpr float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                 -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};
    const float* a = in.a;
    const float4 cop = in.cop, qt = in.qt, n_np = in.n_np, pc = in.pc;
    const float mx = in.mx, my = in.my, mz = in.mz, sx = in.sx, sy = in.sy, sz = in.sz;
    const int D = DEG >= 0 ? DEG : v.D;
    const float dcolr[3] = {a[0], a[1], a[2]};
    // pixel moments of q = G * dL/dalpha summed by the blend kernel (DqoGradRec) -> gradients w.r.t. the 2D mean, the conic and
    // the opacity (backward.cu:964-994): the per-Gaussian constants are applied here, once
    const float g2x = -cop.w * (cop.x * a[3] + cop.y * a[4]) * (0.5f * v.W);
    const float g2y = -cop.w * (cop.z * a[4] + cop.y * a[3]) * (0.5f * v.H);
    const float dcx = -0.5f * cop.w * a[5], dcy = -0.5f * cop.w * a[6], dcz = -0.5f * cop.w * a[7];
    float mean_g[3] = {0.f, 0.f, 0.f};
    float rot_g[4] = {0.f, 0.f, 0.f, 0.f};
    o.dop = a[8];
    o.dcolr[0] = dcolr[0], o.dcolr[1] = dcolr[1], o.dcolr[2] = dcolr[2];
    o.g2x = g2x, o.g2y = g2y;

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
    //   dL/dmean3D = hit1 * V^T n_c + hit0 * V^T e_z,   dL/dn_c = hit[2..4] (cancelled per pixel, as the reference does),
    //   dL/dq = (dn_w/dq)^T V^T dL/dn_c
    if (a[9] != 0.f || a[10] != 0.f || a[11] != 0.f || a[12] != 0.f || a[13] != 0.f) {
        const real h0 = a[9], h1 = a[10], h2x = a[11], h2y = a[12], h2z = a[13];
        const real nx = n_np.x, ny = n_np.y, nz = n_np.z;
        mean_g[0] = (float)(h1 * (nx * view[0] + ny * view[1] + nz * view[2]) + h0 * view[2]);
        mean_g[1] = (float)(h1 * (nx * view[4] + ny * view[5] + nz * view[6]) + h0 * view[6]);
        mean_g[2] = (float)(h1 * (nx * view[8] + ny * view[9] + nz * view[10]) + h0 * view[10]);
        const real n1c = h2x, n2c = h2y, n3c = h2z;
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
        int oi = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i; j < 3; j++) c3[oi++] = Mm[0][i] * Mm[0][j] + Mm[1][i] * Mm[1][j] + Mm[2][i] * Mm[2][j];
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
    // The derivative of the 2x2 inverse (backward.cu:331-340) in DOUBLE, from the float ca, cb, cc: denom = ac - b^2, (denom - ac) = -b^2
    // and the three-term sums below cancel four to five digits for a thin elongated splat (ac / denom ~ 10^3), and in float that is
    // where the chain's result is decided — two float evaluations whose inputs differ in the last bit land 1e-3 of the tensor's
    // largest gradient apart (the reference against itself: its atomics change its sums from run to run, B10; rounds 2-5 of this
    // library against the fp32 oracle: rows up to 5x further from the fp64 derivative than the oracle).  ~35 double operations per
    // VISIBLE GAUSSIAN, three values in and three out; measured (round 6, profiles/r06_chain_precision.txt): rotation rows of the
    // smoke case 1.6e-3 -> 2.0e-5 from fp64, the summed error 0.36 .. 0.86 of the fp32 oracle's own; every other section of the
    // chain in double as well changes nothing.  What comes out is not the reference's float rounding of these ten statements — it is
    // inside the reference's own run-to-run spread and closer to the exact derivative of the same formulas.
    real dL_da = 0, dL_db = 0, dL_dc = 0;
    real dcv[6];
    bool inv_ok;
    {
        const double ca_ = (double)ca, cb_ = (double)cb, cc_ = (double)cc;
        const double dn = ca_ * cc_ - cb_ * cb_;
        const double d2i = 1.0 / ((dn * dn) + (double)0.0000001f);
        inv_ok = d2i != 0.0;
        if (inv_ok) {
            const double dx_ = (double)dcx, dy_ = (double)dcy, dz_ = (double)dcz;
            dL_da = (real)(d2i * (-cc_ * cc_ * dx_ + 2.0 * cb_ * cc_ * dy_ + (dn - ca_ * cc_) * dz_));
            dL_dc = (real)(d2i * (-ca_ * ca_ * dz_ + 2.0 * ca_ * cb_ * dy_ + (dn - ca_ * cc_) * dx_));
            dL_db = (real)(d2i * 2.0 * (cb_ * cc_ * dx_ - (dn + 2.0 * cb_ * cb_) * dy_ + ca_ * cb_ * dz_));
        }
    }
    if (inv_ok) {
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
#pragma unroll
    for (int i = 0; i < 6; i++) o.dcv[i] = (float)dcv[i];
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

#pragma unroll
    for (int k = 0; k < 16; k++) o.w[k] = 0.f;
    o.dRGB[0] = o.dRGB[1] = o.dRGB[2] = 0.f;
    if (with_sh) {
        // SH backward, backward.cu:152-268
        const uint32_t cl = in.cl;
        const float dox = mx - v.campos[0], doy = my - v.campos[1], doz = mz - v.campos[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float dx = dox / len, dy = doy / len, dz = doz / len;
        const float dRGB[3] = {dcolr[0] * ((cl & 1u) ? 0.f : 1.f), dcolr[1] * ((cl & 2u) ? 0.f : 1.f), dcolr[2] * ((cl & 4u) ? 0.f : 1.f)};
        o.dRGB[0] = dRGB[0], o.dRGB[1] = dRGB[1], o.dRGB[2] = dRGB[2];
        // d(colour)/d(direction): evaluated by the forward (dqo_sh_dir_grad above), 9 floats
        const float dRGBdx[3] = {in.dd[0], in.dd[1], in.dd[2]}, dRGBdy[3] = {in.dd[3], in.dd[4], in.dd[5]}, dRGBdz[3] = {in.dd[6], in.dd[7], in.dd[8]};
        // dL/dsh[k][c] = w * dRGB[c]: the caller forms the product (one IEEE multiply, as the reference's per-coefficient statement)
#define DQO_SETD(k, wv) o.w[k] = (wv)
        const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
        DQO_SETD(0, bSH_C0);
        if (D > 0) {
            DQO_SETD(1, -bSH_C1 * dy);
            DQO_SETD(2, bSH_C1 * dz);
            DQO_SETD(3, -bSH_C1 * dx);
            if (D > 1) {
                DQO_SETD(4, bSH_C2[0] * xy);
                DQO_SETD(5, bSH_C2[1] * yz);
                DQO_SETD(6, bSH_C2[2] * (2.f * zz - xx - yy));
                DQO_SETD(7, bSH_C2[3] * xz);
                DQO_SETD(8, bSH_C2[4] * (xx - yy));
                if (D > 2) {
                    DQO_SETD(9, bSH_C3[0] * dy * (3.f * xx - yy));
                    DQO_SETD(10, bSH_C3[1] * xy * dz);
                    DQO_SETD(11, bSH_C3[2] * dy * (4.f * zz - xx - yy));
                    DQO_SETD(12, bSH_C3[3] * dz * (2.f * zz - 3.f * xx - 3.f * yy));
                    DQO_SETD(13, bSH_C3[4] * dx * (4.f * zz - xx - yy));
                    DQO_SETD(14, bSH_C3[5] * dz * (xx - yy));
                    DQO_SETD(15, bSH_C3[6] * dx * (xx - 3.f * yy));
                }
            }
        }
#undef DQO_SETD
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
        for (int k = 0; k < 3; k++) o.dsc[k] = (float)(Rm[0][k] * dM[k][0] + Rm[1][k] * dM[k][1] + Rm[2][k] * dM[k][2]);
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
#undef RL
    o.mean_g[0] = mean_g[0], o.mean_g[1] = mean_g[1], o.mean_g[2] = mean_g[2];
    o.rot_g[0] = rot_g[0], o.rot_g[1] = rot_g[1], o.rot_g[2] = rot_g[2], o.rot_g[3] = rot_g[3];
}
