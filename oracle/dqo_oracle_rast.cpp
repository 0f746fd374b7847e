// ============================================================================
// ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not a fallback.
//
// CPU restatement (single thread, readable, scalar) of the reference's depth-aware
// differentiable 3D-Gaussian rasteriser.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library, and only as the checker / the
// timed CPU baseline.  The product path (dqo-map_amd/) never links or calls it.
//
// PARITY STATUS: *parity unpinned by reference tests*.  The reference ships no tests,
// golden vectors or fixtures for the rasteriser, and its CUDA sources cannot be
// compiled or run in the authoring container (no nvcc, no NVIDIA GPU).  This file is
// pinned instead by (a) analytic known-answer tests, (b) fp64 finite differences of
// its own forward (the code is templated on the scalar type for that purpose) and
// (c) cross-checks against the reference's importable pure-python helpers
// (utils/sh_utils.py eval_sh, utils/graphics_utils.py) — see tests/test_oracle_*.py.
//
// Every function cites the reference file:line it restates (paths relative to
// /root/reference/submodules/diff-gaussian-rasterizer-depth/).
// Mixed float/double sub-expressions of the CUDA code (double literals such as 0.5,
// 1e-8, 1.3) are reproduced literally when R = float.
// ============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

// OpenMP build (oracle/_build/libdqo_oracle_omp.so, `make omp`): the same statements, with the loops over Gaussians / tiles
// shared between the host cores — tiles are independent in K6 / K7 except for the per-Gaussian sums, which are double
// accumulators here (order-insensitive to ~1e-16; the reference's float atomicAdd order is not deterministic either, B10) and
// become `omp atomic` updates.  The serial build stays the parity reference of the small tests; the OpenMP build serves the
// full-size GPU tests and bench.py's cpu_baseline.
#ifdef _OPENMP
#include <omp.h>
#define ORC_PARALLEL_FOR _Pragma("omp parallel for schedule(dynamic, 64)")
#define ORC_PARALLEL_FOR_BIG _Pragma("omp parallel for schedule(dynamic, 2048)")
#define ORC_ATOMIC _Pragma("omp atomic")
#else
#define ORC_PARALLEL_FOR
#define ORC_PARALLEL_FOR_BIG
#define ORC_ATOMIC
#endif

namespace {

constexpr int BLOCK_X = 16;  // cuda_rasterizer/config.h:15-16
constexpr int BLOCK_Y = 16;

// cuda_rasterizer/auxiliary.h:22-37
constexpr double SH_C0 = 0.28209479177387814;
constexpr double SH_C1 = 0.4886025119029199;
constexpr double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                             -1.0925484305920792, 0.5462742152960396};
constexpr double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                             0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                             -0.5900435899266435};

template <typename R>
struct V3 {
    R x, y, z;
};

template <typename R>
struct RastCtx {
    // scalar settings
    int P = 0, D = 0, M = 0, W = 0, H = 0;
    R tanfovx, tanfovy, cx, cy, scale_mod, color_sigma, opaque_thr, depth_thr, normal_thr, T_thr;
    R focal_x, focal_y;
    R bg[3];
    R view[16], proj[16], campos[3];
    int gx = 0, gy = 0;  // tile grid
    bool has_sh = false, has_scales = false;
    // borrowed copies of the inputs needed by backward
    std::vector<R> means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp;
    std::vector<int32_t> tile_mask;
    // Object gate (NOT a reference feature: SURVEY.md §8e's per-object render, test infrastructure for the sharded job).  Empty = off =
    // the reference's semantics.  On: a list entry acts on a pixel only if the Gaussian's object id equals the pixel's owner id (a
    // negative owner: no entry acts), i.e. every pixel sees the render of its own object alone; list positions (`contributor`) are
    // counted as before.
    std::vector<int32_t> gauss_obj, pix_obj;
    // GeometryState (rasterizer_impl.h:30-44)
    std::vector<R> depths, means2D, cov3D, conic_opacity, rgb;
    std::vector<uint8_t> clamped;
    std::vector<int32_t> radii;
    std::vector<uint32_t> tiles_touched, point_offsets;
    // BinningState (rasterizer_impl.h:57-66)
    std::vector<uint32_t> point_list;
    std::vector<uint64_t> point_keys;  // (tile<<32 | rank) only meaningful for R=float
    std::vector<uint32_t> point_tile;
    // ImageState (rasterizer_impl.h:46-55)
    std::vector<uint32_t> ranges;  // [T][2]
    std::vector<R> final_T, weight_sum, hit_normal_c, hit_point_c;
    std::vector<uint32_t> n_blend;  // workload statistic (not a reference quantity): entries blended into each pixel
    // workload statistic, collected when ip[5] != 0: per list position a 256-bit mask of the tile's pixels (bit ty * 16 + tx) that
    // saw the entry with alpha >= 1/255 before they finished — the (pixel, entry) pairs a blend kernel has arithmetic for
    std::vector<uint64_t> pair_mask;
    std::vector<uint32_t> n_contrib;
    std::vector<int32_t> tile_indices;  // active tiles, row-major
    std::vector<int32_t> hit_depth_id;  // copy of out_hit_depth for backward
    int num_rendered = 0;
};

// auxiliary.h:59-97
template <typename R>
inline V3<R> transformPoint4x3(const V3<R>& p, const R* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
template <typename R>
inline void transformPoint4x4(const V3<R>& p, const R* m, R out[4]) {
    out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
template <typename R>
inline V3<R> transformVec4x3(const V3<R>& p, const R* m) {
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
            m[2] * p.x + m[6] * p.y + m[10] * p.z};
}
template <typename R>
inline V3<R> transformVec4x3Transpose(const V3<R>& p, const R* m) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
}

// auxiliary.h:139-165 (in_frustum).  The comparisons against 1.3 are double comparisons.
template <typename R>
inline bool in_frustum(const R* orig_points, int idx, const R* view, const R* proj, V3<R>& p_view) {
    V3<R> p{orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
    R hom[4];
    transformPoint4x4(p, proj, hom);
    R p_w = R(1) / (hom[3] + R(0.0000001f));
    R px = hom[0] * p_w, py = hom[1] * p_w;
    p_view = transformPoint4x3(p, view);
    if (p_view.z <= R(0.2f) || (double)px < -1.3 || (double)px > 1.3 || (double)py < -1.3 || (double)py > 1.3)
        return false;
    return true;
}

// auxiliary.h:44-47: `v * S * 0.5 + cx` — (float*int) -> float, then double arithmetic, rounded to float.
template <typename R>
inline R ndc2Pix(R v, int S, R c) {
    R vs = v * (R)S;
    return (R)((double)vs * 0.5 + (double)c);
}

// auxiliary.h:49-57
template <typename R>
inline void getRect(R px, R py, int max_radius, int gx, int gy, int rmin[2], int rmax[2]) {
    rmin[0] = std::min(gx, std::max(0, (int)((px - (R)max_radius) / (R)BLOCK_X)));
    rmin[1] = std::min(gy, std::max(0, (int)((py - (R)max_radius) / (R)BLOCK_Y)));
    rmax[0] = std::min(gx, std::max(0, (int)((px + (R)max_radius + (R)(BLOCK_X - 1)) / (R)BLOCK_X)));
    rmax[1] = std::min(gy, std::max(0, (int)((py + (R)max_radius + (R)(BLOCK_Y - 1)) / (R)BLOCK_Y)));
}

// Rotation matrix of an (un-normalised, B1) quaternion q=(r,x,y,z); Rm[i][j] is the usual R(q).
// forward.cu:57-67, 211-221 (glm literals are column-major, i.e. the transpose of how they read).
template <typename R>
inline void quatToR(const R* q, R Rm[3][3]) {
    R r = q[0], x = q[1], y = q[2], z = q[3];
    Rm[0][0] = R(1) - R(2) * (y * y + z * z);
    Rm[0][1] = R(2) * (x * y - r * z);
    Rm[0][2] = R(2) * (x * z + r * y);
    Rm[1][0] = R(2) * (x * y + r * z);
    Rm[1][1] = R(1) - R(2) * (x * x + z * z);
    Rm[1][2] = R(2) * (y * z - r * x);
    Rm[2][0] = R(2) * (x * z - r * y);
    Rm[2][1] = R(2) * (y * z + r * x);
    Rm[2][2] = R(1) - R(2) * (x * x + y * y);
}

// forward.cu:20-52
template <typename R>
inline int argMax3(R a, R b, R c) {
    if (a >= b && a >= c) return 0;
    if (b >= a && b >= c) return 1;
    return 2;
}
template <typename R>
inline int argMin3(R a, R b, R c) {
    if (a <= b && a <= c) return 0;
    if (b <= a && b <= c) return 1;
    return 2;
}

// forward.cu:202-235 (computeCov3D): Sigma = R diag(mod*s)^2 R^T, upper triangle.
template <typename R>
inline void computeCov3D(const R* scale, R mod, const R* rot, R* cov3D) {
    R Rm[3][3];
    quatToR(rot, Rm);
    R s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    // M = S * R_glm with R_glm = R(q)^T  =>  M[k][i] = s_k * Rm[i][k]
    R Mm[3][3];
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++) Mm[k][i] = s[k] * Rm[i][k];
    auto S = [&](int i, int j) { return Mm[0][i] * Mm[0][j] + Mm[1][i] * Mm[1][j] + Mm[2][i] * Mm[2][j]; };
    cov3D[0] = S(0, 0);
    cov3D[1] = S(0, 1);
    cov3D[2] = S(0, 2);
    cov3D[3] = S(1, 1);
    cov3D[4] = S(1, 2);
    cov3D[5] = S(2, 2);
}

// Shared by forward (forward.cu:158-197) and backward (backward.cu:294-327):
// t with the 1.3*tanfov clamp, A = J*W (2x3), cov2D = A Sigma A^T (+0.3 on the diagonal).
template <typename R>
struct Cov2DInter {
    V3<R> t;
    R txtz, tytz, limx, limy;
    R A[2][3];
    R a, b, c;  // dilated cov2D
};
template <typename R>
inline Cov2DInter<R> cov2DInter(const V3<R>& mean, R fx, R fy, R tan_fovx, R tan_fovy, const R* cov3D, const R* view) {
    Cov2DInter<R> o;
    o.t = transformPoint4x3(mean, view);
    o.limx = R(1.3f) * tan_fovx;
    o.limy = R(1.3f) * tan_fovy;
    o.txtz = o.t.x / o.t.z;
    o.tytz = o.t.y / o.t.z;
    o.t.x = std::min(o.limx, std::max(-o.limx, o.txtz)) * o.t.z;
    o.t.y = std::min(o.limy, std::max(-o.limy, o.tytz)) * o.t.z;
    const R J00 = fx / o.t.z, J02 = -(fx * o.t.x) / (o.t.z * o.t.z);
    const R J11 = fy / o.t.z, J12 = -(fy * o.t.y) / (o.t.z * o.t.z);
    // Rv[i][j] = view[j*4 + i]   (p_view = Rv p + t)
    for (int j = 0; j < 3; j++) {
        o.A[0][j] = J00 * view[j * 4 + 0] + J02 * view[j * 4 + 2];
        o.A[1][j] = J11 * view[j * 4 + 1] + J12 * view[j * 4 + 2];
    }
    const R V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    R VA0[3], VA1[3];
    for (int i = 0; i < 3; i++) {
        VA0[i] = V[i][0] * o.A[0][0] + V[i][1] * o.A[0][1] + V[i][2] * o.A[0][2];
        VA1[i] = V[i][0] * o.A[1][0] + V[i][1] * o.A[1][1] + V[i][2] * o.A[1][2];
    }
    o.a = o.A[0][0] * VA0[0] + o.A[0][1] * VA0[1] + o.A[0][2] * VA0[2] + R(0.3f);
    o.b = o.A[0][0] * VA1[0] + o.A[0][1] * VA1[1] + o.A[0][2] * VA1[2];
    o.c = o.A[1][0] * VA1[0] + o.A[1][1] * VA1[1] + o.A[1][2] * VA1[2] + R(0.3f);
    return o;
}

// forward.cu:104-155 (computeColorFromSH); coefficient layout [P][M][3].
template <typename R>
inline void colorFromSH(int idx, int deg, int max_coeffs, const R* means, const R* campos, const R* shs, uint8_t* clamped,
                        R out[3]) {
    R dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    R len = std::sqrt(dx * dx + dy * dy + dz * dz);
    R x = dx / len, y = dy / len, z = dz / len;
    const R* sh = shs + (size_t)idx * max_coeffs * 3;
    for (int ch = 0; ch < 3; ch++) {
        auto S = [&](int k) { return sh[3 * k + ch]; };
        R result = R(SH_C0) * S(0);
        if (deg > 0) {
            result = result - R(SH_C1) * y * S(1) + R(SH_C1) * z * S(2) - R(SH_C1) * x * S(3);
            if (deg > 1) {
                R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                result = result + R(SH_C2[0]) * xy * S(4) + R(SH_C2[1]) * yz * S(5) +
                         R(SH_C2[2]) * (R(2) * zz - xx - yy) * S(6) + R(SH_C2[3]) * xz * S(7) + R(SH_C2[4]) * (xx - yy) * S(8);
                if (deg > 2) {
                    result = result + R(SH_C3[0]) * y * (R(3) * xx - yy) * S(9) + R(SH_C3[1]) * xy * z * S(10) +
                             R(SH_C3[2]) * y * (R(4) * zz - xx - yy) * S(11) +
                             R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy) * S(12) +
                             R(SH_C3[4]) * x * (R(4) * zz - xx - yy) * S(13) + R(SH_C3[5]) * z * (xx - yy) * S(14) +
                             R(SH_C3[6]) * x * (xx - R(3) * yy) * S(15);
                }
            }
        }
        result += R(0.5f);
        clamped[3 * idx + ch] = (result < 0);
        out[ch] = std::max(result, R(0));
    }
}

// forward.cu:92-100 (ndc2ray)
template <typename R>
inline V3<R> ndc2ray(uint32_t px, uint32_t py, R fx, R fy, R cx, R cy) {
    V3<R> ray{((R)px - cx) / fx, ((R)py - cy) / fy, R(1)};
    R n = R(1) / std::sqrt(ray.x * ray.x + ray.y * ray.y + ray.z * ray.z);
    ray.x *= n;
    ray.y *= n;
    ray.z *= n;
    return ray;
}

// ---------------------------------------------------------------------------------------------
// Forward: rasterizer_impl.cu:205-441 (host sequence), forward.cu:238-354 (K1), rasterizer_impl.cu:70-142
// (K3,K5), :348-365 (H2 active tile list), forward.cu:636-866 (K6 renderCUDA_withMask).
// ---------------------------------------------------------------------------------------------
template <typename R>
RastCtx<R>* rast_forward(const int* ip, const double* fp, const R* bg, const R* means3D, const R* shs,
                         const R* colors_precomp, const R* opacities, const R* scales, const R* rotations,
                         const R* cov3D_precomp, const R* view, const R* proj, const R* campos, const int32_t* tile_mask,
                         R* out_color, R* out_depth, int32_t* out_hit_color, int32_t* out_hit_depth, R* out_hit_color_w,
                         R* out_hit_depth_w, R* out_T, int32_t* n_touched, int32_t* radii_out, const int32_t* gauss_obj = nullptr,
                         const int32_t* pix_obj = nullptr) {
    auto* c = new RastCtx<R>();
    const int P = c->P = ip[0];
    const int D = c->D = ip[1];
    const int M = c->M = ip[2];
    const int W = c->W = ip[3];
    const int H = c->H = ip[4];
    c->tanfovx = (R)fp[0];
    c->tanfovy = (R)fp[1];
    c->cx = (R)fp[2];
    c->cy = (R)fp[3];
    c->scale_mod = (R)fp[4];
    c->color_sigma = (R)fp[5];
    c->opaque_thr = (R)fp[6];
    c->depth_thr = (R)fp[7];
    c->normal_thr = (R)fp[8];
    c->T_thr = (R)fp[9];
    // rasterizer_impl.cu:245-246
    c->focal_y = (R)H / (R(2) * c->tanfovy);
    c->focal_x = (R)W / (R(2) * c->tanfovx);
    const R fx = c->focal_x, fy = c->focal_y, cx = c->cx, cy = c->cy;
    for (int i = 0; i < 3; i++) c->bg[i] = bg[i];
    for (int i = 0; i < 16; i++) c->view[i] = view[i], c->proj[i] = proj[i];
    for (int i = 0; i < 3; i++) c->campos[i] = campos[i];
    const int gx = c->gx = (W + BLOCK_X - 1) / BLOCK_X;
    const int gy = c->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int T = gx * gy;
    c->has_sh = (colors_precomp == nullptr);
    c->has_scales = (cov3D_precomp == nullptr);
    c->means3D.assign(means3D, means3D + 3 * (size_t)P);
    if (c->has_sh) c->shs.assign(shs, shs + (size_t)P * M * 3);
    else c->colors_precomp.assign(colors_precomp, colors_precomp + 3 * (size_t)P);
    c->opacities.assign(opacities, opacities + P);
    if (scales) c->scales.assign(scales, scales + 3 * (size_t)P);
    if (rotations) c->rotations.assign(rotations, rotations + 4 * (size_t)P);
    if (cov3D_precomp) c->cov3D_precomp.assign(cov3D_precomp, cov3D_precomp + 6 * (size_t)P);
    c->tile_mask.assign(tile_mask, tile_mask + T);
    const bool gated = gauss_obj != nullptr && pix_obj != nullptr;
    if (gated) c->gauss_obj.assign(gauss_obj, gauss_obj + P), c->pix_obj.assign(pix_obj, pix_obj + (size_t)H * W);

    // Initial fills: rasterize_points.cu:79-89 (ids are 0, not -1; T is 1) — quirk B7.
    const size_t HW = (size_t)H * W;
    std::fill(out_color, out_color + 3 * HW, R(0));
    std::fill(out_depth, out_depth + HW, R(0));
    std::fill(out_hit_color, out_hit_color + HW, 0);
    std::fill(out_hit_depth, out_hit_depth + HW, 0);
    std::fill(out_hit_color_w, out_hit_color_w + HW, R(0));
    std::fill(out_hit_depth_w, out_hit_depth_w + HW, R(0));
    std::fill(out_T, out_T + HW, R(1));
    std::fill(n_touched, n_touched + P, 0);
    std::fill(radii_out, radii_out + P, 0);

    c->depths.assign(P, 0);
    c->means2D.assign(2 * (size_t)P, 0);
    c->cov3D.assign(6 * (size_t)P, 0);
    c->conic_opacity.assign(4 * (size_t)P, 0);
    c->rgb.assign(3 * (size_t)P, 0);
    c->clamped.assign(3 * (size_t)P, 0);
    c->radii.assign(P, 0);
    c->tiles_touched.assign(P, 0);
    c->point_offsets.assign(P, 0);
    c->ranges.assign(2 * (size_t)T, 0);
    c->final_T.assign(HW, 0);
    c->weight_sum.assign(HW, 0);
    c->n_blend.assign(HW, 0);
    c->n_contrib.assign(HW, 0);
    c->hit_normal_c.assign(3 * HW, 0);
    c->hit_point_c.assign(3 * HW, 0);
    c->hit_depth_id.assign(HW, 0);
    if (P == 0) return c;  // rasterize_points.cu:103 (B9)

    // ---- K1 preprocessCUDA, forward.cu:238-354 ----
    ORC_PARALLEL_FOR_BIG
    for (int idx = 0; idx < P; idx++) {
        V3<R> p_view;
        if (!in_frustum(means3D, idx, view, proj, p_view)) continue;
        V3<R> p_orig{means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
        R hom[4];
        transformPoint4x4(p_orig, proj, hom);
        R p_w = R(1) / (hom[3] + R(0.0000001f));
        R projx = hom[0] * p_w, projy = hom[1] * p_w;
        const R* cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * (size_t)idx;
        else {
            computeCov3D(scales + 3 * (size_t)idx, c->scale_mod, rotations + 4 * (size_t)idx, &c->cov3D[6 * (size_t)idx]);
            cov3D = &c->cov3D[6 * (size_t)idx];
        }
        Cov2DInter<R> ci = cov2DInter(p_orig, fx, fy, c->tanfovx, c->tanfovy, cov3D, view);
        R det = ci.a * ci.c - ci.b * ci.b;
        if (det == R(0)) continue;
        R det_inv = R(1) / det;
        R conic[3] = {ci.c * det_inv, -ci.b * det_inv, ci.a * det_inv};
        R mid = R(0.5f) * (ci.a + ci.c);
        R lambda1 = mid + std::sqrt(std::max(R(0.1f), mid * mid - det));
        R lambda2 = mid - std::sqrt(std::max(R(0.1f), mid * mid - det));
        R my_radius = std::ceil(c->color_sigma * std::sqrt(std::max(lambda1, lambda2)));
        R pix_x = ndc2Pix(projx, W, cx), pix_y = ndc2Pix(projy, H, cy);
        int rmin[2], rmax[2];
        getRect(pix_x, pix_y, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (c->has_sh) colorFromSH(idx, D, M, means3D, campos, shs, c->clamped.data(), &c->rgb[3 * (size_t)idx]);
        c->depths[idx] = p_view.z;
        c->radii[idx] = (int)my_radius;
        radii_out[idx] = (int)my_radius;
        c->means2D[2 * (size_t)idx] = pix_x;
        c->means2D[2 * (size_t)idx + 1] = pix_y;
        c->conic_opacity[4 * (size_t)idx + 0] = conic[0];
        c->conic_opacity[4 * (size_t)idx + 1] = conic[1];
        c->conic_opacity[4 * (size_t)idx + 2] = conic[2];
        c->conic_opacity[4 * (size_t)idx + 3] = opacities[idx];
        int touch = 0;
        for (int x = rmin[0]; x < rmax[0]; x++)
            for (int y = rmin[1]; y < rmax[1]; y++)
                if (tile_mask[y * gx + x]) touch++;
        c->tiles_touched[idx] = touch;
    }
    // ---- K2 inclusive scan, rasterizer_impl.cu:303 ----
    uint32_t run = 0;
    for (int i = 0; i < P; i++) {
        run += c->tiles_touched[i];
        c->point_offsets[i] = run;
    }
    const int N = c->num_rendered = (int)run;
    // ---- K3 duplicateWithKeys, rasterizer_impl.cu:70-115 ----
    struct Inst {
        uint32_t tile;
        R depth;
        uint32_t id;
    };
    std::vector<Inst> inst(N);
    ORC_PARALLEL_FOR_BIG
    for (int idx = 0; idx < P; idx++) {
        if (c->radii[idx] <= 0) continue;
        uint32_t off = idx == 0 ? 0 : c->point_offsets[idx - 1];
        int rmin[2], rmax[2];
        getRect(c->means2D[2 * (size_t)idx], c->means2D[2 * (size_t)idx + 1], c->radii[idx], gx, gy, rmin, rmax);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                uint32_t tile = y * gx + x;
                if (tile_mask[tile]) inst[off++] = Inst{tile, c->depths[idx], (uint32_t)idx};
            }
    }
    // ---- K4 stable radix sort on (tile, depth bits), rasterizer_impl.cu:327-336.  Depths are > 0.2 so the
    // float bit pattern orders like the value; stable => ties keep emission (= ascending id) order. ----
    std::stable_sort(inst.begin(), inst.end(), [](const Inst& a, const Inst& b) {
        if (a.tile != b.tile) return a.tile < b.tile;
        return a.depth < b.depth;
    });
    c->point_list.resize(N);
    c->point_tile.resize(N);
    for (int i = 0; i < N; i++) c->point_list[i] = inst[i].id, c->point_tile[i] = inst[i].tile;
    // ---- K5 identifyTileRanges, rasterizer_impl.cu:120-142 (after the memset to 0 at :338) ----
    for (int i = 0; i < N; i++) {
        uint32_t cur = inst[i].tile;
        if (i == 0) c->ranges[2 * cur] = 0;
        else {
            uint32_t prev = inst[i - 1].tile;
            if (cur != prev) {
                c->ranges[2 * prev + 1] = i;
                c->ranges[2 * cur] = i;
            }
        }
        if (i == N - 1) c->ranges[2 * cur + 1] = N;
    }
    // ---- H2 active tile list, rasterizer_impl.cu:348-365 ----
    for (int t = 0; t < T; t++)
        if (c->ranges[2 * t] != c->ranges[2 * t + 1]) c->tile_indices.push_back(t);

    const bool want_masks = ip[5] != 0;
    if (want_masks) c->pair_mask.assign(4 * (size_t)N, 0);
    // ---- K6 renderCUDA_withMask, forward.cu:636-866 ----
    const R* features = c->has_sh ? c->rgb.data() : colors_precomp;
    const int n_active = (int)c->tile_indices.size();
    ORC_PARALLEL_FOR
    for (int ti = 0; ti < n_active; ti++) {
        const int real_tile = c->tile_indices[ti];
        const int tile_x = real_tile % gx, tile_y = real_tile / gx;
        const uint32_t r0 = c->ranges[2 * real_tile], r1 = c->ranges[2 * real_tile + 1];
        for (int ty = 0; ty < BLOCK_Y; ty++)
            for (int tx = 0; tx < BLOCK_X; tx++) {
                const uint32_t px = tile_x * BLOCK_X + tx, py = tile_y * BLOCK_Y + ty;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;  // done = !inside
                const size_t pix_id = (size_t)W * py + px;
                const R pixfx = (R)px, pixfy = (R)py;
                const V3<R> ray = ndc2ray(px, py, fx, fy, cx, cy);
                R Tt = R(1), end_T = R(1);
                uint32_t contributor = 0, last_contributor = 0;
                R C[3] = {0, 0, 0};
                R depth_ = 0;
                bool hit_gaussian = false, done = false;
                int hit_id = -1, hit_color_id = -1;
                R color_weight_max = R(-1), hit_color_weight = 0, hit_depth_weight = 0, weight_sum = 0;
                uint32_t n_blend = 0;
                for (uint32_t k = r0; k < r1 && !done; k++) {
                    contributor++;
                    const int g = (int)c->point_list[k];
                    if (gated && (c->pix_obj[pix_id] < 0 || c->gauss_obj[g] != c->pix_obj[pix_id])) continue;  // object gate (off = reference)
                    const R dx = c->means2D[2 * (size_t)g] - pixfx, dy = c->means2D[2 * (size_t)g + 1] - pixfy;
                    const R* co = &c->conic_opacity[4 * (size_t)g];
                    const R power = R(-0.5f) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > R(0)) continue;
                    const R alpha = std::min(R(0.99f), co[3] * std::exp(power));
                    if (alpha < R(1.0f / 255.0f)) continue;
                    if (want_masks) c->pair_mask[4 * (size_t)k + (ty >> 2)] |= 1ull << ((ty & 3) * 16 + tx);
                    // forward.cu:779-791: surfel normal / ray-plane hit, recomputed per pair in the reference
                    R Rm[3][3];
                    quatToR(&rotations[4 * (size_t)g], Rm);
                    const R* sc = &scales[3 * (size_t)g];
                    const int naxis = argMin3(sc[0], sc[1], sc[2]);
                    const int maxis = argMax3(sc[0], sc[1], sc[2]);
                    const V3<R> n_w{Rm[0][naxis], Rm[1][naxis], Rm[2][naxis]};
                    const R smax = sc[maxis] * c->scale_mod;
                    const V3<R> p_w{means3D[3 * (size_t)g], means3D[3 * (size_t)g + 1], means3D[3 * (size_t)g + 2]};
                    const V3<R> n_c = transformVec4x3(n_w, view);
                    const V3<R> p_c = transformPoint4x3(p_w, view);
                    const R num = p_c.x * n_c.x + p_c.y * n_c.y + p_c.z * n_c.z;
                    const R den = ray.x * n_c.x + ray.y * n_c.y + ray.z * n_c.z;
                    const R t = (R)((double)num / ((double)den + 1e-8));
                    const V3<R> hit_point{t * ray.x, t * ray.y, t * ray.z};
                    const R angle_distance = std::abs(den);
                    const R depth_distance = std::abs(hit_point.z - p_c.z);
                    if (!hit_gaussian && alpha >= c->opaque_thr) {
                        const R opaque_depth = c->depths[g];
                        hit_id = g;
                        hit_depth_weight = alpha * Tt;
                        if (depth_distance <= smax * c->depth_thr && angle_distance >= c->normal_thr) depth_ = t * ray.z;
                        else depth_ = opaque_depth;
                        c->hit_normal_c[3 * pix_id + 0] = n_c.x;
                        c->hit_normal_c[3 * pix_id + 1] = n_c.y;
                        c->hit_normal_c[3 * pix_id + 2] = n_c.z;
                        c->hit_point_c[3 * pix_id + 0] = hit_point.x;
                        c->hit_point_c[3 * pix_id + 1] = hit_point.y;
                        c->hit_point_c[3 * pix_id + 2] = hit_point.z;
                        hit_gaussian = true;
                    }
                    const R test_T = Tt * (R(1) - alpha);
                    if (test_T < c->T_thr && hit_gaussian) {
                        done = true;
                        continue;
                    }
                    if (test_T >= c->T_thr) {
                        const R color_weight = alpha * Tt;
                        weight_sum += color_weight;
                        n_blend++;
                        for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * (size_t)g + ch] * color_weight;
                        if (color_weight > color_weight_max) {
                            color_weight_max = color_weight;
                            hit_color_id = g;
                            hit_color_weight = color_weight_max;
                        }
                        if (test_T > R(0.5f)) {  // forward.cu:833-835 (B8)
                            ORC_ATOMIC
                            n_touched[g] += 1;
                        }
                        last_contributor = contributor;
                        end_T = test_T;
                    }
                    Tt = test_T;
                }
                c->final_T[pix_id] = end_T;
                c->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + Tt * bg[ch];  // running T (B2)
                out_depth[pix_id] = depth_;
                out_hit_depth[pix_id] = hit_id;
                c->hit_depth_id[pix_id] = hit_id;
                out_hit_color[pix_id] = hit_color_id;
                out_hit_color_w[pix_id] = hit_color_weight;
                out_hit_depth_w[pix_id] = hit_depth_weight;
                c->weight_sum[pix_id] = weight_sum;
                c->n_blend[pix_id] = n_blend;
                out_T[pix_id] = end_T;
            }
    }
    return c;
}

// backward.cu:100-148 (propagateRotationGrad): d(column `axis` of R(q)) / dq, literal table.
template <typename R>
inline void rotColGrad(const R* q, int axis, R d0[3], R d1[3], R d2[3], R d3[3]) {
    const R q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    if (axis == 0) {
        d0[0] = 0, d0[1] = 2 * q3, d0[2] = -2 * q2;
        d1[0] = 0, d1[1] = 2 * q2, d1[2] = 2 * q3;
        d2[0] = -4 * q2, d2[1] = 2 * q1, d2[2] = -2 * q0;
        d3[0] = -4 * q3, d3[1] = 2 * q0, d3[2] = 2 * q1;
    } else if (axis == 1) {
        d0[0] = -2 * q3, d0[1] = 0, d0[2] = 2 * q1;
        d1[0] = 2 * q2, d1[1] = -4 * q1, d1[2] = 2 * q0;
        d2[0] = 2 * q1, d2[1] = 0, d2[2] = 2 * q3;
        d3[0] = -2 * q0, d3[1] = -4 * q3, d3[2] = 2 * q2;
    } else {
        d0[0] = 2 * q2, d0[1] = -2 * q1, d0[2] = 0;
        d1[0] = 2 * q3, d1[1] = -2 * q0, d1[2] = -4 * q1;
        d2[0] = 2 * q0, d2[1] = 2 * q3, d2[2] = -4 * q2;
        d3[0] = 2 * q1, d3[1] = 2 * q2, d3[2] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: rasterizer_impl.cu:445-564; K7 backward.cu:808-1066; K8 :273-422; K9 :492-548 (+152-268, 426-487).
// Cross-pixel sums (the reference's float atomicAdds, order-dependent: B10) are accumulated in double and
// rounded once, so the oracle is the correctly rounded value both implementations approximate.
// ---------------------------------------------------------------------------------------------
template <typename R>
void rast_backward(RastCtx<R>* c, const R* dL_dpixels, const R* dL_dpixel_depths, R* dL_dmeans3D, R* dL_dsh, R* dL_dcolors,
                   R* dL_dopacity, R* dL_dscales, R* dL_drot, R* dL_dcov3D, R* dL_dmeans2D_out, R* dL_dconic_out) {
    const int P = c->P, W = c->W, H = c->H, M = c->M, D = c->D;
    const size_t HW = (size_t)H * W;
    // rasterize_points.cu:198-206 zero-initialised gradient tensors
    std::fill(dL_dmeans3D, dL_dmeans3D + 3 * (size_t)P, R(0));
    if (dL_dsh) std::fill(dL_dsh, dL_dsh + (size_t)P * M * 3, R(0));
    std::fill(dL_dcolors, dL_dcolors + 3 * (size_t)P, R(0));
    std::fill(dL_dopacity, dL_dopacity + P, R(0));
    std::fill(dL_dscales, dL_dscales + 3 * (size_t)P, R(0));
    std::fill(dL_drot, dL_drot + 4 * (size_t)P, R(0));
    std::fill(dL_dcov3D, dL_dcov3D + 6 * (size_t)P, R(0));
    if (P == 0) return;
    std::vector<double> a_color(3 * (size_t)P, 0), a_mean2D(2 * (size_t)P, 0), a_conic(3 * (size_t)P, 0), a_opac(P, 0),
        a_mean3D(3 * (size_t)P, 0), a_rot(4 * (size_t)P, 0);
    const R fx = c->focal_x, fy = c->focal_y, cx = c->cx, cy = c->cy;
    const R* view = c->view;
    const R* colors = c->has_sh ? c->rgb.data() : c->colors_precomp.data();
    const int gx = c->gx;

    // ---- K7 renderCUDA_flat ----
    const int n_active = (int)c->tile_indices.size();
    ORC_PARALLEL_FOR
    for (int ti = 0; ti < n_active; ti++) {
        const int real_tile = c->tile_indices[ti];
        const int tile_x = real_tile % gx, tile_y = real_tile / gx;
        const uint32_t r0 = c->ranges[2 * real_tile], r1 = c->ranges[2 * real_tile + 1];
        for (int ty = 0; ty < BLOCK_Y; ty++)
            for (int tx = 0; tx < BLOCK_X; tx++) {
                const uint32_t px = tile_x * BLOCK_X + tx, py = tile_y * BLOCK_Y + ty;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const size_t pix_id = (size_t)W * py + px;
                const R pixfx = (R)px, pixfy = (R)py;
                const R T_final = c->final_T[pix_id];
                R Tt = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = c->n_contrib[pix_id];
                R accum_rec[3] = {0, 0, 0}, dL_dpixel[3], last_color[3] = {0, 0, 0};
                for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix_id];
                R last_alpha = 0;
                const R ddelx_dx = (R)(0.5 * W), ddely_dy = (R)(0.5 * H);
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    const int g = (int)c->point_list[k];
                    if (contributor >= last_contributor) continue;
                    if (!c->gauss_obj.empty() && (c->pix_obj[pix_id] < 0 || c->gauss_obj[g] != c->pix_obj[pix_id])) continue;  // object gate
                    const R dx = c->means2D[2 * (size_t)g] - pixfx, dy = c->means2D[2 * (size_t)g + 1] - pixfy;
                    const R* co = &c->conic_opacity[4 * (size_t)g];
                    const R power = R(-0.5f) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > R(0)) continue;
                    const R G = std::exp(power);
                    const R alpha = std::min(R(0.99f), co[3] * G);
                    if (alpha < R(1.0f / 255.0f)) continue;
                    Tt = Tt / (R(1) - alpha);
                    const R dchannel_dcolor = alpha * Tt;
                    R dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        const R col = colors[3 * (size_t)g + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (R(1) - last_alpha) * accum_rec[ch];
                        last_color[ch] = col;
                        const R dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (col - accum_rec[ch]) * dL_dchannel;
                        ORC_ATOMIC
                        a_color[3 * (size_t)g + ch] += (double)(dchannel_dcolor * dL_dchannel);
                    }
                    dL_dalpha *= Tt;
                    last_alpha = alpha;
                    R bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++) bg_dot_dpixel += c->bg[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (R(1) - alpha)) * bg_dot_dpixel;  // end_T, not running T (B2)
                    const R dL_dG = co[3] * dL_dalpha;
                    const R gdx = G * dx, gdy = G * dy;
                    const R dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const R dG_ddely = -gdy * co[2] - gdx * co[1];
                    ORC_ATOMIC
                    a_mean2D[2 * (size_t)g + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    ORC_ATOMIC
                    a_mean2D[2 * (size_t)g + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    ORC_ATOMIC
                    a_conic[3 * (size_t)g + 0] += (double)(R(-0.5f) * gdx * dx * dL_dG);
                    ORC_ATOMIC
                    a_conic[3 * (size_t)g + 1] += (double)(R(-0.5f) * gdx * dy * dL_dG);
                    ORC_ATOMIC
                    a_conic[3 * (size_t)g + 2] += (double)(R(-0.5f) * gdy * dy * dL_dG);
                    ORC_ATOMIC
                    a_opac[g] += (double)(G * dL_dalpha);
                }
                // hit-Gaussian depth gradient, backward.cu:997-1065
                const int hid = c->hit_depth_id[pix_id];
                if (hid >= 0) {
                    const int g = hid;
                    const V3<R> ray = ndc2ray(px, py, fx, fy, cx, cy);
                    const R* sc = &c->scales[3 * (size_t)g];
                    const R scale_max = std::max(std::max(sc[0], sc[1]), sc[2]);  // raw scales (B6)
                    const V3<R> n_c{c->hit_normal_c[3 * pix_id], c->hit_normal_c[3 * pix_id + 1], c->hit_normal_c[3 * pix_id + 2]};
                    const V3<R> p_w{c->means3D[3 * (size_t)g], c->means3D[3 * (size_t)g + 1], c->means3D[3 * (size_t)g + 2]};
                    const V3<R> p_c = transformPoint4x3(p_w, view);
                    const R hit_z = c->hit_point_c[3 * pix_id + 2];
                    const R nr_f = n_c.x * ray.x + n_c.y * ray.y + n_c.z * ray.z;
                    const R angle_distance = std::abs(nr_f);
                    const R depth_distance = std::abs(hit_z - p_c.z);
                    const R dL_ddi = dL_dpixel_depths[pix_id];
                    if (depth_distance <= c->depth_thr * scale_max && angle_distance >= c->normal_thr) {
                        const R nr = (R)((double)nr_f + 1e-8);
                        const R inv_nr = R(1) / nr, inv_nr2 = inv_nr * inv_nr;
                        const R np = n_c.x * p_c.x + n_c.y * p_c.y + n_c.z * p_c.z;
                        const R dpx = ray.z * n_c.x * inv_nr, dpy = ray.z * n_c.y * inv_nr, dpz = ray.z * n_c.z * inv_nr;
                        ORC_ATOMIC
                        a_mean3D[3 * (size_t)g + 0] += (double)(dL_ddi * (dpx * view[0] + dpy * view[1] + dpz * view[2]));
                        ORC_ATOMIC
                        a_mean3D[3 * (size_t)g + 1] += (double)(dL_ddi * (dpx * view[4] + dpy * view[5] + dpz * view[6]));
                        ORC_ATOMIC
                        a_mean3D[3 * (size_t)g + 2] += (double)(dL_ddi * (dpx * view[8] + dpy * view[9] + dpz * view[10]));
                        const int axis = argMin3(sc[0], sc[1], sc[2]);
                        const R n1c = ray.z * (nr * p_c.x - np * ray.x) * inv_nr2;
                        const R n2c = ray.z * (nr * p_c.y - np * ray.y) * inv_nr2;
                        const R n3c = ray.z * (nr * p_c.z - np * ray.z) * inv_nr2;
                        const R n1w = n1c * view[0] + n2c * view[1] + n3c * view[2];
                        const R n2w = n1c * view[4] + n2c * view[5] + n3c * view[6];
                        const R n3w = n1c * view[8] + n2c * view[9] + n3c * view[10];
                        R d0[3], d1[3], d2[3], d3[3];
                        rotColGrad(&c->rotations[4 * (size_t)g], axis, d0, d1, d2, d3);
                        ORC_ATOMIC
                        a_rot[4 * (size_t)g + 0] += (double)(dL_ddi * (n1w * d0[0] + n2w * d0[1] + n3w * d0[2]));
                        ORC_ATOMIC
                        a_rot[4 * (size_t)g + 1] += (double)(dL_ddi * (n1w * d1[0] + n2w * d1[1] + n3w * d1[2]));
                        ORC_ATOMIC
                        a_rot[4 * (size_t)g + 2] += (double)(dL_ddi * (n1w * d2[0] + n2w * d2[1] + n3w * d2[2]));
                        ORC_ATOMIC
                        a_rot[4 * (size_t)g + 3] += (double)(dL_ddi * (n1w * d3[0] + n2w * d3[1] + n3w * d3[2]));
                    } else {
                        ORC_ATOMIC
                        a_mean3D[3 * (size_t)g + 0] += (double)(dL_ddi * view[2]);
                        ORC_ATOMIC
                        a_mean3D[3 * (size_t)g + 1] += (double)(dL_ddi * view[6]);
                        ORC_ATOMIC
                        a_mean3D[3 * (size_t)g + 2] += (double)(dL_ddi * view[10]);
                    }
                }
            }
    }
    for (size_t i = 0; i < 3 * (size_t)P; i++) dL_dcolors[i] = (R)a_color[i], dL_dmeans3D[i] = (R)a_mean3D[i];
    for (size_t i = 0; i < (size_t)P; i++) dL_dopacity[i] = (R)a_opac[i];
    for (size_t i = 0; i < 4 * (size_t)P; i++) dL_drot[i] = (R)a_rot[i];
    std::vector<R> dL_dmean2D(2 * (size_t)P), dL_dconic(3 * (size_t)P);
    for (size_t i = 0; i < 2 * (size_t)P; i++) dL_dmean2D[i] = (R)a_mean2D[i];
    for (size_t i = 0; i < 3 * (size_t)P; i++) dL_dconic[i] = (R)a_conic[i];
    if (dL_dmeans2D_out) std::copy(dL_dmean2D.begin(), dL_dmean2D.end(), dL_dmeans2D_out);
    if (dL_dconic_out) std::copy(dL_dconic.begin(), dL_dconic.end(), dL_dconic_out);

    const R* cov3Ds = c->has_scales ? c->cov3D.data() : c->cov3D_precomp.data();
    ORC_PARALLEL_FOR_BIG
    for (int idx = 0; idx < P; idx++) {
        if (!(c->radii[idx] > 0)) continue;
        // ---- K8 computeCov2DCUDA, backward.cu:273-422 ----
        const R* cov3D = cov3Ds + 6 * (size_t)idx;
        const V3<R> mean{c->means3D[3 * (size_t)idx], c->means3D[3 * (size_t)idx + 1], c->means3D[3 * (size_t)idx + 2]};
        const R dcx = dL_dconic[3 * (size_t)idx], dcy = dL_dconic[3 * (size_t)idx + 1], dcz = dL_dconic[3 * (size_t)idx + 2];
        Cov2DInter<R> ci = cov2DInter(mean, fx, fy, c->tanfovx, c->tanfovy, cov3D, view);
        const R x_grad_mul = (ci.txtz < -ci.limx || ci.txtz > ci.limx) ? R(0) : R(1);
        const R y_grad_mul = (ci.tytz < -ci.limy || ci.tytz > ci.limy) ? R(0) : R(1);
        const R a = ci.a, b = ci.b, cc = ci.c;
        const R denom = a * cc - b * b;
        R dL_da = 0, dL_db = 0, dL_dc = 0;
        const R denom2inv = R(1) / ((denom * denom) + R(0.0000001f));
        const R(*A)[3] = ci.A;
        R* dcov = dL_dcov3D + 6 * (size_t)idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-cc * cc * dcx + 2 * b * cc * dcy + (denom - a * cc) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * cc) * dcx);
            dL_db = denom2inv * 2 * (b * cc * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
            dcov[0] = A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc;
            dcov[3] = A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc;
            dcov[5] = A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc;
            dcov[1] = 2 * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db + 2 * A[1][0] * A[1][1] * dL_dc;
            dcov[2] = 2 * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db + 2 * A[1][0] * A[1][2] * dL_dc;
            dcov[4] = 2 * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db + 2 * A[1][1] * A[1][2] * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        const R V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
        R dT0[3], dT1[3];
        for (int j = 0; j < 3; j++) {
            const R A0V = A[0][0] * V[j][0] + A[0][1] * V[j][1] + A[0][2] * V[j][2];
            const R A1V = A[1][0] * V[j][0] + A[1][1] * V[j][1] + A[1][2] * V[j][2];
            dT0[j] = 2 * A0V * dL_da + A1V * dL_db;
            dT1[j] = 2 * A1V * dL_dc + A0V * dL_db;
        }
        // Rv[i][j] = view[j*4+i]
        const R dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
        const R dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
        const R dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
        const R dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
        const R tz = R(1) / ci.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const R dL_dtx = x_grad_mul * -fx * tz2 * dJ02;
        const R dL_dty = y_grad_mul * -fy * tz2 * dJ12;
        const R dL_dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * ci.t.x) * tz3 * dJ02 + (2 * fy * ci.t.y) * tz3 * dJ12;
        const V3<R> dmean_cov = transformVec4x3Transpose(V3<R>{dL_dtx, dL_dty, dL_dtz}, view);
        R* dm = dL_dmeans3D + 3 * (size_t)idx;
        dm[0] += dmean_cov.x;
        dm[1] += dmean_cov.y;
        dm[2] += dmean_cov.z;

        // ---- K9 preprocessCUDA (backward), backward.cu:492-548 ----
        const R* proj = c->proj;
        R hom[4];
        transformPoint4x4(mean, proj, hom);
        const R m_w = R(1) / (hom[3] + R(0.0000001f));
        const R mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const R mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const R g2x = dL_dmean2D[2 * (size_t)idx], g2y = dL_dmean2D[2 * (size_t)idx + 1];
        dm[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dm[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dm[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

        if (c->has_sh) {
            // backward.cu:152-268 (computeColorFromSH backward)
            const R* campos = c->campos;
            const R dox = mean.x - campos[0], doy = mean.y - campos[1], doz = mean.z - campos[2];
            const R len = std::sqrt(dox * dox + doy * doy + doz * doz);
            const R x = dox / len, y = doy / len, z = doz / len;
            const R* sh = c->shs.data() + (size_t)idx * M * 3;
            R dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolors[3 * (size_t)idx + ch] * (c->clamped[3 * (size_t)idx + ch] ? R(0) : R(1));
            R* dsh = dL_dsh + (size_t)idx * M * 3;
            R dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
            auto S = [&](int k, int ch) { return sh[3 * k + ch]; };
            auto setd = [&](int k, R w) {
                for (int ch = 0; ch < 3; ch++) dsh[3 * k + ch] = w * dRGB[ch];
            };
            setd(0, R(SH_C0));
            if (D > 0) {
                setd(1, -R(SH_C1) * y);
                setd(2, R(SH_C1) * z);
                setd(3, -R(SH_C1) * x);
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] = -R(SH_C1) * S(3, ch);
                    dRGBdy[ch] = -R(SH_C1) * S(1, ch);
                    dRGBdz[ch] = R(SH_C1) * S(2, ch);
                }
                if (D > 1) {
                    const R xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    setd(4, R(SH_C2[0]) * xy);
                    setd(5, R(SH_C2[1]) * yz);
                    setd(6, R(SH_C2[2]) * (R(2) * zz - xx - yy));
                    setd(7, R(SH_C2[3]) * xz);
                    setd(8, R(SH_C2[4]) * (xx - yy));
                    for (int ch = 0; ch < 3; ch++) {
                        dRGBdx[ch] += R(SH_C2[0]) * y * S(4, ch) + R(SH_C2[2]) * R(2) * -x * S(6, ch) + R(SH_C2[3]) * z * S(7, ch) +
                                      R(SH_C2[4]) * R(2) * x * S(8, ch);
                        dRGBdy[ch] += R(SH_C2[0]) * x * S(4, ch) + R(SH_C2[1]) * z * S(5, ch) + R(SH_C2[2]) * R(2) * -y * S(6, ch) +
                                      R(SH_C2[4]) * R(2) * -y * S(8, ch);
                        dRGBdz[ch] += R(SH_C2[1]) * y * S(5, ch) + R(SH_C2[2]) * R(2) * R(2) * z * S(6, ch) + R(SH_C2[3]) * x * S(7, ch);
                    }
                    if (D > 2) {
                        setd(9, R(SH_C3[0]) * y * (R(3) * xx - yy));
                        setd(10, R(SH_C3[1]) * xy * z);
                        setd(11, R(SH_C3[2]) * y * (R(4) * zz - xx - yy));
                        setd(12, R(SH_C3[3]) * z * (R(2) * zz - R(3) * xx - R(3) * yy));
                        setd(13, R(SH_C3[4]) * x * (R(4) * zz - xx - yy));
                        setd(14, R(SH_C3[5]) * z * (xx - yy));
                        setd(15, R(SH_C3[6]) * x * (xx - R(3) * yy));
                        for (int ch = 0; ch < 3; ch++) {
                            dRGBdx[ch] += (R(SH_C3[0]) * S(9, ch) * R(3) * R(2) * xy + R(SH_C3[1]) * S(10, ch) * yz +
                                           R(SH_C3[2]) * S(11, ch) * R(-2) * xy + R(SH_C3[3]) * S(12, ch) * R(-3) * R(2) * xz +
                                           R(SH_C3[4]) * S(13, ch) * (R(-3) * xx + R(4) * zz - yy) + R(SH_C3[5]) * S(14, ch) * R(2) * xz +
                                           R(SH_C3[6]) * S(15, ch) * R(3) * (xx - yy));
                            dRGBdy[ch] += (R(SH_C3[0]) * S(9, ch) * R(3) * (xx - yy) + R(SH_C3[1]) * S(10, ch) * xz +
                                           R(SH_C3[2]) * S(11, ch) * (R(-3) * yy + R(4) * zz - xx) +
                                           R(SH_C3[3]) * S(12, ch) * R(-3) * R(2) * yz + R(SH_C3[4]) * S(13, ch) * R(-2) * xy +
                                           R(SH_C3[5]) * S(14, ch) * R(-2) * yz + R(SH_C3[6]) * S(15, ch) * R(-3) * R(2) * xy);
                            dRGBdz[ch] += (R(SH_C3[1]) * S(10, ch) * xy + R(SH_C3[2]) * S(11, ch) * R(4) * R(2) * yz +
                                           R(SH_C3[3]) * S(12, ch) * R(3) * (R(2) * zz - xx - yy) +
                                           R(SH_C3[4]) * S(13, ch) * R(4) * R(2) * xz + R(SH_C3[5]) * S(14, ch) * (xx - yy));
                        }
                    }
                }
            }
            const R ddx = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
            const R ddy = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
            const R ddz = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
            // auxiliary.h:107-117 (dnormvdv)
            const R sum2 = dox * dox + doy * doy + doz * doz;
            const R invsum32 = R(1) / std::sqrt(sum2 * sum2 * sum2);
            dm[0] += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
            dm[1] += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
            dm[2] += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
        }
        if (c->has_scales) {
            // backward.cu:426-487 (computeCov3D backward); no quaternion-norm Jacobian (B1); rot grads ADD onto K7's.
            const R* q = &c->rotations[4 * (size_t)idx];
            const R r = q[0], x = q[1], y = q[2], z = q[3];
            R Rm[3][3];
            quatToR(q, Rm);
            const R* sc = &c->scales[3 * (size_t)idx];
            const R s[3] = {c->scale_mod * sc[0], c->scale_mod * sc[1], c->scale_mod * sc[2]};
            // M[k][i] = s_k Rm[i][k];  dL_dSigma symmetric with halved off-diagonals
            const R dS[3][3] = {{dcov[0], R(0.5f) * dcov[1], R(0.5f) * dcov[2]},
                                {R(0.5f) * dcov[1], dcov[3], R(0.5f) * dcov[4]},
                                {R(0.5f) * dcov[2], R(0.5f) * dcov[4], dcov[5]}};
            // dL_dM = 2 M dSigma  (math 3x3, row k)
            R dM[3][3];
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < 3; j++)
                    dM[k][j] = R(2) * (s[k] * Rm[0][k] * dS[0][j] + s[k] * Rm[1][k] * dS[1][j] + s[k] * Rm[2][k] * dS[2][j]);
            // dL_dscale_k = sum_j Rm[j][k] * dM[k][j]   (glm: dot(Rt[k], dL_dMt[k]))
            R* dsc = dL_dscales + 3 * (size_t)idx;
            for (int k = 0; k < 3; k++) dsc[k] = Rm[0][k] * dM[k][0] + Rm[1][k] * dM[k][1] + Rm[2][k] * dM[k][2];
            // dL_dMt[k] *= s_k ; glm dL_dMt[a][b] = dM[a][b]
            R Mt[3][3];
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < 3; j++) Mt[k][j] = dM[k][j] * s[k];
            R dq[4];
            dq[0] = 2 * z * (Mt[0][1] - Mt[1][0]) + 2 * y * (Mt[2][0] - Mt[0][2]) + 2 * x * (Mt[1][2] - Mt[2][1]);
            dq[1] = 2 * y * (Mt[1][0] + Mt[0][1]) + 2 * z * (Mt[2][0] + Mt[0][2]) + 2 * r * (Mt[1][2] - Mt[2][1]) - 4 * x * (Mt[2][2] + Mt[1][1]);
            dq[2] = 2 * x * (Mt[1][0] + Mt[0][1]) + 2 * r * (Mt[2][0] - Mt[0][2]) + 2 * z * (Mt[1][2] + Mt[2][1]) - 4 * y * (Mt[2][2] + Mt[0][0]);
            dq[3] = 2 * r * (Mt[0][1] - Mt[1][0]) + 2 * x * (Mt[2][0] + Mt[0][2]) + 2 * y * (Mt[1][2] + Mt[2][1]) - 4 * z * (Mt[1][1] + Mt[0][0]);
            R* drot = dL_drot + 4 * (size_t)idx;
            for (int i = 0; i < 4; i++) drot[i] += dq[i];
        }
    }
}

template <typename R>
void ctx_copy(RastCtx<R>* c, int which, void* dst) {
    auto cp = [&](const auto& v) { std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    switch (which) {
        case 0: cp(c->point_list); break;
        case 1: cp(c->ranges); break;
        case 2: cp(c->tile_indices); break;
        case 3: cp(c->means2D); break;
        case 4: cp(c->depths); break;
        case 5: cp(c->conic_opacity); break;
        case 6: cp(c->rgb); break;
        case 7: cp(c->cov3D); break;
        case 8: cp(c->tiles_touched); break;
        case 9: cp(c->final_T); break;
        case 10: cp(c->n_contrib); break;
        case 11: cp(c->hit_normal_c); break;
        case 12: cp(c->hit_point_c); break;
        case 13: cp(c->clamped); break;
        case 14: cp(c->point_tile); break;
        case 15: cp(c->weight_sum); break;
        case 16: cp(c->n_blend); break;
        case 17: cp(c->pair_mask); break;
        default: break;
    }
}

}  // namespace

#define ORC_EXPORT extern "C" __attribute__((visibility("default")))

#define DEFINE_RAST_API(SUF, R)                                                                                              \
    ORC_EXPORT void* orc_rast_forward_##SUF(const int* ip, const double* fp, const R* bg, const R* means3D, const R* shs,    \
                                            const R* colors_precomp, const R* opacities, const R* scales, const R* rotations, \
                                            const R* cov3D_precomp, const R* view, const R* proj, const R* campos,            \
                                            const int32_t* tile_mask, R* out_color, R* out_depth, int32_t* out_hit_color,     \
                                            int32_t* out_hit_depth, R* out_hit_color_w, R* out_hit_depth_w, R* out_T,         \
                                            int32_t* n_touched, int32_t* radii) {                                             \
        return rast_forward<R>(ip, fp, bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, view,   \
                               proj, campos, tile_mask, out_color, out_depth, out_hit_color, out_hit_depth, out_hit_color_w,  \
                               out_hit_depth_w, out_T, n_touched, radii);                                                     \
    }                                                                                                                         \
    ORC_EXPORT void* orc_rast_forward_gated_##SUF(const int* ip, const double* fp, const R* bg, const R* means3D, const R* shs, \
                                                  const R* colors_precomp, const R* opacities, const R* scales,               \
                                                  const R* rotations, const R* cov3D_precomp, const R* view, const R* proj,   \
                                                  const R* campos, const int32_t* tile_mask, R* out_color, R* out_depth,      \
                                                  int32_t* out_hit_color, int32_t* out_hit_depth, R* out_hit_color_w,         \
                                                  R* out_hit_depth_w, R* out_T, int32_t* n_touched, int32_t* radii,           \
                                                  const int32_t* gauss_obj, const int32_t* pix_obj) {                         \
        return rast_forward<R>(ip, fp, bg, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, view,   \
                               proj, campos, tile_mask, out_color, out_depth, out_hit_color, out_hit_depth, out_hit_color_w,  \
                               out_hit_depth_w, out_T, n_touched, radii, gauss_obj, pix_obj);                                 \
    }                                                                                                                         \
    ORC_EXPORT void orc_rast_backward_##SUF(void* h, const R* dL_dpix, const R* dL_ddepth, R* dmeans3D, R* dsh, R* dcolors,   \
                                            R* dopacity, R* dscales, R* drot, R* dcov3D, R* dmeans2D, R* dconic) {            \
        rast_backward<R>((RastCtx<R>*)h, dL_dpix, dL_ddepth, dmeans3D, dsh, dcolors, dopacity, dscales, drot, dcov3D,         \
                         dmeans2D, dconic);                                                                                   \
    }                                                                                                                         \
    ORC_EXPORT void orc_rast_ctx_info_##SUF(void* h, int* out) {                                                             \
        auto* c = (RastCtx<R>*)h;                                                                                             \
        out[0] = c->num_rendered;                                                                                             \
        out[1] = (int)c->tile_indices.size();                                                                                 \
        out[2] = c->gx;                                                                                                       \
        out[3] = c->gy;                                                                                                       \
    }                                                                                                                         \
    ORC_EXPORT void orc_rast_ctx_copy_##SUF(void* h, int which, void* dst) { ctx_copy<R>((RastCtx<R>*)h, which, dst); }       \
    ORC_EXPORT void orc_rast_ctx_free_##SUF(void* h) { delete (RastCtx<R>*)h; }

DEFINE_RAST_API(f32, float)
DEFINE_RAST_API(f64, double)

// rasterizer_impl.cu:54-66,145-157 (checkFrustum / markVisible)
ORC_EXPORT int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_EXPORT void orc_mark_visible_f32(int P, const float* means3D, const float* view, const float* proj, uint8_t* present) {
    for (int i = 0; i < P; i++) {
        V3<float> pv;
        present[i] = in_frustum<float>(means3D, i, view, proj, pv) ? 1 : 0;
    }
}
