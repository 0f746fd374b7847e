// The early part of the per-Gaussian forward (K1): frustum test, 3D -> 2D covariance, conic, radius, pixel position, tile rect —
// the statements of /root/reference/submodules/diff-gaussian-rasterizer-depth/cuda_rasterizer/forward.cu:243-354 (preprocessCUDA) with
// :158-235 (computeCov2D / computeCov3D) and auxiliary.h:44-57, 139-165, in a fixed rounding order.  ONE definition, two homes:
//   preprocess_kernel (rast_forward.hip)       a launch of its own — the drop-in operator, and every frame that is not pre-zeroed;
//   bin_count_kernel<true> (rast_binning.hip)  at the head of the binning kernel, whose blocks own the same Gaussians and need exactly
//                                              these results (rect, conic, pixel position): a replayed iteration then has one launch
//                                              less and the binning kernel reads raw parameters instead of the tables it used to wait for.
// Same statements, same bits (tests/test_gpu_fused_mapping.py).  Everything here has internal linkage.
#pragma once
#include "dqo_k1_late.h"

namespace {

struct K1Early {
    int radius;                      // 0 = culled
    int rminx, rminy, rmaxx, rmaxy;  // tile rect (empty for a culled Gaussian)
    float4 co, xy;                   // what went into g.conic_opacity / g.xy_depth (only written when radius > 0)
};

// Writes the Gaussian's rows of conic_opacity / xy_depth (visible ones), rect16, radii and n_touched (all of them), and — LATE — the
// late part's tables.  view / proj: the matrices in registers (wave-uniform loads of the caller).
template <bool LATE>
__device__ __forceinline__ K1Early k1_early(const DqoView& v, const float (&view)[16], const float (&proj)[16], const float cam0,
                                            const float cam1, const float cam2, const int idx, const float* __restrict__ means3D,
                                            const float* __restrict__ scales, const float* __restrict__ rotations,
                                            const float* __restrict__ opacities, const float* __restrict__ shs,
                                            const float* __restrict__ colors_precomp, const int32_t* __restrict__ gobj, DqoGeomLayout& g,
                                            int32_t* __restrict__ radii_out, int32_t* __restrict__ n_touched_out) {
#pragma clang fp contract(off)
    K1Early e;
    e.co = make_float4(0.f, 0.f, 0.f, 0.f), e.xy = e.co;
    int radius = 0;
    int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
    do {
        // a hidden row (DqoRastInputs.row_flags) is no Gaussian of this render: culled like one behind the camera
        if (v.row_flags != nullptr && (v.row_flags[idx] & DQO_ROW_HIDDEN) != 0u) break;
        const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
        // in_frustum, auxiliary.h:139-165
        const float hx = proj[0] * px + proj[4] * py + proj[8] * pz + proj[12];
        const float hy = proj[1] * px + proj[5] * py + proj[9] * pz + proj[13];
        const float hw = proj[3] * px + proj[7] * py + proj[11] * pz + proj[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;
        const float tvx = view[0] * px + view[4] * py + view[8] * pz + view[12];
        const float tvy = view[1] * px + view[5] * py + view[9] * pz + view[13];
        const float tvz = view[2] * px + view[6] * py + view[10] * pz + view[14];
        if (tvz <= 0.2f || (double)projx < -1.3 || (double)projx > 1.3 || (double)projy < -1.3 || (double)projy > 1.3) break;
        const float opac = opacities[idx];  // (with the scales / rotation round: used only by the stores at the very end)
        // DqoObjectGate: the Gaussian's object id travels to the blend kernels in the spare word of its xy record
        const int obj_id = gobj != nullptr ? gobj[idx] : 0;
        // computeCov3D, forward.cu:202-235
        const float sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
        const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
        float Rm[3][3];
        quat_to_R(q, Rm);
        const float s[3] = {v.scale_mod * sx, v.scale_mod * sy, v.scale_mod * sz};
        float Mm[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 3; i++) Mm[k][i] = s[k] * Rm[i][k];
        float c3[6];
        {
            int o = 0;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = i; j < 3; j++) c3[o++] = Mm[0][i] * Mm[0][j] + Mm[1][i] * Mm[1][j] + Mm[2][i] * Mm[2][j];
        }
        // computeCov2D, forward.cu:158-197
        const float limx = 1.3f * v.tanfovx, limy = 1.3f * v.tanfovy;
        const float txtz = tvx / tvz, tytz = tvy / tvz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * tvz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * tvz;
        const float J00 = v.focal_x / tvz, J02 = -(v.focal_x * tx) / (tvz * tvz);
        const float J11 = v.focal_y / tvz, J12 = -(v.focal_y * ty) / (tvz * tvz);
        float A0[3], A1[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A0[j] = J00 * view[j * 4 + 0] + J02 * view[j * 4 + 2];
            A1[j] = J11 * view[j * 4 + 1] + J12 * view[j * 4 + 2];
        }
        const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
        float VA0[3], VA1[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            VA0[i] = V[i][0] * A0[0] + V[i][1] * A0[1] + V[i][2] * A0[2];
            VA1[i] = V[i][0] * A1[0] + V[i][1] * A1[1] + V[i][2] * A1[2];
        }
        const float ca = A0[0] * VA0[0] + A0[1] * VA0[1] + A0[2] * VA0[2] + 0.3f;
        const float cb = A0[0] * VA1[0] + A0[1] * VA1[1] + A0[2] * VA1[2];
        const float cc = A1[0] * VA1[0] + A1[1] * VA1[1] + A1[2] * VA1[2] + 0.3f;
        const float det = ca * cc - cb * cb;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float conx = cc * det_inv, cony = -cb * det_inv, conz = ca * det_inv;
        const float mid = 0.5f * (ca + cc);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(v.color_sigma * sqrtf(fmaxf(lambda1, lambda2)));
        // ndc2Pix(v, S, c) = v * S * 0.5 + c with double intermediate, auxiliary.h:44-47
        const float pixx = (float)((double)(projx * (float)v.W) * 0.5 + (double)v.cx);
        const float pixy = (float)((double)(projy * (float)v.H) * 0.5 + (double)v.cy);
        // getRect, auxiliary.h:49-57
        const int ir = (int)my_radius;
        rminx = min(v.gx, max(0, (int)((pixx - (float)ir) / (float)DQO_TILE)));
        rminy = min(v.gy, max(0, (int)((pixy - (float)ir) / (float)DQO_TILE)));
        rmaxx = min(v.gx, max(0, (int)((pixx + (float)ir + (float)(DQO_TILE - 1)) / (float)DQO_TILE)));
        rmaxy = min(v.gy, max(0, (int)((pixy + (float)ir + (float)(DQO_TILE - 1)) / (float)DQO_TILE)));
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) break;
        if constexpr (LATE) k1_late_part(v, view, cam0, cam1, cam2, idx, px, py, pz, tvx, tvy, tvz, sx, sy, sz, Rm, shs, colors_precomp, g);
        radius = ir;
        e.co = make_float4(conx, cony, conz, opac);
        e.xy = make_float4(pixx, pixy, tvz, __int_as_float(gobj != nullptr ? obj_id : ir));
        g.conic_opacity[idx] = e.co;
        g.xy_depth[idx] = e.xy;
    } while (false);
    radii_out[idx] = radius;
    n_touched_out[idx] = 0;
    g.rect16[idx] = make_uint2((uint32_t)rminx | ((uint32_t)rmaxx << 16), (uint32_t)rminy | ((uint32_t)rmaxy << 16));
    e.radius = radius, e.rminx = rminx, e.rminy = rminy, e.rmaxx = rmaxx, e.rmaxy = rmaxy;
    return e;
}

}  // namespace
