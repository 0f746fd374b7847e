// The late part of the per-Gaussian forward (forward.cu:104-155 colour from SH, :54-74 / 779-785 surfel normal and camera-space point) and
// where it runs (preprocess_kernel<true>, tile_sort_wave_kernel<true> or tile_sort_kernel<true>, rast_forward.hip).  Everything here has internal linkage.
#pragma once
#include <cstdlib>

#include "dqo_common.h"
#include "dqo_gauss_chain.h"

namespace {

__device__ __forceinline__ void quat_to_R(const float4 q, float Rm[3][3]) {
#pragma clang fp contract(off)
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    Rm[0][0] = 1.f - 2.f * (y * y + z * z);
    Rm[0][1] = 2.f * (x * y - r * z);
    Rm[0][2] = 2.f * (x * z + r * y);
    Rm[1][0] = 2.f * (x * y + r * z);
    Rm[1][1] = 1.f - 2.f * (x * x + z * z);
    Rm[1][2] = 2.f * (y * z - r * x);
    Rm[2][0] = 2.f * (x * z - r * y);
    Rm[2][1] = 2.f * (y * z + r * x);
    Rm[2][2] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ int arg_min3(float a, float b, float c) { return (a <= b && a <= c) ? 0 : ((b <= a && b <= c) ? 1 : 2); }
__device__ __forceinline__ int arg_max3(float a, float b, float c) { return (a >= b && a >= c) ? 0 : ((b >= a && b >= c) ? 1 : 2); }

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

// Where the late part of the per-Gaussian forward runs (k1_late_part below).  Inside preprocess_kernel it costs 15-20 us on a 500 k map
// and 100 us on a 2 M one; nothing before the blend kernel needs its results, and the sort kernels between are serial chains (per-tile
// sorts: a few busy waves) or LDS / barrier-bound (the long-list sorts) that leave the memory pipes idle.  So it rides one of their
// launches as EXTRA BLOCKS behind the sort blocks (one heterogeneous launch: no second stream, the sorts start first):
//   0: inside preprocess_kernel<true> (DQO_K1_WHERE=0 forces it: the A/B baseline);
//   1: tile_sort_wave_kernel<true>  — maps of up to 768 Ki Gaussians: -5..7 us per iteration on cfg 3 (-15 us with every Gaussian in
//      view), -3 us on cfg 2; on bigger maps there is more late work than the per-tile sorts can hide (+-0 on cfg 4, +12 us on cfg 5);
//   2: tile_sort_kernel<true>       — bigger maps, whose long lists keep that kernel running for 30-95 us: -56 us per iteration on
//      cfg 5 (preprocess 154 -> 48 us, long-list sort 94 -> 131 us), +-0 on cfg 4.
// Measured and not used: extra blocks of the bin_count_kernel launch (its atomics keep the memory system busy: nothing hides, bin_count
// 55 -> 72 us on cfg 3, 137 -> 276 us on cfg 5) and a kernel of its own on a side stream (no gain on any configuration).
// Same-box A/B with tools/ab_where.sh.
static int dqo_k1_where(int P) {
    static const int forced = [] {
        const char* e = getenv("DQO_K1_WHERE");
        return (e != nullptr && e[0] != '\0') ? atoi(e) : -1;
    }();
    if (forced >= 0 && forced <= 2) return forced;
    return P <= 786432 ? 1 : 2;
}

// The part of the per-Gaussian forward that only the blend kernels and the backward need (colour from SH + its direction derivative,
// surfel normal, camera-space point): the statements of forward.cu:104-155 and :54-74, 779-785.  Called by preprocess_kernel itself, or
// — where dqo_k1_where says so — by the extra blocks of a sort launch (k1_late_block), where it runs beside the sorts:
// those keep a few waves busy for their whole serial chain and leave the rest of the GPU idle, this part is memory traffic (the 192-byte
// SH row) and plain arithmetic that nothing before the blend kernel waits for.
__device__ __forceinline__ void k1_late_part(const DqoView& v, const float (&view)[16], const float cam0, const float cam1, const float cam2,
                                             const int idx, const float px, const float py, const float pz, const float tvx, const float tvy,
                                             const float tvz, const float sx, const float sy, const float sz, const float (&Rm)[3][3],
                                             const float* __restrict__ shs, const float* __restrict__ colors_precomp, const DqoGeomLayout& g) {
#pragma clang fp contract(off)
    // colour: computeColorFromSH, forward.cu:104-155
    float rgb[3];
    float dd[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t clampbits = 0;
    if (colors_precomp == nullptr) {
        const float dxx = px - cam0, dyy = py - cam1, dzz = pz - cam2;
        const float len = sqrtf(dxx * dxx + dyy * dyy + dzz * dzz);
        const float x = dxx / len, y = dyy / len, z = dzz / len;
        // all coefficients of the active degree in ONE batch of loads: fetched inside the per-degree blocks below they
        // would come in twelve small groups (three channels x four degrees), each waited for before the next is issued
        const float* shp = shs + (size_t)idx * v.M * 3;
        float sh[48];
        if (v.D >= 3) {
#pragma unroll
            for (int i = 0; i < 48; i++) sh[i] = shp[i];
        } else if (v.D == 2) {
#pragma unroll
            for (int i = 0; i < 27; i++) sh[i] = shp[i];
        } else if (v.D == 1) {
#pragma unroll
            for (int i = 0; i < 12; i++) sh[i] = shp[i];
        } else {
#pragma unroll
            for (int i = 0; i < 3; i++) sh[i] = shp[i];
        }
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            float result = kSH_C0 * sh[ch];
            if (v.D > 0) {
                result = result - kSH_C1 * y * sh[3 + ch] + kSH_C1 * z * sh[6 + ch] - kSH_C1 * x * sh[9 + ch];
                if (v.D > 1) {
                    result = result + kSH_C2[0] * xy * sh[12 + ch] + kSH_C2[1] * yz * sh[15 + ch] +
                             kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + kSH_C2[3] * xz * sh[21 + ch] +
                             kSH_C2[4] * (xx - yy) * sh[24 + ch];
                    if (v.D > 2) {
                        result = result + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] + kSH_C3[1] * xy * z * sh[30 + ch] +
                                 kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                                 kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                                 kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + kSH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                                 kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                    }
                }
            }
            result += 0.5f;
            if (result < 0.f) clampbits |= 1u << ch;
            rgb[ch] = fmaxf(result, 0.0f);
        }
        // d(colour)/d(direction) for the backward's view-direction gradient (backward.cu:168-258): a function of the coefficients
        // and the direction alone, both in registers here — 9 floats instead of the backward gathering the 48-float row again
        dqo_sh_dir_grad(v.D, sh, x, y, z, dd);
    } else {
        rgb[0] = colors_precomp[3 * idx], rgb[1] = colors_precomp[3 * idx + 1], rgb[2] = colors_precomp[3 * idx + 2];
    }
    // surfel normal / camera-space point, hoisted from the blend loop (forward.cu:54-74, 779-785)
    const int naxis = arg_min3(sx, sy, sz), maxis = arg_max3(sx, sy, sz);
    const float nwx = Rm[0][naxis], nwy = Rm[1][naxis], nwz = Rm[2][naxis];
    const float smax = (maxis == 0 ? sx : (maxis == 1 ? sy : sz)) * v.scale_mod;
    const float ncx = view[0] * nwx + view[4] * nwy + view[8] * nwz;
    const float ncy = view[1] * nwx + view[5] * nwy + view[9] * nwz;
    const float ncz = view[2] * nwx + view[6] * nwy + view[10] * nwz;
    const float npc = tvx * ncx + tvy * ncy + tvz * ncz;

    g.rgb_smax[idx] = make_float4(rgb[0], rgb[1], rgb[2], smax);
    g.normal_c[idx] = make_float4(ncx, ncy, ncz, npc);
    // .w = max of the RAW scales: the backward's depth test uses it without scale_modifier (backward.cu:1009, quirk B6)
    g.point_c[idx] = make_float4(tvx, tvy, tvz, maxis == 0 ? sx : (maxis == 1 ? sy : sz));
    g.clamped[idx] = (uint8_t)clampbits;
    if (colors_precomp == nullptr) {
        float4* const ddp = g.drgb_dir + 3 * (size_t)idx;
        ddp[0] = make_float4(dd[0], dd[1], dd[2], 0.f), ddp[1] = make_float4(dd[3], dd[4], dd[5], 0.f);
        ddp[2] = make_float4(dd[6], dd[7], dd[8], 0.f);
    }
}

template <int THREADS>
__device__ __forceinline__ void k1_late_block(const DqoK1Late& a, const DqoGeomLayout& g, const int block) {
#pragma clang fp contract(off)
    const int idx = block * THREADS + (int)threadIdx.x;
    if (idx >= a.v.P) return;
    // the Gaussians preprocess_kernel kept: a non-empty tile rect (every cull of forward.cu:238-354 leaves an empty one)
    const uint2 rc = g.rect16[idx];
    if (!(((rc.x >> 16) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.y & 0xffffu)))) return;
    float view[16];
#pragma unroll
    for (int i = 0; i < 16; i++) view[i] = a.v.view[i];
    const float px = a.means3D[3 * idx], py = a.means3D[3 * idx + 1], pz = a.means3D[3 * idx + 2];
    const float tvx = view[0] * px + view[4] * py + view[8] * pz + view[12];
    const float tvy = view[1] * px + view[5] * py + view[9] * pz + view[13];
    const float tvz = view[2] * px + view[6] * py + view[10] * pz + view[14];
    const float sx = a.scales[3 * idx], sy = a.scales[3 * idx + 1], sz = a.scales[3 * idx + 2];
    const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
    float Rm[3][3];
    quat_to_R(q, Rm);
    k1_late_part(a.v, view, a.v.campos[0], a.v.campos[1], a.v.campos[2], idx, px, py, pz, tvx, tvy, tvz, sx, sy, sz, Rm, a.shs, a.colors_precomp, g);
}
}  // namespace
