// Small per-point kernels of the map-growth step (SURVEY.md §8 row f: map growth) for gfx950.  The growth step is bound by the HOST — a
// chain of torch ops with host-side decisions in between — so each of these replaces a chain of element-wise torch ops of the reference
// by one launch, statement by statement (separate IEEE multiplies and adds in the torch chain's order, correctly rounded sqrt / divide:
// the results equal the torch restatements in dqo_mapgrowth bit for bit, tests/test_gpu_mapgrowth.py):
//
//   growth_scales_kernel  GaussianPointCloud.update_geometry's scale initialisation (SLAM/gaussian_pointcloud.py:519-556) behind its two
//                         searches, for the per-object job: a new point's three nearest neighbours among (the other new points of its
//                         object, the existing Gaussians of its object), the gaps to their 3-sigma spheres -> scale, invalid flag;
//   growth_inside_kernel  Mapping.temp_points_filter's decision (SLAM/multiprocess/mapper.py:1372-1380): inside one of the (up to) three
//                         nearest existing Gaussians = closer than 0.6 x its radius;
//   error_maps_kernel     the per-pixel colour / depth error images that feed accumulate_gaussian_error (mapper.py:1016-1033).
#include "dqo_common.h"

namespace {

__global__ __launch_bounds__(256) void growth_scales_kernel(int n, const float* __restrict__ xyz, const int32_t* __restrict__ obj,
                                                            const float* __restrict__ radius, const int32_t* __restrict__ i_new,
                                                            const float* __restrict__ d2_old, const int32_t* __restrict__ i_old,
                                                            const float* __restrict__ extra_radius, float reach2, float min_radius,
                                                            float max_radius, float* __restrict__ scales, uint8_t* __restrict__ invalid) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float inf = __builtin_inff();
    // six candidates in the torch chain's order (three other new points, three existing points); a missing one is at infinity
    float cd[6], cr[6];
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const int32_t oi = obj[i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cd[k] = inf, cr[k] = 0.f;
        if (i_new != nullptr) {
            const int32_t jr = i_new[3 * i + k];
            const int j = jr < n ? (jr < 0 ? 0 : jr) : n - 1;  // (clamp(max = n - 1); INT_MAX = fewer than three other new points)
            const float dx = x - xyz[3 * j], dy = y - xyz[3 * j + 1], dz = z - xyz[3 * j + 2];
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            cr[k] = radius[j];
            if (jr < n && obj[j] == oi && d2 < reach2) cd[k] = d2;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cd[3 + k] = inf, cr[3 + k] = 0.f;
        if (i_old != nullptr) {
            const int32_t j = i_old[3 * i + k];
            cr[3 + k] = extra_radius[j < 0 ? 0 : j];
            if (j >= 0) cd[3 + k] = d2_old[3 * i + k];
        }
    }
    // the three smallest, ascending (torch.topk(largest=False): equal distances keep the candidates' order)
    float td[3] = {inf, inf, inf}, tr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        float d = cd[k], r = cr[k];
#pragma unroll
        for (int s = 0; s < 3; s++) {
            if (d < td[s]) {
                const float sd = td[s], sr = tr[s];
                td[s] = d, tr[s] = r;
                d = sd, r = sr;
            }
        }
    }
    float g[3];
#pragma unroll
    for (int s = 0; s < 3; s++) g[s] = sqrtf(td[s]) - 3.f * tr[s];
    invalid[i] = (g[0] < 0.f || g[1] < 0.f || g[2] < 0.f) ? 1 : 0;
    // (torch divides a tensor by a host scalar as a multiplication by its float reciprocal)
    const float sc = sqrtf(((g[0] * g[0] + g[1] * g[1]) + g[2] * g[2]) * (1.0f / 3.0f));
    scales[i] = fminf(fmaxf(sc, min_radius), max_radius);  // torch.clip
}

__global__ __launch_bounds__(256) void growth_inside_kernel(int n, const float* __restrict__ d2, const int32_t* __restrict__ idx,
                                                            const float* __restrict__ radius, uint8_t* __restrict__ inside) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool in = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int32_t j = idx[3 * i + k];
        in = in || (j >= 0 && sqrtf(d2[3 * i + k]) < radius[j < 0 ? 0 : j] * 0.6f);
    }
    inside[i] = in ? 1 : 0;
}

// depth_err = max(gt_depth - depth, 0), zero where gt_depth == 0, the pixel has no depth hit (index -1) or lies outside the mask;
// color_err = sum_c |gt_color - render| (channel order 0, 1, 2), zero where gt_depth == 0 or outside the mask
__global__ __launch_bounds__(256) void error_maps_kernel(int64_t HW, const float* __restrict__ gt_color, const float* __restrict__ gt_depth,
                                                         const float* __restrict__ render, const float* __restrict__ depth,
                                                         const int32_t* __restrict__ depth_index, const uint8_t* __restrict__ mask,
                                                         float* __restrict__ color_err, float* __restrict__ depth_err) {
#pragma clang fp contract(off)
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float gd = gt_depth[p];
    const bool m = mask == nullptr || mask[p] != 0;
    const bool off = gd == 0.f || !m;
    const float de = fmaxf(gd - depth[p], 0.f);
    depth_err[p] = (off || depth_index[p] == -1) ? 0.f : de;
    const float ce = (fabsf(gt_color[p] - render[p]) + fabsf(gt_color[HW + p] - render[HW + p])) + fabsf(gt_color[2 * HW + p] - render[2 * HW + p]);
    color_err[p] = off ? 0.f : ce;
}

}  // namespace

int dqo_launch_growth_scales(int n, const float* xyz, const int32_t* obj, const float* radius, const int32_t* i_new, const float* d2_old,
                             const int32_t* i_old, const float* extra_radius, float reach2, float min_radius, float max_radius, float* scales,
                             uint8_t* invalid, hipStream_t s) {
    if (n > 0)
        DQO_LAUNCH("growth_scales_kernel", growth_scales_kernel, dim3((n + 255) / 256), dim3(256), s, n, xyz, obj, radius, i_new, d2_old, i_old,
                   extra_radius, reach2, min_radius, max_radius, scales, invalid);
    return DQO_OK;
}

int dqo_launch_growth_inside(int n, const float* d2, const int32_t* idx, const float* radius, uint8_t* inside, hipStream_t s) {
    if (n > 0) DQO_LAUNCH("growth_inside_kernel", growth_inside_kernel, dim3((n + 255) / 256), dim3(256), s, n, d2, idx, radius, inside);
    return DQO_OK;
}

int dqo_launch_error_maps(int64_t HW, const float* gt_color, const float* gt_depth, const float* render, const float* depth,
                          const int32_t* depth_index, const uint8_t* mask, float* color_err, float* depth_err, hipStream_t s) {
    if (HW > 0)
        DQO_LAUNCH("error_maps_kernel", error_maps_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), s, HW, gt_color, gt_depth, render, depth,
                   depth_index, mask, color_err, depth_err);
    return DQO_OK;
}
