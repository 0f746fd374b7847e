// The per-object job's temp_points_attach decision (SURVEY.md §8 row f: map growth) for gfx950 — the torch chain of
// /root/reference/SLAM/multiprocess/mapper.py:1384-1430 (Mapping.temp_points_attach: project the new points with get_uv,
// scene/cameras.py:207-214; look up the stable cloud's colour-hit index at their pixels; point-to-plane test against that Gaussian's
// normal, SLAM/gaussian_pointcloud.py:780-791 + utils/general_utils.py:108-137) as two launches, one thread per candidate:
//
//   attach_pixels_kernel   candidate -> its pixel (or "outside"), and the sparse object gate of the attach render: pixel_object with every
//                          pixel that holds no candidate ownerless, and per 16x16 tile the 64-bit set of the owners among its candidate
//                          pixels (DqoObjectGate.pixel_object / .tile_objects) — the render then only works for one pixel in twenty;
//   attach_decide_kernel   after the gated render of the stable cloud: hit index / weight at the pixel, the hit Gaussian's normal from its
//                          raw quaternion and raw scales, |(x_stable - x) . n| < plane_thr, same object.
//
// Both follow dqo_mapgrowth.temp_points_pixels / temp_points_attach_mask_per_object operation by operation (separate IEEE multiplies and
// adds in the same order, correctly rounded sqrt and divide), so that the masks are equal bit for bit (tests/test_gpu_mapgrowth.py).  A
// growth step is bound by the HOST (a chain of small launches with host-side decisions in between): the two launches replace ~110
// torch ops of that chain.
#include "dqo_common.h"

namespace {

__global__ __launch_bounds__(256) void attach_clear_kernel(int64_t HW, int T, int32_t* __restrict__ sparse,
                                                           unsigned long long* __restrict__ tile_objects) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < HW) sparse[i] = -1;
    if (i < T) tile_objects[i] = 0ull;
}

// V: the op's viewmatrix (world_view_transform as the reference passes it: p_view = p @ V[:3, :3] + V[3, :3]) = w2c transposed
__global__ __launch_bounds__(256) void attach_pixels_kernel(int n, const float* __restrict__ xyz, const float* __restrict__ V, float fx,
                                                            float fy, float cx, float cy, int W, int H, int gx,
                                                            const int32_t* __restrict__ pixel_object, int32_t* __restrict__ lin,
                                                            int32_t* __restrict__ sparse, unsigned long long* __restrict__ tile_objects) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    // xyz @ R.T + t (R = w2c[:3, :3], t = w2c[:3, 3]), then @ K.T with K = [[fx, 0, cx], [0, fy, cy], [0, 0, 1]]
    const float xc = ((x * V[0] + y * V[4]) + z * V[8]) + V[12];
    const float yc = ((x * V[1] + y * V[5]) + z * V[9]) + V[13];
    const float zc = ((x * V[2] + y * V[6]) + z * V[10]) + V[14];
    const float uf = (xc * fx + zc * cx) / zc, vf = (yc * fy + zc * cy) / zc;
    // trunc toward zero (`.long()`): pixel >= 0 <=> coordinate > -1 (a point up to one pixel left of / above the image counts as inside)
    const bool inside = uf > -1.0f && uf < (float)W && vf > -1.0f && vf < (float)H;  // (NaN: outside)
    int32_t l = -1;
    if (inside) {
        const int u = (int)uf, v = (int)vf;
        l = v * W + u;
        const int32_t owner = pixel_object[l];
        sparse[l] = owner;  // (every candidate of the pixel writes the same value)
        if (owner >= 0) atomicOr(&tile_objects[(v >> 4) * gx + (u >> 4)], 1ull << (owner & 63));
    }
    lin[i] = l;
}

__global__ __launch_bounds__(256) void attach_decide_kernel(int n, const float* __restrict__ xyz, const float* __restrict__ opacity,
                                                            const int32_t* __restrict__ obj, const int32_t* __restrict__ lin,
                                                            const int32_t* __restrict__ hit_index, const float* __restrict__ hit_weight,
                                                            const float* __restrict__ sxyz, const float* __restrict__ scaling_raw,
                                                            const float* __restrict__ rotation_raw, const int32_t* __restrict__ gobj,
                                                            float plane_thr, float opacity_low, uint8_t* __restrict__ out) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool ok = false;
    const int32_t l = lin[i];
    if (l >= 0 && opacity[i] > opacity_low) {
        const int32_t s = hit_index[l];
        // index 0 with weight 0 = the op's zero fill of a never-rendered tile: no hit
        if (s >= 0 && !(s == 0 && hit_weight[l] == 0.f) && gobj[s] == obj[i]) {
            // GaussianPointCloud.get_normal: the column of R(q / |q|) along the smallest scale, normalised with + 1e-8
            float r = rotation_raw[4 * s], qx = rotation_raw[4 * s + 1], qy = rotation_raw[4 * s + 2], qz = rotation_raw[4 * s + 3];
            const float nq = sqrtf(((r * r + qx * qx) + qy * qy) + qz * qz);
            r = r / nq, qx = qx / nq, qy = qy / nq, qz = qz / nq;
            const float s0 = scaling_raw[3 * s], s1 = scaling_raw[3 * s + 1], s2 = scaling_raw[3 * s + 2];
            const int k = (s0 <= s1 && s0 <= s2) ? 0 : (s1 <= s2 ? 1 : 2);  // argmin, the first of equal minima
            float a, b, c;
            if (k == 0) a = 1.f - 2.f * (qy * qy + qz * qz), b = 2.f * (qx * qy + r * qz), c = 2.f * (qx * qz - r * qy);
            else if (k == 1) a = 2.f * (qx * qy - r * qz), b = 1.f - 2.f * (qx * qx + qz * qz), c = 2.f * (qy * qz + r * qx);
            else a = 2.f * (qx * qz + r * qy), b = 2.f * (qy * qz - r * qx), c = 1.f - 2.f * (qx * qx + qy * qy);
            const float nn = sqrtf((a * a + b * b) + c * c) + 1e-8f;
            a = a / nn, b = b / nn, c = c / nn;
            const float d = ((sxyz[3 * s] - xyz[3 * i]) * a + (sxyz[3 * s + 1] - xyz[3 * i + 1]) * b) + (sxyz[3 * s + 2] - xyz[3 * i + 2]) * c;
            ok = fabsf(d) < plane_thr;
        }
    }
    out[i] = ok ? 1 : 0;
}

}  // namespace

int dqo_launch_attach_pixels(int n, const float* xyz, const float* V, float fx, float fy, float cx, float cy, int W, int H,
                             const int32_t* pixel_object, int32_t* lin, int32_t* sparse, unsigned long long* tile_objects, hipStream_t s) {
    const int gx = (W + DQO_TILE - 1) / DQO_TILE, gy = (H + DQO_TILE - 1) / DQO_TILE;
    const int64_t HW = (int64_t)W * H;
    DQO_LAUNCH("attach_clear_kernel", attach_clear_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), s, HW, gx * gy, sparse, tile_objects);
    if (n > 0)
        DQO_LAUNCH("attach_pixels_kernel", attach_pixels_kernel, dim3((n + 255) / 256), dim3(256), s, n, xyz, V, fx, fy, cx, cy, W, H, gx,
                   pixel_object, lin, sparse, tile_objects);
    return DQO_OK;
}

int dqo_launch_attach_decide(int n, const float* xyz, const float* opacity, const int32_t* obj, const int32_t* lin, const int32_t* hit_index,
                             const float* hit_weight, const float* sxyz, const float* scaling_raw, const float* rotation_raw,
                             const int32_t* gobj, float plane_thr, float opacity_low, uint8_t* out, hipStream_t s) {
    if (n > 0)
        DQO_LAUNCH("attach_decide_kernel", attach_decide_kernel, dim3((n + 255) / 256), dim3(256), s, n, xyz, opacity, obj, lin, hit_index,
                   hit_weight, sxyz, scaling_raw, rotation_raw, gobj, plane_thr, opacity_low, out);
    return DQO_OK;
}
