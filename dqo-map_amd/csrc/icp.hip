// Point-to-plane ICP normal equations for gfx950.
//
// Replaces, per Gauss-Newton iteration of /root/reference/SLAM/icp.py:
//   ICP.compute_residuals_jacobian :51-108 (transform, projective data association through warp_features / grid_sample
//   "nearest" :131-149, masks, residual, Jacobian), ICP.compute_jtj :110-115 and ICP.compute_jtr :117-123
// i.e. ~40 eager torch kernels and three [HW, 6]-sized temporaries, by ONE pass over the two vertex / normal maps that leaves
// J^T J (6x6), J^T r (6) and the number of valid pixels.  The 6x6 solve and the se(3) update stay on the host side (icp.py:
// 125-129, 248-312), as in the reference.
//
// Per pixel: p = R v0 + t, n = R n0; (u, v) = projection of p; q, m = vertex1 / normal1 at the nearest pixel to (u, v);
// r = m . (p - q); J = [p x m, m] (rotation first, icp.py:95-100); the pixel counts iff it is in view, both depths are
// positive, |p - q| <= distance_threshold and n . m > normal_threshold.  Sums are accumulated in fp64 per thread and reduced
// in a fixed order (block partials, then one finishing block): bitwise reproducible.
#include "dqo_common.h"

namespace {

constexpr int ICP_THREADS = 256;
constexpr int ICP_NSUM = 28;  // 21 upper-triangular J^T J + 6 J^T r + valid count

struct IcpPose {
    float R[9], t[3];
};

__global__ __launch_bounds__(ICP_THREADS) void icp_partial_kernel(int H, int W, const float* __restrict__ vertex0,
                                                                  const float* __restrict__ vertex1, const float* __restrict__ normal0,
                                                                  const float* __restrict__ normal1, const float* __restrict__ pose10,
                                                                  float fx, float fy, float cx, float cy, float dist_thr,
                                                                  float normal_thr, double* __restrict__ partial) {
#pragma clang fp contract(off)
    __shared__ double s_red[ICP_THREADS / 64][ICP_NSUM];
    float R[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) R[3 * i + j] = pose10[4 * i + j];
        t[i] = pose10[4 * i + 3];
    }
    double acc[ICP_NSUM];
#pragma unroll
    for (int i = 0; i < ICP_NSUM; i++) acc[i] = 0.0;
    const int HW = H * W;
    const float hw = (float)(W - 1) / 2.0f, hh = (float)(H - 1) / 2.0f;
    for (int p = blockIdx.x * ICP_THREADS + threadIdx.x; p < HW; p += gridDim.x * ICP_THREADS) {
        const float v0x = vertex0[3 * p], v0y = vertex0[3 * p + 1], v0z = vertex0[3 * p + 2];
        const bool mask0 = v0z > 0.0f;
        // torch's matmul of a 3x3 by a 3-vector in fp32: sum in index order
        const float px = (R[0] * v0x + R[1] * v0y + R[2] * v0z) + t[0];
        const float py = (R[3] * v0x + R[4] * v0y + R[5] * v0z) + t[1];
        const float pz = (R[6] * v0x + R[7] * v0y + R[8] * v0z) + t[2];
        const float n0x = normal0[3 * p], n0y = normal0[3 * p + 1], n0z = normal0[3 * p + 2];
        const float nx = R[0] * n0x + R[1] * n0y + R[2] * n0z;
        const float ny = R[3] * n0x + R[4] * n0y + R[5] * n0z;
        const float nz = R[6] * n0x + R[7] * n0y + R[8] * n0z;
        const float u = (px / pz) * fx + cx, v = (py / pz) * fy + cy;
        const bool inview = (u > 0.f) && (u < (float)(W - 1)) && (v > 0.f) && (v < (float)(H - 1));
        bool ok = mask0 && inview;  // (NaN coordinates fail `inview`)
        float rx = 0.f, J0 = 0.f, J1 = 0.f, J2 = 0.f, J3 = 0.f, J4 = 0.f, J5 = 0.f;
        if (ok) {
            // warp_features -> grid_sample(mode="nearest", padding_mode="border", align_corners=True): normalise, unnormalise,
            // clip, round half to even
            const float un = u / hw - 1.0f, vn = v / hh - 1.0f;
            float gxf = ((un + 1.0f) / 2.0f) * (float)(W - 1), gyf = ((vn + 1.0f) / 2.0f) * (float)(H - 1);
            gxf = fminf((float)(W - 1), fmaxf(gxf, 0.f));
            gyf = fminf((float)(H - 1), fmaxf(gyf, 0.f));
            const int xi = (int)nearbyintf(gxf), yi = (int)nearbyintf(gyf);
            const int q = yi * W + xi;
            const float qx = vertex1[3 * q], qy = vertex1[3 * q + 1], qz = vertex1[3 * q + 2];
            const float mx = normal1[3 * q], my = normal1[3 * q + 1], mz = normal1[3 * q + 2];
            const float dx = px - qx, dy = py - qy, dz = pz - qz;
            const bool mask1 = qz > 0.f;
            const bool ndm = ((nx * mx + ny * my) + nz * mz) > normal_thr;
            const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
            ok = mask1 && ndm && !(dn > dist_thr);
            if (ok) {
                rx = (mx * dx + my * dy) + mz * dz;
                // J_rot = -(m^T [p]_x) = p x m (icp.py:96-97), J_trs = m
                J0 = -(my * pz - mz * py), J1 = -(-mx * pz + mz * px), J2 = -(mx * py - my * px);
                J3 = mx, J4 = my, J5 = mz;
            }
        }
        if (ok) {
            const double J[6] = {J0, J1, J2, J3, J4, J5};
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = a; b < 6; b++) acc[k++] += J[a] * J[b];
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] += J[a] * (double)rx;
            acc[27] += 1.0;
        }
    }
    // wave reduction, then the block's four waves in order
#pragma unroll
    for (int i = 0; i < ICP_NSUM; i++) {
        double x = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        acc[i] = x;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int i = 0; i < ICP_NSUM; i++) s_red[wave][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < ICP_NSUM) {
        double x = 0.0;
        for (int w = 0; w < ICP_THREADS / 64; w++) x += s_red[w][threadIdx.x];
        partial[(size_t)blockIdx.x * ICP_NSUM + threadIdx.x] = x;
    }
}

__global__ void icp_finish_kernel(int nblk, const double* __restrict__ partial, float* __restrict__ JtJ, float* __restrict__ JtR,
                                  int32_t* __restrict__ valid_count) {
    const int i = threadIdx.x;
    if (i >= ICP_NSUM) return;
    double x = 0.0;
    for (int b = 0; b < nblk; b++) x += partial[(size_t)b * ICP_NSUM + i];
    if (i < 21) {
        int k = 0;
        for (int a = 0; a < 6; a++)
            for (int c = a; c < 6; c++, k++)
                if (k == i) JtJ[6 * a + c] = JtJ[6 * c + a] = (float)x;
    } else if (i < 27) {
        JtR[i - 21] = (float)x;
    } else {
        *valid_count = (int32_t)x;
    }
}

}  // namespace

size_t dqo_icp_ws_bytes(void) { return sizeof(double) * ICP_NSUM * 1024; }

int dqo_launch_icp(int H, int W, const float* vertex0, const float* vertex1, const float* normal0, const float* normal1, const float* pose10,
                   float fx, float fy, float cx, float cy, float dist_thr, float normal_thr, float* JtJ, float* JtR, int32_t* valid_count,
                   void* ws, hipStream_t s) {
    const int HW = H * W;
    const int nblk = min(1024, (HW + ICP_THREADS - 1) / ICP_THREADS);
    double* partial = (double*)ws;
    DQO_LAUNCH("icp_partial_kernel", icp_partial_kernel, dim3(nblk), dim3(ICP_THREADS), s, H, W, vertex0, vertex1, normal0, normal1, pose10, fx,
               fy, cx, cy, dist_thr, normal_thr, partial);
    DQO_LAUNCH("icp_finish_kernel", icp_finish_kernel, dim3(1), dim3(64), s, nblk, partial, JtJ, JtR, valid_count);
    return DQO_OK;
}
