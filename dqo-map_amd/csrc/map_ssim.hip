// Fused SSIM term of the mapping loss for gfx950 ("masked L1 + depth L1 + SSIM in one pass") — the unmasked branch of
// Mapping.loss_update, /root/reference/SLAM/multiprocess/mapper.py:839-845:  loss += 0.2 * (1 - ssim(image, gt)),  ssim =
// /root/reference/utils/loss_utils.py:41-100 (11 x 11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over
// all channels and pixels), and its gradient w.r.t. the rendered image (torch autograd through five conv2d calls in the reference:
// ~40 eager launches forward + backward).
//
// Two kernels, a 16 x 16 pixel tile of one channel per workgroup, the 26 x 26 halo tile staged in LDS, separable window (all the
// windows of a kernel in one horizontal and one vertical sweep: two barriers):
//   ssim_map_kernel   mu1, mu2, E[x1^2], E[x2^2], E[x1 x2] per pixel -> the SSIM value of the pixel (summed per workgroup, fixed order ->
//                     partial[block]) and the three partial derivatives of it that the backward needs:
//                         m = A1 A2 / (B1 B2),  A1 = 2 mu1 mu2 + C1,  A2 = 2 s12 + C2,  B1 = mu1^2 + mu2^2 + C1,  B2 = s1 + s2 + C2,
//                         s1 = E[x1^2] - mu1^2,  s12 = E[x1 x2] - mu1 mu2
//                         dm/dmu1 (at fixed E[.]),  dm/dE[x1^2] = -m / B2,  dm/dE[x1 x2] = 2 A1 / (B1 B2)
//   ssim_grad_kernel  dL/dx1(q) = scale * ( conv(dm/dmu1)(q) + 2 x1(q) conv(dm/dE11)(q) + x2(q) conv(dm/dE12)(q) )   (the window is
//                     symmetric: correlation = convolution), scale = -weight / (3 H W); written to, or added onto, dL_dimg.
// The per-workgroup partial sums are added up by one workgroup in a fixed order (ssim_finish_kernel): reproducible.
#include "dqo_common.h"

namespace {

constexpr int SS_T = 16, SS_R = 5, SS_W = SS_T + 2 * SS_R;  // tile, window radius, halo tile

struct SsimWin {
    float g[11];
};

__global__ __launch_bounds__(SS_T* SS_T) void ssim_map_kernel(int W, int H, const float* __restrict__ img1, const float* __restrict__ img2,
                                                             SsimWin w, float* __restrict__ maps, float* __restrict__ partial) {
    __shared__ float s_a[SS_W][SS_W + 1], s_b[SS_W][SS_W + 1];
    __shared__ float s_h5[5][SS_W][SS_T + 1];  // horizontal sums of x1, x2, x1^2, x2^2, x1 x2
    __shared__ float s_red[SS_T * SS_T / 64];
    const int tid = threadIdx.x, tx = tid & (SS_T - 1), ty = tid / SS_T;
    const int ch = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const size_t HW = (size_t)W * H;
    const float* p1 = img1 + (size_t)ch * HW;
    const float* p2 = img2 + (size_t)ch * HW;
    for (int i = tid; i < SS_W * SS_W; i += SS_T * SS_T) {  // zero padding outside the image (conv2d padding = 5)
        const int r = i / SS_W, c = i - r * SS_W;
        const int x = x0 + c - SS_R, y = y0 + r - SS_R;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
        s_a[r][c] = in ? p1[(size_t)y * W + x] : 0.f;
        s_b[r][c] = in ? p2[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    // the five windows in one sweep: every LDS value is read once per tap and feeds all five sums
    for (int i = tid; i < SS_W * SS_T; i += SS_T * SS_T) {
        const int r = i / SS_T, c = i - r * SS_T;
        float h1 = 0.f, h2 = 0.f, h11 = 0.f, h22 = 0.f, h12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float a = s_a[r][c + k], b = s_b[r][c + k], g = w.g[k];
            h1 += g * a, h2 += g * b, h11 += g * (a * a), h22 += g * (b * b), h12 += g * (a * b);
        }
        s_h5[0][r][c] = h1, s_h5[1][r][c] = h2, s_h5[2][r][c] = h11, s_h5[3][r][c] = h22, s_h5[4][r][c] = h12;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float g = w.g[k];
        mu1 += g * s_h5[0][ty + k][tx], mu2 += g * s_h5[1][ty + k][tx], e11 += g * s_h5[2][ty + k][tx], e22 += g * s_h5[3][ty + k][tx],
            e12 += g * s_h5[4][ty + k][tx];
    }
    const int x = x0 + tx, y = y0 + ty;
    const bool in = x < W && y < H;
    float m = 0.f;
    if (in) {
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1s = mu1 * mu1, mu2s = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1s, s2 = e22 - mu2s, s12 = e12 - mu12;
        const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1s + mu2s + C1, B2 = s1 + s2 + C2;
        const float inv = 1.f / (B1 * B2);
        m = A1 * A2 * inv;
        // dm/dmu1 at fixed E[.]: mu1 enters A1 (2 mu2), A2 (-2 mu2 through s12), B1 (2 mu1) and B2 (-2 mu1 through s1)
        const float dmu1 = (2.f * mu2 * A2 - 2.f * mu2 * A1) * inv - m * (2.f * mu1 / B1 - 2.f * mu1 / B2);
        const float de11 = -m / B2;
        const float de12 = 2.f * A1 * inv;
        const size_t o = (size_t)ch * HW + (size_t)y * W + x;
        maps[o] = dmu1, maps[3 * HW + o] = de11, maps[6 * HW + o] = de12;
    }
    // the workgroup's sum of the pixel values, in a fixed order
    float v = m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < SS_T * SS_T / 64; i++) t += s_red[i];
        partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void ssim_finish_kernel(int n, const float* __restrict__ partial, double inv_count, float weight,
                                                           float* __restrict__ out, float* __restrict__ loss8) {
    // one workgroup, a fixed summation tree (strided per thread -> wave -> workgroup): the same bits at every call
    __shared__ double s_w[16];
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) t += (double)partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 16; i++) tot += s_w[i];
        const float ssim = (float)(tot * inv_count);
        out[0] = ssim;                   // ssim(image, gt)
        out[1] = weight * (1.f - ssim);  // the loss term
        if (loss8 != nullptr) {          // dqo_map_loss_fwd_bwd's loss_out of the same iteration: total += term, slot 3 = 1 - ssim
            loss8[0] += weight * (1.f - ssim);
            loss8[3] = 1.f - ssim;
        }
    }
}

__global__ __launch_bounds__(SS_T* SS_T) void ssim_grad_kernel(int W, int H, const float* __restrict__ img1, const float* __restrict__ img2,
                                                              SsimWin w, const float* __restrict__ maps, float scale, int accumulate,
                                                              float* __restrict__ dL_dimg) {
    __shared__ float s_m[3][SS_W][SS_W + 1];
    __shared__ float s_h3[3][SS_W][SS_T + 1];
    const int tid = threadIdx.x, tx = tid & (SS_T - 1), ty = tid / SS_T;
    const int ch = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const size_t HW = (size_t)W * H;
    for (int i = tid; i < SS_W * SS_W; i += SS_T * SS_T) {  // (the SSIM map only exists on image pixels: zero outside)
        const int r = i / SS_W, cc = i - r * SS_W;
        const int x = x0 + cc - SS_R, y = y0 + r - SS_R;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
        const size_t o = (size_t)ch * HW + (size_t)(in ? y : 0) * W + (in ? x : 0);
#pragma unroll
        for (int k = 0; k < 3; k++) s_m[k][r][cc] = in ? maps[(size_t)3 * k * HW + o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < SS_W * SS_T; i += SS_T * SS_T) {
        const int r = i / SS_T, cc = i - r * SS_T;
        float h0 = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = w.g[k];
            h0 += g * s_m[0][r][cc + k], h1 += g * s_m[1][r][cc + k], h2 += g * s_m[2][r][cc + k];
        }
        s_h3[0][r][cc] = h0, s_h3[1][r][cc] = h1, s_h3[2][r][cc] = h2;
    }
    __syncthreads();
    float c[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float g = w.g[k];
        c[0] += g * s_h3[0][ty + k][tx], c[1] += g * s_h3[1][ty + k][tx], c[2] += g * s_h3[2][ty + k][tx];
    }
    const int x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
        const size_t o = (size_t)ch * HW + (size_t)y * W + x;
        const float g = scale * (c[0] + 2.f * img1[o] * c[1] + img2[o] * c[2]);
        dL_dimg[o] = accumulate ? dL_dimg[o] + g : g;
    }
}

}  // namespace

size_t dqo_map_ssim_ws_bytes(int W, int H) {
    const size_t HW = (size_t)W * H;
    const size_t blocks = (size_t)((W + SS_T - 1) / SS_T) * ((H + SS_T - 1) / SS_T) * 3;
    return dqo_align_up(sizeof(float) * 9 * HW, 256) + dqo_align_up(sizeof(float) * blocks, 256);
}

int dqo_launch_map_ssim(int W, int H, const float* img, const float* gt, float weight, float* ssim_out, float* dL_dimg, int accumulate,
                        float* loss8, void* ws, hipStream_t s) {
    const size_t HW = (size_t)W * H;
    float* maps = reinterpret_cast<float*>(ws);
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + dqo_align_up(sizeof(float) * 9 * HW, 256));
    // utils/loss_utils.py:41-58: float32(exp(-(x - 5)^2 / (2 sigma^2))) / their float32 sum.  The SSIM value is sensitive to the LAST BIT of
    // that sum: a window summing to 1 + e shifts sigma^2 = E[x^2] - mu^2 by -e mu^2, amplified by mu^2 / (sigma^2 + C2) ~ 10^2 on smooth
    // images (a sequential float sum, one ulp below torch's, moved ssim by 5e-6).  torch's float sum of these eleven values is the
    // correctly rounded one: formed here in double and rounded once -> weights bit-identical to the reference's window.
    SsimWin w;
    double sum = 0.0;
    for (int i = 0; i < 11; i++) {
        w.g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
        sum += (double)w.g[i];
    }
    const float sum_f = (float)sum;
    for (int i = 0; i < 11; i++) w.g[i] /= sum_f;
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, 3), block(SS_T * SS_T);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    DQO_LAUNCH("ssim_map_kernel", ssim_map_kernel, grid, block, s, W, H, img, gt, w, maps, partial);
    DQO_LAUNCH("ssim_finish_kernel", ssim_finish_kernel, dim3(1), dim3(1024), s, nblocks, partial, 1.0 / (3.0 * (double)HW), weight, ssim_out, loss8);
    if (dL_dimg != nullptr) {
        const float scale = (float)(-(double)weight / (3.0 * (double)HW));
        DQO_LAUNCH("ssim_grad_kernel", ssim_grad_kernel, grid, block, s, W, H, img, gt, w, maps, scale, accumulate, dL_dimg);
    }
    return DQO_OK;
}
