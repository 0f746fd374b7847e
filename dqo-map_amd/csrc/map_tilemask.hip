// Tile-mask producers of the mapping loop for gfx950.
//
// Replaces the torch pooling chains of /root/reference/SLAM/utils.py:
//   pixelmask2tilemask :731-743 (pad + max_pool2d),  transmission2tilemask :752-763 (pad + avg_pool2d + threshold),
//   meanpool :720-729 / colorerror2tilemask :766-799 (pad + avg_pool2d; the top-k selection stays in the caller),
// and the mask / colour-error images of evaluate_render_range, SLAM/multiprocess/mapper.py:930-988
//   (render_mask = T_map != 1;  color_error = sum_c |render - gt| with pixels whose rendered colour sums to 0 zeroed).
// The reference pads to a multiple of the stride with zeros and pools with count_include_pad: every 16x16 tile is divided
// by 256 whatever part of it lies inside the image.  One 256-thread block per tile, one pixel per thread: each image is read
// once, coalesced (64-byte rows), and reduced with a wave ballot / DPP sum + one LDS hop — the reference runs 4-7 eager
// kernels per mask and materialises the padded copies.
#include "dqo_common.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// mode 0: mask_in != 0 (uint8 pixel mask).  mode 1: T_map != 1 (also written to mask_out).
__global__ __launch_bounds__(256) void tile_count_kernel(int W, int H, int gx, int mode, const uint8_t* __restrict__ mask_in,
                                                         const float* __restrict__ T_map, uint8_t* __restrict__ mask_out,
                                                         int32_t* __restrict__ tile_count, int32_t* __restrict__ total) {
    __shared__ int s_cnt[4];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int px = (tile % gx) * DQO_TILE + (tid & 15), py = (tile / gx) * DQO_TILE + (tid >> 4);
    bool m = false;
    if (px < W && py < H) {
        const size_t pid = (size_t)py * W + px;
        if (mode == 0) {
            m = mask_in[pid] != 0;
        } else {
            m = T_map[pid] != 1.0f;
            if (mask_out) mask_out[pid] = m ? 1 : 0;
        }
    }
    const int c = (int)__popcll(__builtin_amdgcn_ballot_w64(m));
    if ((tid & 63) == 0) s_cnt[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        tile_count[tile] = n;
        if (total && n) atomicAdd(total, n);
    }
}

__global__ __launch_bounds__(256) void tile_color_error_kernel(int W, int H, int gx, const float* __restrict__ render,
                                                               const float* __restrict__ gt, float* __restrict__ err_px,
                                                               float* __restrict__ tile_sum) {
#pragma clang fp contract(off)
    __shared__ float s_sum[4];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int px = (tile % gx) * DQO_TILE + (tid & 15), py = (tile / gx) * DQO_TILE + (tid >> 4);
    const size_t HW = (size_t)W * H;
    float e = 0.f;
    if (px < W && py < H) {
        const size_t pid = (size_t)py * W + px;
        const float r0 = render[pid], r1 = render[HW + pid], r2 = render[2 * HW + pid];
        e = (fabsf(r0 - gt[pid]) + fabsf(r1 - gt[HW + pid])) + fabsf(r2 - gt[2 * HW + pid]);  // torch.sum over the last dim
        if ((r0 + r1) + r2 == 0.f) e = 0.f;                                                   // mapper.py:955-956
        if (err_px) err_px[pid] = e;
    }
    const float s = wave_sum_f(e);
    if ((tid & 63) == 0) s_sum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) tile_sum[tile] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
}

}  // namespace

int dqo_launch_tile_count(int W, int H, int mode, const uint8_t* mask_in, const float* T_map, uint8_t* mask_out, int32_t* tile_count,
                          int32_t* total, hipStream_t s) {
    const int gx = (W + DQO_TILE - 1) / DQO_TILE, gy = (H + DQO_TILE - 1) / DQO_TILE;
    if (total) {
        int rc = dqo_launch_zero_words(reinterpret_cast<uint32_t*>(total), 1, s);
        if (rc) return rc;
    }
    DQO_LAUNCH("tile_count_kernel", tile_count_kernel, dim3(gx * gy), dim3(256), s, W, H, gx, mode, mask_in, T_map, mask_out, tile_count,
               total);
    return DQO_OK;
}

int dqo_launch_tile_color_error(int W, int H, const float* render, const float* gt, float* err_px, float* tile_sum, hipStream_t s) {
    const int gx = (W + DQO_TILE - 1) / DQO_TILE, gy = (H + DQO_TILE - 1) / DQO_TILE;
    DQO_LAUNCH("tile_color_error_kernel", tile_color_error_kernel, dim3(gx * gy), dim3(256), s, W, H, gx, render, gt, err_px, tile_sum);
    return DQO_OK;
}
