// Per-Gaussian error accumulation for gfx950 — replaces
// /root/reference/submodules/cuda_utils/map_process.cu:33-245 (acuumulate_error_preprocessCUDA, accumulate_error_meanCUDA) and
// cuda_utils.cu:17-62 (accumulate_gaussian_error): scatter the per-pixel colour / depth / normal errors onto the Gaussians
// named by the rasteriser's hit-index maps (max or mean), and count per Gaussian how many pixels exceed the thresholds; and
// map_process.cu:247-360 (accumulate_gaussian_confidence), further down.
//
// The reference implements float atomicMax as a compare-and-swap loop.  The accumulators start at 0 and only values that
// compare greater replace them, so only strictly positive floats ever win — and for positive floats the IEEE bit pattern
// orders like the value: one hardware integer atomic max per pixel instead of a CAS loop, bit-identical result.
#include "dqo_common.h"

namespace {

__device__ __forceinline__ void atomic_max_pos(float* addr, float val) {
    if (val > 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(val));  // NaN and values <= 0 never beat the 0 init
}

__global__ __launch_bounds__(256) void accumulate_error_kernel(int HW, int P, const float* __restrict__ color_err,
                                                               const float* __restrict__ depth_err, const float* __restrict__ normal_err,
                                                               const int32_t* __restrict__ color_index, const int32_t* __restrict__ depth_index,
                                                               float color_thr, float depth_thr, float normal_thr, int check_max,
                                                               float* __restrict__ gs_color, float* __restrict__ gs_depth,
                                                               float* __restrict__ gs_normal, float* __restrict__ rescale,
                                                               int32_t* __restrict__ counters /* [3][P], mean mode only */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float ce = color_err[i], de = depth_err[i], ne = normal_err[i];
    const int ci = color_index[i], di = depth_index[i];
    if (ci >= 0 && ci < P) {
        if (check_max) atomic_max_pos(&gs_color[ci], ce);
        else {
            atomicAdd(&gs_color[ci], ce);
            atomicAdd(&counters[ci], 1);
        }
        if (ce > color_thr) atomicAdd(&rescale[ci], 1.0f);
    }
    if (di >= 0 && di < P) {
        if (check_max) {
            atomic_max_pos(&gs_depth[di], de);
            atomic_max_pos(&gs_normal[di], ne);
        } else {
            atomicAdd(&gs_depth[di], de);
            atomicAdd(&gs_normal[di], ne);
            atomicAdd(&counters[P + di], 1);
        }
        if (de > depth_thr) atomicAdd(&rescale[di], 1.0f);
        if (ne > normal_thr) atomicAdd(&rescale[di], 1.0f);
    }
}

__global__ void error_mean_kernel(int P, const int32_t* __restrict__ counters, float* __restrict__ gs_color, float* __restrict__ gs_depth,
                                  float* __restrict__ gs_normal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int cc = counters[i], dc = counters[P + i];
    if (cc > 0) gs_color[i] = gs_color[i] / cc;
    if (dc > 0) {
        gs_depth[i] = gs_depth[i] / dc;
        gs_normal[i] = gs_normal[i] / dc;  // normal_error_counter == depth_error_counter (map_process.cu:103-104)
    }
}

// ---- accumulate_gaussian_confidence (map_process.cu:247-360, cuda_utils.cu:62-83) --------------------------------------------
// Per Gaussian named by an index map: count, sum, maximum and minimum of a per-pixel confidence; afterwards mean = sum / count, and
// Gaussians that no pixel named get 0 in all three.  The reference's float atomicMax / atomicMin are compare-and-swap loops that
// replace only on a strict comparison (NaN never wins).  Here: one hardware integer atomic per value — non-negative floats order like
// their bit patterns as signed integers, negative floats in reverse as unsigned integers, and the two cases compose because the
// accumulators start at -FLT_MAX / +FLT_MAX (cuda_utils.cu:71-72).
// (-0.0f is folded into +0.0f first: as an integer it is INT_MIN, which the signed minimum would take for the smallest value of all)
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v != v) return;
    v += 0.0f;
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
    if (v != v) return;
    v += 0.0f;
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void confidence_init_kernel(int P, int32_t* __restrict__ counter, float* __restrict__ gmax, float* __restrict__ gmin,
                                       float* __restrict__ gmean) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    counter[i] = 0, gmean[i] = 0.f, gmax[i] = -3.402823466e+38f, gmin[i] = 3.402823466e+38f;
}

__global__ __launch_bounds__(256) void accumulate_confidence_kernel(int HW, int P, const float* __restrict__ confidence,
                                                                    const int32_t* __restrict__ index, int32_t* __restrict__ counter,
                                                                    float* __restrict__ gmax, float* __restrict__ gmin,
                                                                    float* __restrict__ gmean) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const int gi = index[i];
    if (gi < 0 || gi >= P) return;
    const float c = confidence[i];
    atomicAdd(&counter[gi], 1);
    atomicAdd(&gmean[gi], c);
    atomic_max_f32(&gmax[gi], c);
    atomic_min_f32(&gmin[gi], c);
}

__global__ void confidence_mean_kernel(int P, const int32_t* __restrict__ counter, float* __restrict__ gmax, float* __restrict__ gmin,
                                       float* __restrict__ gmean) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int n = counter[i];
    if (n > 0) gmean[i] = gmean[i] / n;
    else gmean[i] = 0.f, gmax[i] = 0.f, gmin[i] = 0.f;
}

}  // namespace

int dqo_launch_accumulate_confidence(int H, int W, int P, const int32_t* index, const float* confidence, float* gmax, float* gmin,
                                     float* gmean, int32_t* counter, hipStream_t s) {
    const int HW = H * W;
    DQO_LAUNCH("confidence_init_kernel", confidence_init_kernel, dim3((P + 255) / 256), dim3(256), s, P, counter, gmax, gmin, gmean);
    DQO_LAUNCH("accumulate_confidence_kernel", accumulate_confidence_kernel, dim3((HW + 255) / 256), dim3(256), s, HW, P, confidence,
               index, counter, gmax, gmin, gmean);
    DQO_LAUNCH("confidence_mean_kernel", confidence_mean_kernel, dim3((P + 255) / 256), dim3(256), s, P, counter, gmax, gmin, gmean);
    return DQO_OK;
}

int dqo_launch_accumulate_error(int H, int W, int P, const float* color_err, const float* depth_err, const float* normal_err,
                                const int32_t* color_index, const int32_t* depth_index, float color_thr, float depth_thr,
                                float normal_thr, int check_max, float* gs_color, float* gs_depth, float* gs_normal, float* rescale,
                                int32_t* counters, hipStream_t s) {
    for (float* q : {gs_color, gs_depth, gs_normal, rescale}) {
        int rc = dqo_launch_zero_words(reinterpret_cast<uint32_t*>(q), (size_t)P, s);
        if (rc) return rc;
    }
    if (!check_max) {
        int rc = dqo_launch_zero_words(reinterpret_cast<uint32_t*>(counters), 2 * (size_t)P, s);
        if (rc) return rc;
    }
    const int HW = H * W;
    DQO_LAUNCH("accumulate_error_kernel", accumulate_error_kernel, dim3((HW + 255) / 256), dim3(256), s, HW, P, color_err, depth_err,
               normal_err, color_index, depth_index, color_thr, depth_thr, normal_thr, check_max, gs_color, gs_depth, gs_normal, rescale,
               counters);
    if (!check_max) {
        DQO_LAUNCH("error_mean_kernel", error_mean_kernel, dim3((P + 255) / 256), dim3(256), s, P, counters, gs_color, gs_depth, gs_normal);
    }
    return DQO_OK;
}
