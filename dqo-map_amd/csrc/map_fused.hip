// Fused kernels around the rasteriser for one mapping iteration ("next" tier).
//
// Replaces, for the masked-loss case that DQO-MAP's optimiser loop runs (SLAM/multiprocess/mapper.py:531-605):
//   * the activation ops of SLAM/gaussian_pointcloud.py:732-733, 746-747, 815-822 (exp / sigmoid / normalize)  -> activate_kernel
//   * the loss of SLAM/multiprocess/mapper.py:836-875 (masked L1 colour + masked depth L1; SSIM is skipped when a render mask
//     is given, quirk B14) and its autograd backward                                                          -> loss_*_kernel
//   * autograd through the activations + torch.optim.Adam over the six parameter groups
//     (gaussian_pointcloud.py:331-378; mapper.py:548 Adam(lr=0, eps=1e-15))                                   -> adam_kernel
// The reference runs these as ~110 eager launches per iteration (0.4 ms of Adam + ~0.75 ms of tiny kernels on MI355X);
// here they are 4 launches and every tensor crosses HBM once: the loss never materialises |x|, masks or index tensors,
// the optimiser reads the gradient w.r.t. the ACTIVATED parameters straight from the rasteriser's backward and applies
// the activation Jacobians in registers.
#include "dqo_adam.h"
#include "dqo_common.h"

namespace {

// ---------------------------------------------------------------- activations ----------------------------------------
__global__ __launch_bounds__(256) void activate_kernel(int P, const float* __restrict__ opacity_raw, const float* __restrict__ scaling_raw,
                                                       const float* __restrict__ rotation_raw, float* __restrict__ opacity,
                                                       float* __restrict__ scales, float* __restrict__ rotations) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    opacity[i] = 1.0f / (1.0f + expf(-opacity_raw[i]));  // torch.sigmoid
#pragma unroll
    for (int k = 0; k < 3; k++) scales[3 * i + k] = expf(scaling_raw[3 * i + k]);  // torch.exp
    const float4 q = reinterpret_cast<const float4*>(rotation_raw)[i];
    const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);  // F.normalize(eps=1e-12)
    reinterpret_cast<float4*>(rotations)[i] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
}

// ---------------------------------------------------------------- loss ------------------------------------------------
// pass 1: masked sums and counts; pass 2: losses + gradients.  acc = {sum|dc|, n_color_px, sum|dd|, n_depth_px}
constexpr int LOSS_THREADS = 256;

__device__ __forceinline__ float wave_red(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Both loss kernels stream eight image planes.  V = 4: four pixels per thread and trip through 16-byte loads (planes 16-byte
// aligned, H*W a multiple of 4), every load of a trip issued before the first use and none under a lane condition (the mask only
// selects); V = 1: the general path.
template <int V>
struct LossPix {
    float c[3][V], g[3][V], d[V], gd[V];
    int di[V];
    bool m[V];
};
template <int V>
__device__ __forceinline__ LossPix<V> loss_load(int i, int n, const float* color, const float* depth, const int32_t* depth_index,
                                                const float* gt_color, const float* gt_depth, const uint8_t* render_mask) {
    LossPix<V> p;
    if (V == 4) {
        const float4* C = reinterpret_cast<const float4*>(color);
        const float4* G = reinterpret_cast<const float4*>(gt_color);
        float4 cc[3], gg[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) cc[ch] = C[(size_t)ch * n + i], gg[ch] = G[(size_t)ch * n + i];
        const float4 dd = reinterpret_cast<const float4*>(depth)[i], gdd = reinterpret_cast<const float4*>(gt_depth)[i];
        const int4 ii = reinterpret_cast<const int4*>(depth_index)[i];
        const uint32_t mk = render_mask ? reinterpret_cast<const uint32_t*>(render_mask)[i] : 0x01010101u;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            p.c[ch][0] = cc[ch].x, p.c[ch][1] = cc[ch].y, p.c[ch][2] = cc[ch].z, p.c[ch][3] = cc[ch].w;
            p.g[ch][0] = gg[ch].x, p.g[ch][1] = gg[ch].y, p.g[ch][2] = gg[ch].z, p.g[ch][3] = gg[ch].w;
        }
        p.d[0] = dd.x, p.d[1] = dd.y, p.d[2] = dd.z, p.d[3] = dd.w;
        p.gd[0] = gdd.x, p.gd[1] = gdd.y, p.gd[2] = gdd.z, p.gd[3] = gdd.w;
        p.di[0] = ii.x, p.di[1] = ii.y, p.di[2] = ii.z, p.di[3] = ii.w;
#pragma unroll
        for (int k = 0; k < 4; k++) p.m[k] = ((mk >> (8 * k)) & 0xffu) != 0u;
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) p.c[ch][0] = color[(size_t)ch * n + i], p.g[ch][0] = gt_color[(size_t)ch * n + i];
        p.d[0] = depth[i], p.gd[0] = gt_depth[i], p.di[0] = depth_index[i];
        const uint8_t mk = render_mask ? render_mask[i] : (uint8_t)1;
        p.m[0] = mk != 0;
    }
    return p;
}

template <int V>
__global__ __launch_bounds__(LOSS_THREADS) void loss_reduce_kernel(int HW, const float* __restrict__ color, const float* __restrict__ depth,
                                                                   const int32_t* __restrict__ depth_index,
                                                                   const float* __restrict__ gt_color, const float* __restrict__ gt_depth,
                                                                   const uint8_t* __restrict__ render_mask, float add_depth_thres,
                                                                   double* __restrict__ partial /*[gridDim.x][4]*/) {
    __shared__ float s[4][LOSS_THREADS / 64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int n = HW / V;  // units of V pixels
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const LossPix<V> p = loss_load<V>(i, n, color, depth, depth_index, gt_color, gt_depth, render_mask);
#pragma unroll
        for (int k = 0; k < V; k++) {
            const float e = fabsf(p.c[0][k] - p.g[0][k]) + fabsf(p.c[1][k] - p.g[1][k]) + fabsf(p.c[2][k] - p.g[2][k]);
            a0 += p.m[k] ? e : 0.f;
            a1 += p.m[k] ? 1.f : 0.f;
            const float err = p.d[k] - p.gd[k];
            const bool valid = p.m[k] && p.di[k] != -1 && p.gd[k] > 0.f && err < add_depth_thres;  // mapper.py:850-856
            a2 += valid ? fabsf(err) : 0.f;
            a3 += valid ? 1.f : 0.f;
        }
    }
    a0 = wave_red(a0), a1 = wave_red(a1), a2 = wave_red(a2), a3 = wave_red(a3);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s[0][wave] = a0, s[1][wave] = a1, s[2][wave] = a2, s[3][wave] = a3;
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int w = 0; w < LOSS_THREADS / 64; w++) t += (double)s[threadIdx.x][w];
        partial[4 * blockIdx.x + threadIdx.x] = t;
    }
}

template <int V>
__global__ __launch_bounds__(LOSS_THREADS) void loss_grad_kernel(int HW, int n_partial, const float* __restrict__ color,
                                                                 const float* __restrict__ depth, const int32_t* __restrict__ depth_index,
                                                                 const float* __restrict__ gt_color, const float* __restrict__ gt_depth,
                                                                 const uint8_t* __restrict__ render_mask, float add_depth_thres,
                                                                 float color_weight, float depth_weight,
                                                                 const double* __restrict__ partial, float* __restrict__ loss_out,
                                                                 float* __restrict__ dL_dcolor, float* __restrict__ dL_ddepth) {
    // every block re-derives the four totals from the per-block partials: parallel, fixed order (reproducible)
    __shared__ double s_tot[4];
    __shared__ double s_w[4][LOSS_THREADS / 64];
    {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = threadIdx.x; b < n_partial; b += LOSS_THREADS)
#pragma unroll
            for (int c = 0; c < 4; c++) t[c] += partial[4 * b + c];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t[c] += __shfl_xor(t[c], off);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0)
#pragma unroll
            for (int c = 0; c < 4; c++) s_w[c][wave] = t[c];
        __syncthreads();
        if (threadIdx.x < 4) {
            double tt = 0.0;
            for (int w = 0; w < LOSS_THREADS / 64; w++) tt += s_w[threadIdx.x][w];
            s_tot[threadIdx.x] = tt;
        }
    }
    __syncthreads();
    const float n_col = fmaxf((float)s_tot[1], 1.f), n_dep = fmaxf((float)s_tot[3], 1.f);
    const float color_loss = (float)(s_tot[0] / (3.0 * (double)n_col));
    const float depth_loss = (float)(s_tot[2] / (double)n_dep);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        loss_out[0] = depth_weight * depth_loss + color_weight * color_loss;  // mapper.py:870-875 (normal / ssim terms are 0 here)
        loss_out[1] = color_loss;
        loss_out[2] = depth_loss;
        loss_out[3] = 0.f;
        // the unnormalised sums: what shards of one map exchange (one packed all-reduce) to report the loss over ALL objects
        loss_out[4] = (float)s_tot[0], loss_out[5] = (float)s_tot[1], loss_out[6] = (float)s_tot[2], loss_out[7] = (float)s_tot[3];
    }
    const float gc = color_weight / (3.f * n_col), gdw = depth_weight / n_dep;
    const int n = HW / V;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const LossPix<V> p = loss_load<V>(i, n, color, depth, depth_index, gt_color, gt_depth, render_mask);
        float oc[3][V], od[V];
#pragma unroll
        for (int k = 0; k < V; k++) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float d = p.c[ch][k] - p.g[ch][k];
                oc[ch][k] = p.m[k] ? (d > 0.f ? gc : (d < 0.f ? -gc : 0.f)) : 0.f;  // d|x|/dx = sign(x), 0 at 0
            }
            const float err = p.d[k] - p.gd[k];
            const bool valid = p.m[k] && p.di[k] != -1 && p.gd[k] > 0.f && err < add_depth_thres;
            od[k] = valid ? (err > 0.f ? gdw : (err < 0.f ? -gdw : 0.f)) : 0.f;
        }
        if (V == 4) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                reinterpret_cast<float4*>(dL_dcolor)[(size_t)ch * n + i] = make_float4(oc[ch][0], oc[ch][1], oc[ch][2], oc[ch][3]);
            reinterpret_cast<float4*>(dL_ddepth)[i] = make_float4(od[0], od[1], od[2], od[3]);
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) dL_dcolor[(size_t)ch * n + i] = oc[ch][0];
            dL_ddepth[i] = od[0];
        }
    }
}

// ---------------------------------------------------------------- attach loss ----------------------------------------
// dqo_map_attach_loss_fwd_bwd: the attach loss of Mapping.loss_update (mapper.py:812-829) and its gradient in one pass — for callers
// that keep their own optimiser (the drop-in path): ~12 eager launches forward + ~25 backward as two.
constexpr int ATT_THREADS = 256;
__global__ __launch_bounds__(ATT_THREADS) void attach_loss_kernel(int P, const float* __restrict__ scaling, const float* __restrict__ xyz,
                                                                  const float* __restrict__ rotation, const float* __restrict__ scaling0,
                                                                  const float* __restrict__ xyz0, const float* __restrict__ rotation0,
                                                                  const uint8_t* __restrict__ mask, float g3, float g4,
                                                                  float* __restrict__ g_scaling, float* __restrict__ g_xyz,
                                                                  float* __restrict__ g_rotation, double* __restrict__ partial) {
    __shared__ double s_w[ATT_THREADS / 64];
    const int i = blockIdx.x * ATT_THREADS + threadIdx.x;
    double acc = 0.0;
    if (i < P) {
        const bool a = mask[i] != 0;
        float ds[3], dx[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            ds[k] = a ? scaling[3 * i + k] - scaling0[3 * i + k] : 0.f;
            dx[k] = a ? xyz[3 * i + k] - xyz0[3 * i + k] : 0.f;
            g_scaling[3 * i + k] = g3 * ds[k];
            g_xyz[3 * i + k] = g3 * dx[k];
        }
        const float4 q = reinterpret_cast<const float4*>(rotation)[i], q0 = reinterpret_cast<const float4*>(rotation0)[i];
        const float4 d = a ? make_float4(q.x - q0.x, q.y - q0.y, q.z - q0.z, q.w - q0.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(g_rotation)[i] = make_float4(g4 * d.x, g4 * d.y, g4 * d.z, g4 * d.w);
        acc = 0.5 * (double)g3 * ((double)ds[0] * ds[0] + (double)ds[1] * ds[1] + (double)ds[2] * ds[2] + (double)dx[0] * dx[0] +
                                  (double)dx[1] * dx[1] + (double)dx[2] * dx[2]) +
              0.5 * (double)g4 * ((double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}
__global__ __launch_bounds__(ATT_THREADS) void attach_loss_sum_kernel(int n, const double* __restrict__ partial, float* __restrict__ loss) {
    __shared__ double s_w[ATT_THREADS / 64];
    double acc = 0.0;
    for (int b = threadIdx.x; b < n; b += ATT_THREADS) acc += partial[b];  // fixed order: reproducible
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

// ---------------------------------------------------------------- history merge ---------------------------------------
// Mapping.history_merge (SLAM/multiprocess/mapper.py:607-650) + slerp (SLAM/utils.py:650-709) in one launch: a block owns HM_THREADS
// Gaussians — a thread per Gaussian for xyz / scaling / rotation, then the block's SH rows element by element (coalesced).  The
// reference's eager ops are separate IEEE operations: contraction off, same statement order (the lerps come out bit for bit; the
// slerp's acos / sin differ from torch's CPU libm in the last bits).
constexpr int HM_THREADS = 256;
__device__ __forceinline__ float hm_weight(const float max_weight, const float c0, const float c) {
#pragma clang fp contract(off)
    return max_weight * c0 / (c + 0.000001f);  // mapper.py:610-614
}
// torch.lerp(start, end, weight): aten's lerp — weight < 0.5 ? start + weight * (end - start) : end - (end - start) * (1 - weight)
__device__ __forceinline__ float hm_torch_lerp(const float a, const float b, const float w) {
#pragma clang fp contract(off)
    const float d = b - a;
    return fabsf(w) < 0.5f ? a + w * d : b - d * (1.f - w);
}
__global__ __launch_bounds__(HM_THREADS) void history_merge_kernel(const int P, const int M, const float max_weight, const int first_row,
                                                                   const uint8_t* __restrict__ row_flags, const float* __restrict__ conf0,
                                                                   const float* __restrict__ conf, const float* __restrict__ xyz0,
                                                                   const float* __restrict__ shs0, const float* __restrict__ scaling0,
                                                                   const float* __restrict__ rot0, float* __restrict__ xyz,
                                                                   float* __restrict__ shs, float* __restrict__ scaling,
                                                                   float* __restrict__ rotation) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * HM_THREADS + threadIdx.x;
    // `history_weight[0]` (mapper.py:620-637): ONE weight — the first trained row's — for f_dc, f_rest and scaling of every row
    const float w0 = hm_weight(max_weight, conf0[first_row], conf[first_row]);
    const float u0 = 1.f - w0;
    if (i < P && !(row_flags != nullptr && (row_flags[i] & DQO_ROW_FROZEN) != 0u)) {
        const float w = hm_weight(max_weight, conf0[i], conf[i]);
        const float u = 1.f - w;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            xyz[3 * i + k] = xyz0[3 * i + k] * w + u * xyz[3 * i + k];                       // :615-618
            scaling[3 * i + k] = scaling0[3 * i + k] * w0 + u0 * scaling[3 * i + k];         // :630-633
        }
        // slerp(history rotation, get_rotation, 1 - w), SLAM/utils.py:650-709; get_rotation = F.normalize(_rotation)
        const float4 v0 = reinterpret_cast<const float4*>(rot0)[i];
        const float4 qr = reinterpret_cast<const float4*>(rotation)[i];
        const float nq = fmaxf(sqrtf(qr.x * qr.x + qr.y * qr.y + qr.z * qr.z + qr.w * qr.w), 1e-12f);
        const float4 v1 = make_float4(qr.x / nq, qr.y / nq, qr.z / nq, qr.w / nq);
        const float n0 = sqrtf(v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w);
        const float n1 = sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w);
        const float dot = (v0.x / n0) * (v1.x / n1) + (v0.y / n0) * (v1.y / n1) + (v0.z / n0) * (v1.z / n1) + (v0.w / n0) * (v1.w / n1);
        const float t = u;  // 1 - history_weight
        float4 o;
        if (!(fabsf(dot) <= 0.9995f)) {  // NaN or nearly colinear -> torch.lerp(v0, v1, t), :677-691
            o = make_float4(hm_torch_lerp(v0.x, v1.x, t), hm_torch_lerp(v0.y, v1.y, t), hm_torch_lerp(v0.z, v1.z, t), hm_torch_lerp(v0.w, v1.w, t));
        } else {  // :694-707
            const float th0 = acosf(dot), s_th0 = sinf(th0), tht = th0 * t;
            const float s0 = sinf(th0 - tht) / s_th0, s1 = sinf(tht) / s_th0;
            o = make_float4(s0 * v0.x + s1 * v1.x, s0 * v0.y + s1 * v1.y, s0 * v0.z + s1 * v1.z, s0 * v0.w + s1 * v1.w);
        }
        reinterpret_cast<float4*>(rotation)[i] = o;
    }
    // f_dc / f_rest (:620-628): rows of 3 M floats, the block's HM_THREADS rows as one run of elements
    const int row = 3 * M;
    const size_t e0 = (size_t)blockIdx.x * HM_THREADS * row;
    const int n = min(HM_THREADS, P - blockIdx.x * HM_THREADS) * row;
    for (int e = threadIdx.x; e < n; e += HM_THREADS) {
        const int r = blockIdx.x * HM_THREADS + e / row;
        if (row_flags != nullptr && (row_flags[r] & DQO_ROW_FROZEN) != 0u) continue;
        shs[e0 + e] = shs0[e0 + e] * w0 + u0 * shs[e0 + e];
    }
}

// ---------------------------------------------------------------- Adam ------------------------------------------------
// (AdamArgs, adam1, the per-group passes: dqo_adam.h — shared with the fused per-Gaussian tail, map_fused_tail.hip)
__global__ void adam_advance_kernel(int32_t* step_dev, const DqoRastHeader* frame_header) {
    if (frame_header != nullptr && frame_header->overflow != 0u) return;  // invalid frame: the step did not happen
    *step_dev += 1;
}

// One block per 256 consecutive Gaussians.  The block first lists the Gaussians it has to touch (LDS): all of them in dense mode;
// in the exact sparse mode (DqoAdamStep.moment_live) those with a gradient row (radii > 0) or non-zero moments — the others are
// fixed points of the update and their rows are neither read nor written.  It then runs one pass per parameter group over the
// list (adam_passes).
template <bool SPARSE, bool ATTACH>
__device__ __forceinline__ void adam_block(AdamArgs a, uint8_t* __restrict__ moment_live);

template <bool SPARSE, bool ATTACH>
__global__ __launch_bounds__(ADAM_THREADS) void adam_kernel(const AdamArgs a, uint8_t* __restrict__ moment_live) {
    // A frame flagged invalid by the forward (instance capacity / tile bucket exceeded: lists emptied, every gradient zero) must
    // not train: nothing is read or written, the caller re-captures and continues from the state it had.
    if (a.frame_header != nullptr && a.frame_header->overflow != 0u) return;
    adam_block<SPARSE, ATTACH>(a, moment_live);
    adam_take_ticket(a);
}

template <bool SPARSE, bool ATTACH>
__device__ __forceinline__ void adam_block(AdamArgs a, uint8_t* __restrict__ moment_live) {
    __shared__ uint32_t s_rows[ADAM_THREADS];  // Gaussian index | has-gradient << 31 | attach-loss member << 30
    __shared__ float s_att[ADAM_THREADS / 64];
    __shared__ int s_wave_n[ADAM_THREADS / 64];
    __shared__ float s_ss[7];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.step_dev != nullptr && tid == 0) adam_bias_to_lds(a, s_ss);
    const int idx = blockIdx.x * ADAM_THREADS + tid;
    bool hg = false, act = false, att = false;
    if (idx < a.P && !(a.row_flags != nullptr && (a.row_flags[idx] & DQO_ROW_FROZEN) != 0u)) {  // (a frozen row: DqoAdamStep.row_flags)
        hg = a.radii == nullptr || a.radii[idx] > 0;
        act = !SPARSE || hg || moment_live[idx] != 0;
        if (SPARSE && hg) moment_live[idx] = 1;  // only this thread ever looks at this byte
        if (ATTACH) att = a.attach_mask[idx] != 0;
    }
    const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
    if (lane == 0) s_wave_n[wave] = (int)__popcll(am);
    __syncthreads();
    if (a.step_dev != nullptr) adam_bias_from_lds(a, s_ss);
    if (ATTACH) adam_attach_gains(a);
    int before = 0, n_rows = 0;
#pragma unroll
    for (int w = 0; w < ADAM_THREADS / 64; w++) {
        const int c = s_wave_n[w];
        before += w < wave ? c : 0;
        n_rows += c;
    }
    if (act) s_rows[before + (int)__popcll(am & ((1ull << lane) - 1ull))] = (uint32_t)idx | (hg ? 0x80000000u : 0u) | (att ? 0x40000000u : 0u);
    __syncthreads();
    if (n_rows == 0) {
        if (ATTACH && a.attach_partial != nullptr && tid == 0) a.attach_partial[blockIdx.x] = 0.f;
        return;
    }
    float att_sum = adam_passes<ATTACH, ADAM_THREADS>(a, s_rows, n_rows, AdamGradGlobal{a});
    if (ATTACH && a.attach_partial != nullptr) {  // fixed-order block sum of the attach loss (the reported "scale_loss")
        att_sum = wave_red(att_sum);
        if (lane == 0) s_att[wave] = att_sum;
        __syncthreads();
        if (tid == 0) a.attach_partial[blockIdx.x] = (s_att[0] + s_att[1]) + (s_att[2] + s_att[3]);
    }
}

}  // namespace

int dqo_launch_map_activate(int P, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* opacity,
                            float* scales, float* rotations, hipStream_t s) {
    DQO_LAUNCH("activate_kernel", activate_kernel, dim3((P + 255) / 256), dim3(256), s, P, opacity_raw, scaling_raw, rotation_raw, opacity,
               scales, rotations);
    return DQO_OK;
}

size_t dqo_map_loss_ws_bytes(void) { return 4 * sizeof(double) * 1024; }

int dqo_launch_map_loss(int W, int H, const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                        const float* gt_depth, const uint8_t* render_mask, float color_weight, float depth_weight, float add_depth_thres,
                        float* loss_out, float* dL_dcolor, float* dL_ddepth, void* ws, hipStream_t s) {
    const int HW = W * H;
    double* partial = (double*)ws;
    auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15u) == 0u; };
    const bool vec = (HW % 4 == 0) && al16(color) && al16(depth) && al16(depth_index) && al16(gt_color) && al16(gt_depth) &&
                     al16(dL_dcolor) && al16(dL_ddepth) && (render_mask == nullptr || (reinterpret_cast<uintptr_t>(render_mask) & 3u) == 0u);
    if (vec) {
        const int nblk = max(1, min(512, (HW / 4 + LOSS_THREADS - 1) / LOSS_THREADS));
        DQO_LAUNCH("loss_reduce_kernel", loss_reduce_kernel<4>, dim3(nblk), dim3(LOSS_THREADS), s, HW, color, depth, depth_index, gt_color,
                   gt_depth, render_mask, add_depth_thres, partial);
        DQO_LAUNCH("loss_grad_kernel", loss_grad_kernel<4>, dim3(nblk), dim3(LOSS_THREADS), s, HW, nblk, color, depth, depth_index, gt_color,
                   gt_depth, render_mask, add_depth_thres, color_weight, depth_weight, partial, loss_out, dL_dcolor, dL_ddepth);
    } else {
        const int nblk = min(1024, (HW + LOSS_THREADS - 1) / LOSS_THREADS);
        DQO_LAUNCH("loss_reduce_kernel", loss_reduce_kernel<1>, dim3(nblk), dim3(LOSS_THREADS), s, HW, color, depth, depth_index, gt_color,
                   gt_depth, render_mask, add_depth_thres, partial);
        DQO_LAUNCH("loss_grad_kernel", loss_grad_kernel<1>, dim3(nblk), dim3(LOSS_THREADS), s, HW, nblk, color, depth, depth_index, gt_color,
                   gt_depth, render_mask, add_depth_thres, color_weight, depth_weight, partial, loss_out, dL_dcolor, dL_ddepth);
    }
    return DQO_OK;
}

int dqo_adam_args(const DqoAdamStep* st, int blocks, AdamArgs* out, bool* attach_out) {
    AdamArgs a;
    a.P = st->P, a.M = st->M;
    a.beta1 = st->beta1, a.beta2 = st->beta2, a.eps = st->eps;
    a.beta1_d = dqo_beta_double(st->beta1), a.beta2_d = dqo_beta_double(st->beta2);
    a.omb1 = (float)(1.0 - a.beta1_d), a.omb2 = (float)(1.0 - a.beta2_d);
    // bias corrections in double like torch (python floats), then cast
    const double tstep = st->step >= 1 ? (double)st->step : 1.0;  // (ignored by the kernel when step_dev is given)
    const double bc1 = 1.0 - pow(a.beta1_d, tstep), bc2 = 1.0 - pow(a.beta2_d, tstep);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.step_xyz = (float)((double)st->lr_xyz / bc1);
    a.step_dc = (float)((double)st->lr_f_dc / bc1);
    a.step_rest = (float)((double)st->lr_f_rest / bc1);
    a.step_opacity = (float)((double)st->lr_opacity / bc1);
    a.step_scaling = (float)((double)st->lr_scaling / bc1);
    a.step_rotation = (float)((double)st->lr_rotation / bc1);
    a.xyz = st->xyz, a.shs = st->shs, a.opacity_raw = st->opacity_raw, a.scaling_raw = st->scaling_raw, a.rotation_raw = st->rotation_raw;
    a.g_xyz = st->g_means3D, a.g_shs = st->g_sh, a.g_opacity = st->g_opacity, a.g_scales = st->g_scales, a.g_rot = st->g_rotations;
    a.m_xyz = st->m_xyz, a.m_shs = st->m_shs, a.m_opacity = st->m_opacity, a.m_scaling = st->m_scaling, a.m_rotation = st->m_rotation;
    a.v_xyz = st->v_xyz, a.v_shs = st->v_shs, a.v_opacity = st->v_opacity, a.v_scaling = st->v_scaling, a.v_rotation = st->v_rotation;
    a.act_opacity = st->act_opacity, a.act_scales = st->act_scales, a.act_rotations = st->act_rotations;
    a.radii = st->radii;
    a.step_dev = st->step_dev;
    a.lr_xyz = st->lr_xyz, a.lr_dc = st->lr_f_dc, a.lr_rest = st->lr_f_rest, a.lr_opacity = st->lr_opacity, a.lr_scaling = st->lr_scaling;
    a.lr_rotation = st->lr_rotation;
    a.attach_mask = st->attach_mask, a.init_xyz = st->init_xyz, a.init_scaling = st->init_scaling_raw, a.init_rotation = st->init_rotation_raw;
    a.attach_partial = st->attach_partial, a.frame_header = st->frame_header;
    const bool advance_inside = st->step_dev != nullptr && st->block_ticket != nullptr && blocks > 0;
    a.step_advance = advance_inside ? st->step_dev : nullptr, a.block_ticket = st->block_ticket;
    a.bias_table = (advance_inside && st->step_dev != nullptr) ? st->bias_table : nullptr;
    const bool attach = st->attach_mask != nullptr && (st->attach_count > 0 || st->attach_gains != nullptr);
    a.attach_gains = attach ? st->attach_gains : nullptr;
    a.row_flags = st->row_flags, a.confidence = st->confidence, a.lr_table = st->lr_table;
    DQO_CHECK_ARG(st->lr_table == nullptr || st->step_dev != nullptr, "lr_table is read on the device: it needs step_dev");
    DQO_CHECK_ARG(!attach || (st->init_xyz && st->init_scaling_raw && st->init_rotation_raw), "attach_mask needs the three init_* tensors");
    DQO_CHECK_ARG(st->P < (1 << 30), "P must stay below 2^30");
    // d/dp of 1000 * mean((p - p0)^2) over |a| rows of 3 (scaling, xyz) / 4 (rotation) elements = 2000 (p - p0) / (len |a|)
    a.attach_g3 = (attach && st->attach_count > 0) ? (float)(2000.0 / (3.0 * (double)st->attach_count)) : 0.f;
    a.attach_g4 = (attach && st->attach_count > 0) ? (float)(2000.0 / (4.0 * (double)st->attach_count)) : 0.f;
    DQO_CHECK_ARG(st->M >= 0 && st->M * 3 < 256, "M out of range");
    a.row_magic = st->M > 0 ? ((1ull << 39) + (uint64_t)(3 * st->M) - 1) / (uint64_t)(3 * st->M) : 0ull;
    DQO_CHECK_ARG((int64_t)st->P * (st->M > 0 ? st->M : 1) * 3 < (int64_t)0x7fffffff, "P * M * 3 must stay below 2^31");
    DQO_CHECK_ARG(st->moment_live == nullptr || st->radii != nullptr, "moment_live needs radii");
    *out = a;
    *attach_out = attach;
    return DQO_OK;
}

int dqo_launch_map_adam(const DqoAdamStep* st, hipStream_t s) {
    const int blocks = (st->P + ADAM_THREADS - 1) / ADAM_THREADS;
    AdamArgs a;
    bool attach = false;
    const int rc = dqo_adam_args(st, blocks, &a, &attach);
    if (rc) return rc;
    const bool advance_inside = a.step_advance != nullptr;
    if (blocks > 0) {  // (an empty map still advances the step count)
        if (st->moment_live != nullptr) {
            if (attach) DQO_LAUNCH("adam_kernel", (adam_kernel<true, true>), dim3(blocks), dim3(ADAM_THREADS), s, a, st->moment_live);
            else DQO_LAUNCH("adam_kernel", (adam_kernel<true, false>), dim3(blocks), dim3(ADAM_THREADS), s, a, st->moment_live);
        } else {
            if (attach) DQO_LAUNCH("adam_kernel", (adam_kernel<false, true>), dim3(blocks), dim3(ADAM_THREADS), s, a, st->moment_live);
            else DQO_LAUNCH("adam_kernel", (adam_kernel<false, false>), dim3(blocks), dim3(ADAM_THREADS), s, a, st->moment_live);
        }
    }
    if (st->step_dev != nullptr && !advance_inside)
        DQO_LAUNCH("adam_advance_kernel", adam_advance_kernel, dim3(1), dim3(1), s, st->step_dev, st->frame_header);
    return DQO_OK;
}

// ---- dqo_adam_multi: torch.optim.Adam.step() over a handful of dense tensors in one launch ----
namespace {
constexpr int ADAMM_THREADS = 256, ADAMM_PER_THREAD = 4;
struct AdamMultiArgs {
    float* p[DQO_ADAM_MULTI_MAX];
    const float* g[DQO_ADAM_MULTI_MAX];
    float* m[DQO_ADAM_MULTI_MAX];
    float* v[DQO_ADAM_MULTI_MAX];
    int64_t n[DQO_ADAM_MULTI_MAX];
    float step_size[DQO_ADAM_MULTI_MAX];      // lr / (1 - beta1^t)
    uint32_t first_block[DQO_ADAM_MULTI_MAX + 1];  // tensor t owns blocks [first_block[t], first_block[t + 1])
    int n_tensors;
    float beta1, beta2, omb1, omb2, eps, bc2_sqrt;
    // dqo_adam_multi_dev: the step count lives on the device (a captured graph replays with the count of ITS step, not the capture's)
    const int32_t* step_dev;  // steps taken so far, or NULL: step_size[] / bc2_sqrt above are the host's
    double beta1_d, beta2_d;
    double lr[DQO_ADAM_MULTI_MAX];
};
__global__ void adam_step_bump_kernel(int32_t* step_dev) { step_dev[0] += 1; }
__global__ __launch_bounds__(ADAMM_THREADS) void adam_multi_kernel(const AdamMultiArgs q) {
    int t = 0;  // (block-uniform: at most 16 trips)
    while (t + 1 < q.n_tensors && blockIdx.x >= q.first_block[t + 1]) t++;
    AdamArgs a;  // (adam1 reads these fields only)
    a.beta2 = q.beta2, a.omb1 = q.omb1, a.omb2 = q.omb2, a.eps = q.eps, a.bc2_sqrt = q.bc2_sqrt;
    float ss = q.step_size[t];
    if (q.step_dev != nullptr) {  // (kernel-uniform) the host path's expressions, in double, by one thread of the block
        __shared__ float s_bc[2];
        if (threadIdx.x == 0) {
            const double st = (double)(q.step_dev[0] + 1);
            const double bc1 = 1.0 - pow(q.beta1_d, st), bc2 = 1.0 - pow(q.beta2_d, st);
            s_bc[0] = (float)sqrt(bc2), s_bc[1] = (float)(q.lr[t] / bc1);
        }
        __syncthreads();
        a.bc2_sqrt = s_bc[0], ss = s_bc[1];
    }
    float* const p = q.p[t];
    const float* const g = q.g[t];
    float* const m = q.m[t];
    float* const v = q.v[t];
    const int64_t n = q.n[t];
    const int64_t base = ((int64_t)(blockIdx.x - q.first_block[t]) * ADAMM_THREADS + threadIdx.x) * ADAMM_PER_THREAD;
    if (base + ADAMM_PER_THREAD <= n && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0u) {
        float4 pp = *reinterpret_cast<const float4*>(p + base), mm = *reinterpret_cast<const float4*>(m + base);
        float4 vv = *reinterpret_cast<const float4*>(v + base);
        const float4 gg = *reinterpret_cast<const float4*>(g + base);
        adam1(pp.x, gg.x, mm.x, vv.x, a, ss), adam1(pp.y, gg.y, mm.y, vv.y, a, ss);
        adam1(pp.z, gg.z, mm.z, vv.z, a, ss), adam1(pp.w, gg.w, mm.w, vv.w, a, ss);
        *reinterpret_cast<float4*>(p + base) = pp, *reinterpret_cast<float4*>(m + base) = mm, *reinterpret_cast<float4*>(v + base) = vv;
    } else {
        for (int64_t i = base; i < base + ADAMM_PER_THREAD && i < n; i++) {
            float pi = p[i], mi = m[i], vi = v[i];
            adam1(pi, g[i], mi, vi, a, ss);
            p[i] = pi, m[i] = mi, v[i] = vi;
        }
    }
}
}  // namespace

int dqo_launch_adam_multi(const DqoAdamTensor* ts, int n_tensors, int step, double beta1, double beta2, double eps, hipStream_t s,
                          int32_t* step_dev, int bump) {
    AdamMultiArgs q;
    if (step_dev != nullptr) step = 1;  // (placeholders below; the kernel forms the corrections from the device count)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    q.step_dev = step_dev, q.beta1_d = beta1, q.beta2_d = beta2;
    q.n_tensors = n_tensors, q.beta1 = (float)beta1, q.beta2 = (float)beta2, q.eps = (float)eps, q.bc2_sqrt = (float)sqrt(bc2);
    q.omb1 = (float)(1.0 - beta1), q.omb2 = (float)(1.0 - beta2);
    uint32_t blocks = 0;
    for (int t = 0; t < DQO_ADAM_MULTI_MAX; t++) {
        const bool in = t < n_tensors;
        q.p[t] = in ? ts[t].p : nullptr, q.g[t] = in ? ts[t].g : nullptr, q.m[t] = in ? ts[t].m : nullptr, q.v[t] = in ? ts[t].v : nullptr;
        q.n[t] = in ? ts[t].n : 0;
        q.step_size[t] = in ? (float)(ts[t].lr / bc1) : 0.f;
        q.lr[t] = in ? ts[t].lr : 0.0;
        q.first_block[t] = blocks;
        if (in) blocks += (uint32_t)((ts[t].n + ADAMM_THREADS * ADAMM_PER_THREAD - 1) / (ADAMM_THREADS * ADAMM_PER_THREAD));
    }
    q.first_block[DQO_ADAM_MULTI_MAX] = blocks;
    for (int t = n_tensors; t <= DQO_ADAM_MULTI_MAX; t++) q.first_block[t] = blocks;
    if (blocks != 0) DQO_LAUNCH("adam_multi_kernel", adam_multi_kernel, dim3(blocks), dim3(ADAMM_THREADS), s, q);
    if (step_dev != nullptr && bump) DQO_LAUNCH("adam_step_bump_kernel", adam_step_bump_kernel, dim3(1), dim3(1), s, step_dev);
    return DQO_OK;
}

size_t dqo_map_attach_ws_bytes(int P) { return sizeof(double) * (size_t)((P + ATT_THREADS - 1) / ATT_THREADS + 1); }

int dqo_launch_map_attach(int P, const float* scaling, const float* xyz, const float* rotation, const float* scaling0, const float* xyz0,
                          const float* rotation0, const uint8_t* mask, int attach_count, float* loss, float* g_scaling, float* g_xyz,
                          float* g_rotation, void* ws, hipStream_t s) {
    const int blocks = (P + ATT_THREADS - 1) / ATT_THREADS;
    // d/dp of 1000 * mean((p - p0)^2) over |a| rows of 3 (scaling, xyz) / 4 (rotation) elements = 2000 (p - p0) / (len |a|); with an empty
    // set the loss and its gradient are zero (mapper.py:813)
    const float g3 = attach_count > 0 ? (float)(2000.0 / (3.0 * (double)attach_count)) : 0.f;
    const float g4 = attach_count > 0 ? (float)(2000.0 / (4.0 * (double)attach_count)) : 0.f;
    double* partial = (double*)ws;
    DQO_LAUNCH("attach_loss_kernel", attach_loss_kernel, dim3(blocks), dim3(ATT_THREADS), s, P, scaling, xyz, rotation, scaling0, xyz0, rotation0,
               mask, g3, g4, g_scaling, g_xyz, g_rotation, partial);
    DQO_LAUNCH("attach_loss_sum_kernel", attach_loss_sum_kernel, dim3(1), dim3(ATT_THREADS), s, blocks, partial, loss);
    return DQO_OK;
}

int dqo_launch_history_merge(int P, int M, float max_weight, int first_row, const uint8_t* row_flags, const float* conf0, const float* conf,
                             const float* xyz0, const float* shs0, const float* scaling0, const float* rot0_unit, float* xyz, float* shs,
                             float* scaling_raw, float* rotation_raw, hipStream_t s) {
    if (P <= 0 || !(max_weight > 0.f)) return DQO_OK;  // mapper.py:608-609
    DQO_LAUNCH("history_merge_kernel", history_merge_kernel, dim3((P + HM_THREADS - 1) / HM_THREADS), dim3(HM_THREADS), s, P, M, max_weight,
               first_row, row_flags, conf0, conf, xyz0, shs0, scaling0, rot0_unit, xyz, shs, scaling_raw, rotation_raw);
    return DQO_OK;
}

// the one-thread launch that advances the device step count when the Adam launch does not do it itself (DqoAdamStep.block_ticket == NULL)
int dqo_launch_adam_advance(int32_t* step_dev, const DqoRastHeader* frame_header, hipStream_t s) {
    DQO_LAUNCH("adam_advance_kernel", adam_advance_kernel, dim3(1), dim3(1), s, step_dev, frame_header);
    return DQO_OK;
}
