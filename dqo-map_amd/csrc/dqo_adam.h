// Adam over the six parameter groups of the Gaussian map with the activation Jacobians and the attach loss folded in
// (SLAM/gaussian_pointcloud.py:331-378, SLAM/multiprocess/mapper.py:548, :812-829) as device code shared by
//   adam_kernel          (map_fused.hip)      : gradient rows come from HBM (dqo_rast_backward wrote them)
//   gaussian_tail_kernel (map_fused_tail.hip) : gradient rows come from LDS (the same block has just computed them)
// Both run the SAME statements on the same operands (one template, the gradient source is its parameter), so parameters and moments
// come out bit-identical whichever path produced the gradient.
#pragma once
#include "dqo_common.h"

struct AdamArgs {
    int P, M;
    float beta1, beta2, eps, bc2_sqrt;
    // torch forms 1 - beta and beta^t from python floats (double) and rounds the RESULT to float: 1 - 0.999 is 0.001f there, while
    // 1.f - 0.999f is 0.00099998713 (1.3e-5 off, and with it exp_avg_sq and the bias correction).  The ABI carries the betas as floats:
    // dqo_beta_double() takes them back to the decimal they were written as.
    float omb1, omb2;          // (float)(1 - beta1), (float)(1 - beta2)
    double beta1_d, beta2_d;
    float step_xyz, step_dc, step_rest, step_opacity, step_scaling, step_rotation;
    float *xyz, *shs, *opacity_raw, *scaling_raw, *rotation_raw;                 // parameters (raw), updated in place
    const float *g_xyz, *g_shs, *g_opacity, *g_scales, *g_rot;                    // gradients w.r.t. the ACTIVATED parameters
    float *m_xyz, *m_shs, *m_opacity, *m_scaling, *m_rotation;                    // exp_avg
    float *v_xyz, *v_shs, *v_opacity, *v_scaling, *v_rotation;                    // exp_avg_sq
    float *act_opacity, *act_scales, *act_rotations;                              // optional: activations of the updated parameters
    const int32_t* radii;                                                         // optional: radii == 0 => the gradient row is zero and unread
    uint64_t row_magic;                                                           // ceil(2^39 / (3 M)): division by the SH row length
    const int32_t* step_dev;                                                      // optional: step count on the device (hipGraph replay)
    float lr_xyz, lr_dc, lr_rest, lr_opacity, lr_scaling, lr_rotation;            // used with step_dev
    // attach loss (mapper.py:812-829): elementwise pull of the raw scaling / xyz / rotation towards their values at the start of
    // the mapping call, for the Gaussians of attach_mask
    const uint8_t* attach_mask;
    const float *init_xyz, *init_scaling, *init_rotation;
    float attach_g3, attach_g4;                                                   // 2000 / (3 |a|), 2000 / (4 |a|)
    float* attach_partial;
    const DqoRastHeader* frame_header;                                            // optional: overflow flag => the launch is a no-op
    int32_t* step_advance;                                                        // optional: the last block to finish adds 1 to it
    int32_t* block_ticket;                                                        // (with step_advance) blocks finished so far
    float* bias_table;                                                            // optional (with step_dev): DqoAdamStep.bias_table
    const float* attach_gains;                                                    // optional: DqoAdamStep.attach_gains (replaces attach_g3/4)
    const uint8_t* row_flags;                                                     // optional: DqoAdamStep.row_flags (DQO_ROW_FROZEN rows are skipped)
    float* confidence;                                                            // optional: DqoAdamStep.confidence
    const float* lr_table;                                                        // optional (with step_dev): DqoAdamStep.lr_table
};

// a float beta as the double it was most likely written as: rounded to seven decimals (0.999f = 0.99900001287... -> 0.999)
static inline double dqo_beta_double(float b) { return (double)(long long)((double)b * 1e7 + 0.5) / 1e7; }

// host side: DqoAdamStep -> AdamArgs (argument checks included); blocks = the launch's grid size (for the step-advance ticket)
int dqo_adam_args(const DqoAdamStep* st, int blocks, AdamArgs* out, bool* attach);

#ifdef __HIPCC__
constexpr int ADAM_THREADS = 256;

// The moments and the gradients are touched exactly once per iteration (0.7 GB of the kernel's 0.83 GB): non-temporal loads /
// stores keep them from evicting the rasteriser's tables and the parameters out of L2 / Infinity Cache.
template <typename T>
__device__ __forceinline__ T ldnt(const T* p) {
    return __builtin_nontemporal_load(p);
}
template <typename T>
__device__ __forceinline__ void stnt(T v, T* p) {
    __builtin_nontemporal_store(v, p);
}
// The moments of the SMALL groups ([P,3] xyz / scaling, [P] opacity): plain accesses.  A non-temporal access to part of a line is
// served memory-side without the neighbours' help; through L2 the rows of adjacent Gaussians share their lines (measured, round 4:
// fused tail 129 -> 121 us on cfg 3, 508 -> 467 us on cfg 5; without any of this traffic: 111 / 368 — `-DDQO_SMALL_MV_NT` restores the
// non-temporal form).  The SH moments (whole 192-byte rows, 0.7 GB touched once per iteration) stay non-temporal: plain accesses there
// measured +0 / +20 us.
template <typename T>
__device__ __forceinline__ T ldsm(const T* p) {
#ifdef DQO_SMALL_MV_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void stsm(T v, T* p) {
#ifdef DQO_SMALL_MV_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__device__ __forceinline__ float adam_wave_red(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamArgs& a, float step_size) {
    // torch.optim.Adam (single-/multi-tensor and fused paths share this math):
    //   m = lerp(m, g, 1 - beta1); v = beta2 v + (1 - beta2) g^2; p -= step_size * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
    // Separate IEEE operations (no FMA contraction): every instantiation of the kernel then produces the same bits, which the
    // exact sparse mode's equivalence to the dense update is tested against.
#pragma clang fp contract(off)
    m = m + (g - m) * a.omb1;
    v = v * a.beta2 + a.omb2 * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - step_size * (m / denom);
}

// Gradient source of adam_kernel: the rows dqo_rast_backward wrote to HBM (a row without a gradient may be unwritten memory: it is read
// at a clamped address and discarded — a load under a lane condition would make the compiler drain every earlier load first).
// k = list row of the block, i = element index in the parameter tensor, has_g = the Gaussian has a gradient row.
struct AdamGradGlobal {
    const AdamArgs& a;
    // (the small rows through plain loads, like their moments: ldsm)
    __device__ __forceinline__ float xyz(int, uint32_t, size_t i, bool has_g) const { return ldsm(&a.g_xyz[has_g ? i : 0]); }
    __device__ __forceinline__ float scales(int, uint32_t, size_t i, bool has_g) const { return ldsm(&a.g_scales[has_g ? i : 0]); }
    // SH element j of the row; ei = its element index, e0 = a valid element index of the same trip
    __device__ __forceinline__ float sh(int, uint32_t, uint32_t ei, uint32_t e0, bool has_g) const { return ldnt(&a.g_shs[has_g ? ei : e0]); }
    __device__ __forceinline__ float opacity(int, uint32_t i, bool has_g) const { return ldsm(&a.g_opacity[has_g ? i : 0u]); }
    __device__ __forceinline__ float4 rot(int, uint32_t i, bool has_g) const { return reinterpret_cast<const float4*>(a.g_rot)[has_g ? i : 0u]; }
    // does the row's f_dc gradient (SH coefficient 0) have a non-zero element?  (mapper.py:908-909; three loads at a clamped address)
    __device__ __forceinline__ bool dc_nonzero(int, uint32_t i, bool has_g) const {
        const size_t e = has_g ? (size_t)i * (size_t)(3 * a.M) : 0;
        const float g0 = ldnt(&a.g_shs[e]), g1 = ldnt(&a.g_shs[e + 1]), g2 = ldnt(&a.g_shs[e + 2]);
        return has_g && (g0 != 0.f || g1 != 0.f || g2 != 0.f);
    }
};

// The bias corrections of step t (double, like the host path / torch's python floats): out[0] = sqrt(1 - beta2^t), out[1..6] = the six
// learning rates / (1 - beta1^t).  Two double-precision pow() calls, a square root and six divisions: several hundred instructions.
__device__ __forceinline__ void adam_bias_compute(const AdamArgs& a, const int step, float* out /*[7]*/) {
    const double t = (double)step;
    const double bc1 = 1.0 - pow(a.beta1_d, t), bc2 = 1.0 - pow(a.beta2_d, t);
    out[0] = (float)sqrt(bc2);
    float lr[6] = {a.lr_xyz, a.lr_dc, a.lr_rest, a.lr_opacity, a.lr_scaling, a.lr_rotation};
    if (a.lr_table != nullptr) {  // DqoAdamStep.lr_table: the learning rates of THIS mapping call, from device memory (uniform loads)
#pragma unroll
        for (int i = 0; i < 6; i++) lr[i] = a.lr_table[i];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) out[1 + i] = (float)((double)lr[i] / bc1);
}
// ... of the launch's step (the device-resident count), by the calling thread(s): from DqoAdamStep.bias_table when it holds this step
// (seven loads from uniform addresses), computed otherwise — the same function either way, so the same bits.
__device__ __forceinline__ void adam_bias_to_lds(const AdamArgs& a, float* s_ss /*[7]*/) {
    const int step = *a.step_dev;
    if (a.bias_table != nullptr && __float_as_int(a.bias_table[7]) == step) {
#pragma unroll
        for (int i = 0; i < 7; i++) s_ss[i] = a.bias_table[i];
        return;
    }
    adam_bias_compute(a, step, s_ss);
}
// DqoAdamStep.attach_gains: the attach term's two factors from device memory (uniform loads)
__device__ __forceinline__ void adam_attach_gains(AdamArgs& a) {
    if (a.attach_gains != nullptr) a.attach_g3 = a.attach_gains[0], a.attach_g4 = a.attach_gains[1];
}
__device__ __forceinline__ void adam_bias_from_lds(AdamArgs& a, const float* s_ss) {
    a.bc2_sqrt = s_ss[0], a.step_xyz = s_ss[1], a.step_dc = s_ss[2], a.step_rest = s_ss[3], a.step_opacity = s_ss[4];
    a.step_scaling = s_ss[5], a.step_rotation = s_ss[6];
}

// ---- the element updates of the passes: loaded values in, stores out ----
struct AdamXyzVals {  // one element of the xyz / scaling pass
    float p, m, v, ps, ms, vs, gx_ld, gs_ld, p0, ps0;
};
template <bool ATTACH>
__device__ __forceinline__ void adam_xyz_update(const AdamArgs& a, const uint32_t r, const size_t i, AdamXyzVals x, float& att_sum) {
    // Separate IEEE operations, like adam1: the gradient of the raw scaling is a sum of two products (g exp(s) + gain ds) — torch forms
    // it with two roundings (the exp Jacobian is an op of its own, the attach term arrives through another autograd edge), and under
    // -ffp-contract=fast WHICH product an fma would swallow depends on the code around the inlined call: the fused tail and adam_kernel
    // must not differ by that (profiles/r06_ab_tail_small_rows.txt).
#pragma clang fp contract(off)
    const bool has_g = (r >> 31) != 0u;
    float p = x.p, m = x.m, v = x.v, ps = x.ps, ms = x.ms, vs = x.vs;
    float gx = has_g ? x.gx_ld : 0.f, gs = (has_g ? x.gs_ld : 0.f) * expf(ps);  // d exp(x)/dx = exp(x)
    if (ATTACH) {
        const bool at = ((r >> 30) & 1u) != 0u;
        const float dx = at ? p - x.p0 : 0.f, ds = at ? ps - x.ps0 : 0.f;
        gx += a.attach_g3 * dx, gs += a.attach_g3 * ds;
        att_sum += 0.5f * a.attach_g3 * (dx * dx + ds * ds);
    }
    adam1(p, gx, m, v, a, a.step_xyz);
    a.xyz[i] = p, stsm(m, &a.m_xyz[i]), stsm(v, &a.v_xyz[i]);
    adam1(ps, gs, ms, vs, a, a.step_scaling);
    a.scaling_raw[i] = ps, stsm(ms, &a.m_scaling[i]), stsm(vs, &a.v_scaling[i]);
    if (a.act_scales) a.act_scales[i] = expf(ps);  // = activate_kernel on the updated value
}

struct AdamRowVals {  // one row of the opacity / rotation pass
    float p, m, v, go_ld;
    float4 q, mq, vq, gr_ld, q0;
    float conf;   // DqoAdamStep.confidence of the row (loaded with the others)
    bool dc_nz;   // the row's f_dc gradient has a non-zero element
};
template <bool ATTACH>
__device__ __forceinline__ void adam_row_update(const AdamArgs& a, const uint32_t r, AdamRowVals x, float& att_sum) {
#pragma clang fp contract(off)  // (see adam_xyz_update)
    const uint32_t i = r & 0x3fffffffu;
    const bool has_g = (r >> 31) != 0u;
    float p = x.p, m = x.m, v = x.v;
    float4 q = x.q, mq = x.mq, vq = x.vq;
    const float sg = 1.0f / (1.0f + expf(-p));
    adam1(p, (has_g ? x.go_ld : 0.f) * (sg * (1.f - sg)), m, v, a, a.step_opacity);
    a.opacity_raw[i] = p, stsm(m, &a.m_opacity[i]), stsm(v, &a.v_opacity[i]);
    if (a.act_opacity) a.act_opacity[i] = 1.0f / (1.0f + expf(-p));

    const float4 g = has_g ? x.gr_ld : make_float4(0.f, 0.f, 0.f, 0.f);
    // F.normalize backward: y = q / n, n = max(|q|, eps):  dq = (g - y (y . g)) / n
    const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float yx = q.x / nrm, yy = q.y / nrm, yz = q.z / nrm, yw = q.w / nrm;
    const float dot = yx * g.x + yy * g.y + yz * g.z + yw * g.w;
    float4 gq = make_float4((g.x - yx * dot) / nrm, (g.y - yy * dot) / nrm, (g.z - yz * dot) / nrm, (g.w - yw * dot) / nrm);
    if (ATTACH) {
        const bool at = ((r >> 30) & 1u) != 0u;
        const float4 q0 = x.q0;
        const float4 d = at ? make_float4(q.x - q0.x, q.y - q0.y, q.z - q0.z, q.w - q0.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        gq.x += a.attach_g4 * d.x, gq.y += a.attach_g4 * d.y, gq.z += a.attach_g4 * d.z, gq.w += a.attach_g4 * d.w;
        att_sum += 0.5f * a.attach_g4 * (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
    }
    adam1(q.x, gq.x, mq.x, vq.x, a, a.step_rotation);
    adam1(q.y, gq.y, mq.y, vq.y, a, a.step_rotation);
    adam1(q.z, gq.z, mq.z, vq.z, a, a.step_rotation);
    adam1(q.w, gq.w, mq.w, vq.w, a, a.step_rotation);
    reinterpret_cast<float4*>(a.rotation_raw)[i] = q;
    if (a.act_rotations) {
        const float n2 = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        reinterpret_cast<float4*>(a.act_rotations)[i] = make_float4(q.x / n2, q.y / n2, q.z / n2, q.w / n2);
    }
    reinterpret_cast<float4*>(a.m_rotation)[i] = mq;
    reinterpret_cast<float4*>(a.v_rotation)[i] = vq;
    // mapper.py:908-910: `_confidence[(f_dc.grad.abs() != 0).any(-1)] += 1` for the rows this launch trains
    if (a.confidence != nullptr && x.dc_nz) a.confidence[i] = x.conf + 1.0f;
}

// One pass per parameter group over the block's list of rows (s_rows[k] = Gaussian index | has-gradient << 31 | attach-loss member
// << 30, k < n_rows); element e of a group with rows of `len` floats belongs to list row e / len, so the passes stay dense over the
// list whatever its sparsity (no lane conditions on the loads), and a Gaussian's rows are read as whole contiguous pieces.
// Returns this thread's share of the attach loss at the pre-update parameters.
// THREADS = threads of the block that shares the list (n_rows <= THREADS).
template <bool ATTACH, int THREADS, typename GradSrc>
__device__ __forceinline__ float adam_passes(const AdamArgs& a, const uint32_t* s_rows, const int n_rows, const GradSrc& gsrc) {
    const int tid = threadIdx.x;
    float att_sum = 0.f;
    // xyz (identity activation) and scaling (exp): element-wise, [P,3]
    for (int e = tid; e < 3 * n_rows; e += THREADS) {
        const uint32_t k = (uint32_t)e / 3u, j = (uint32_t)e - 3u * k, r = s_rows[k];
        const bool has_g = (r >> 31) != 0u;
        const size_t i = (size_t)(r & 0x3fffffffu) * 3 + j;
        // all loads of the element in one round (a row outside the attach set reads element 0 and discards it)
        AdamXyzVals x;
        x.p = a.xyz[i], x.m = ldsm(&a.m_xyz[i]), x.v = ldsm(&a.v_xyz[i]);
        x.ps = a.scaling_raw[i], x.ms = ldsm(&a.m_scaling[i]), x.vs = ldsm(&a.v_scaling[i]);
        x.gx_ld = gsrc.xyz((int)k, j, i, has_g), x.gs_ld = gsrc.scales((int)k, j, i, has_g);
        x.p0 = x.ps0 = 0.f;
        if (ATTACH) {
            const size_t ai = ((r >> 30) & 1u) ? i : 0;
            x.p0 = a.init_xyz[ai], x.ps0 = a.init_scaling[ai];
        }
        adam_xyz_update<ATTACH>(a, r, i, x, att_sum);
    }
    // SH coefficients [P,M,3]: coefficient 0 = f_dc (lr feature_lr), the rest = f_rest (feature_lr / 20).  Four elements per
    // trip with every load issued before the first use; all loads unconditional on clamped addresses (a load under a lane
    // condition makes the compiler drain every earlier load first).
    const uint32_t row = (uint32_t)a.M * 3u;
    const int nsh = n_rows * (int)row;
    // The loads of trip t + 1 are issued before trip t is computed and stored (two register sets): the block's memory pipe
    // never idles between trips.
    struct ShTrip {
        float p[4], m[4], v[4], g[4];
        uint32_t ei[4], fl[4];  // element index in the [P * M * 3] arrays; 1 = in range, 2 = has a gradient, 4 = f_dc
    };
    auto load_trip = [&](int e0) {
        ShTrip t;
        uint32_t kk[4], jj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = e0 + u * THREADS;
            const bool in = e < nsh;
            const uint32_t ee = in ? (uint32_t)e : 0u;
            const uint32_t k = (uint32_t)(((uint64_t)ee * a.row_magic) >> 39);  // ee / row, exact for ee < 2^31, row < 2^8
            const uint32_t j = ee - k * row, r = s_rows[k];
            kk[u] = k, jj[u] = j;
            t.ei[u] = (r & 0x3fffffffu) * row + j;
            t.fl[u] = in ? (1u | ((r >> 31) ? 2u : 0u) | (j < 3u ? 4u : 0u)) : 0u;
            t.p[u] = a.shs[t.ei[u]];
            t.m[u] = ldnt(&a.m_shs[t.ei[u]]);
            t.v[u] = ldnt(&a.v_shs[t.ei[u]]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) t.g[u] = gsrc.sh((int)kk[u], jj[u], t.ei[u], t.ei[0], (t.fl[u] & 2u) != 0u);
        return t;
    };
    if (nsh > 0) {
        ShTrip cur = load_trip(tid);
        for (int e0 = tid; e0 < nsh; e0 += 4 * THREADS) {
            const bool more = e0 + 4 * THREADS < nsh;  // (per thread; the loads of an absent trip are skipped as a whole)
            ShTrip nxt = cur;
            if (more) nxt = load_trip(e0 + 4 * THREADS);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!(cur.fl[u] & 1u)) continue;
                float p = cur.p[u], m = cur.m[u], v = cur.v[u];
                adam1(p, (cur.fl[u] & 2u) ? cur.g[u] : 0.f, m, v, a, (cur.fl[u] & 4u) ? a.step_dc : a.step_rest);
                a.shs[cur.ei[u]] = p, stnt(m, &a.m_shs[cur.ei[u]]), stnt(v, &a.v_shs[cur.ei[u]]);
            }
            cur = nxt;
        }
    }
    // opacity (sigmoid) [P] and rotation (normalize) [P,4]
    if (tid < n_rows) {
        const uint32_t r = s_rows[tid], i = r & 0x3fffffffu;
        const bool has_g = (r >> 31) != 0u;
        // all loads of the row in one round
        AdamRowVals x;
        x.p = a.opacity_raw[i], x.m = ldsm(&a.m_opacity[i]), x.v = ldsm(&a.v_opacity[i]);
        x.q = reinterpret_cast<float4*>(a.rotation_raw)[i];
        x.mq = reinterpret_cast<float4*>(a.m_rotation)[i], x.vq = reinterpret_cast<float4*>(a.v_rotation)[i];
        x.go_ld = gsrc.opacity(tid, i, has_g);
        x.gr_ld = gsrc.rot(tid, i, has_g);
        x.q0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ATTACH) x.q0 = reinterpret_cast<const float4*>(a.init_rotation)[((r >> 30) & 1u) ? i : 0u];
        x.conf = 0.f, x.dc_nz = false;
        if (a.confidence != nullptr) x.conf = a.confidence[i], x.dc_nz = gsrc.dc_nonzero(tid, i, has_g);  // (kernel-uniform branch)
        adam_row_update<ATTACH>(a, r, x, att_sum);
    }
    return att_sum;
}

// 16-byte non-temporal forms (the builtins want a native vector type, not HIP's float4 class)
typedef float dqo_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt4(const float* p) {
    const dqo_f4v t = __builtin_nontemporal_load(reinterpret_cast<const dqo_f4v*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void stnt4(const float4 v, float* p) {
    dqo_f4v t;
    t.x = v.x, t.y = v.y, t.z = v.z, t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<dqo_f4v*>(p));
}

// adam_passes with the SH pass in float4s, for the fused tail where ONE WAVE owns the list and its lifetime is a chain of dependent
// memory rounds: rows of 48 floats (M = 16, 16-byte aligned tensors) move as float4s, U per thread and trip — U x THREADS x 4 floats per
// trip instead of 4 x THREADS.  Element for element the arithmetic is adam_passes': same bits.
template <bool ATTACH, int THREADS, bool VEC4, int U, typename GradSrc>
__device__ __forceinline__ float adam_passes_tail(const AdamArgs& a, const uint32_t* s_rows, const int n_rows, const GradSrc& gsrc) {
    static_assert(VEC4, "the scalar form is adam_passes");
    const int tid = threadIdx.x;
    float att_sum = 0.f;
    // Roles of the block's two waves (round 6; in-kernel stamps, profiles/r06_tail_stamps.txt: the row pass — a thread per list row,
    // ~50 rows — ran on wave 0 alone behind the SH pass, 6 us during which wave 1 waited at the block's last barrier): wave 1 takes the
    // row pass (rows 0..63; wave 0 the rows beyond, if any) and in exchange at most ONE trip of the SH pass, wave 0 the rest of it.
    // Element for element the statements are unchanged: same bits (the attach loss is grouped differently: a reported scalar).
#ifndef DQO_TAIL_ROLE_SPLIT
#define DQO_TAIL_ROLE_SPLIT 1
#endif
    constexpr bool SPLIT = DQO_TAIL_ROLE_SPLIT && THREADS == 128;
    const int wave = tid >> 6, lane = tid & 63;
    const int my_row = SPLIT ? (wave == 1 ? lane : 64 + lane) : tid;
    auto rows_pass = [&]() {
        // opacity (sigmoid) [P] and rotation (normalize) [P,4]
        if (my_row < n_rows) {
            const uint32_t r = s_rows[my_row], i = r & 0x3fffffffu;
            const bool has_g = (r >> 31) != 0u;
            AdamRowVals x;
            x.p = a.opacity_raw[i], x.m = ldsm(&a.m_opacity[i]), x.v = ldsm(&a.v_opacity[i]);
            x.q = reinterpret_cast<float4*>(a.rotation_raw)[i];
            x.mq = reinterpret_cast<float4*>(a.m_rotation)[i], x.vq = reinterpret_cast<float4*>(a.v_rotation)[i];
            x.go_ld = gsrc.opacity(my_row, i, has_g);
            x.gr_ld = gsrc.rot(my_row, i, has_g);
            x.q0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ATTACH) x.q0 = reinterpret_cast<const float4*>(a.init_rotation)[((r >> 30) & 1u) ? i : 0u];
            x.conf = 0.f, x.dc_nz = false;
            if (a.confidence != nullptr) x.conf = a.confidence[i], x.dc_nz = gsrc.dc_nonzero(my_row, i, has_g);  // (kernel-uniform branch)
            adam_row_update<ATTACH>(a, r, x, att_sum);
        }
    };
    // xyz (identity activation) and scaling (exp): element-wise, [P,3]
    for (int e = tid; e < 3 * n_rows; e += THREADS) {
        const uint32_t k = (uint32_t)e / 3u, j = (uint32_t)e - 3u * k, r = s_rows[k];
        const bool has_g = (r >> 31) != 0u;
        const size_t i = (size_t)(r & 0x3fffffffu) * 3 + j;
        AdamXyzVals x;
        x.p = a.xyz[i], x.m = ldsm(&a.m_xyz[i]), x.v = ldsm(&a.v_xyz[i]);
        x.ps = a.scaling_raw[i], x.ms = ldsm(&a.m_scaling[i]), x.vs = ldsm(&a.v_scaling[i]);
        x.gx_ld = gsrc.xyz((int)k, j, i, has_g), x.gs_ld = gsrc.scales((int)k, j, i, has_g);
        x.p0 = x.ps0 = 0.f;
        if (ATTACH) {
            const size_t ai = ((r >> 30) & 1u) ? i : 0;
            x.p0 = a.init_xyz[ai], x.ps0 = a.init_scaling[ai];
        }
        adam_xyz_update<ATTACH>(a, r, i, x, att_sum);
    }
    // SH pass: a row of 48 floats = 12 float4s; float4 f of the list belongs to list row f / 12
    {
        struct ShTrip4 {
            float4 p[U], m[U], v[U], g[U];
            uint32_t ei[U], fl[U];  // first element index of the float4; 1 = in range, 2 = has a gradient, 4 = elements 0..2 are f_dc
        };
        const int n4 = n_rows * 12;
        // this wave's share of the float4s: [f_begin, f_end), walked with a stride of FSTRIDE lanes
        constexpr int FSTRIDE = SPLIT ? 64 : THREADS;
        const int s0 = n4 <= 2 * U * 64 ? (n4 + 1) / 2 : n4 - U * 64;  // (SPLIT) wave 1: one trip at most
        const int f_begin = SPLIT ? (wave == 0 ? 0 : s0) : 0, f_end = SPLIT ? (wave == 0 ? s0 : n4) : n4;
        auto load_trip4 = [&](int f0) {
            ShTrip4 t;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int f = f0 + u * FSTRIDE;
                const bool in = f < f_end;
                const uint32_t fc = in ? (uint32_t)f : 0u;
                const uint32_t k = (fc * 10923u) >> 17, j4 = fc - 12u * k, r = s_rows[k];  // fc / 12, exact for fc < 2^15
                t.ei[u] = (r & 0x3fffffffu) * 48u + 4u * j4;
                t.fl[u] = in ? (1u | ((r >> 31) ? 2u : 0u) | (j4 == 0u ? 4u : 0u)) : 0u;
                t.p[u] = *reinterpret_cast<const float4*>(&a.shs[t.ei[u]]);
                t.m[u] = ldnt4(&a.m_shs[t.ei[u]]);
                t.v[u] = ldnt4(&a.v_shs[t.ei[u]]);
                t.g[u] = gsrc.sh4((int)k, 4u * j4);
            }
            return t;
        };
        if (SPLIT && wave == 1) rows_pass();  // (its loads go out first; the SH share follows)
        for (int f0 = f_begin + (SPLIT ? lane : tid); f0 < f_end; f0 += U * FSTRIDE) {
            const ShTrip4 cur = load_trip4(f0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!(cur.fl[u] & 1u)) continue;
                const bool hg = (cur.fl[u] & 2u) != 0u, dc = (cur.fl[u] & 4u) != 0u;
                float4 p = cur.p[u], m = cur.m[u], v = cur.v[u];
                const float4 g = cur.g[u];
                adam1(p.x, hg ? g.x : 0.f, m.x, v.x, a, dc ? a.step_dc : a.step_rest);
                adam1(p.y, hg ? g.y : 0.f, m.y, v.y, a, dc ? a.step_dc : a.step_rest);
                adam1(p.z, hg ? g.z : 0.f, m.z, v.z, a, dc ? a.step_dc : a.step_rest);
                adam1(p.w, hg ? g.w : 0.f, m.w, v.w, a, a.step_rest);  // (element 3 of a row is f_rest)
                *reinterpret_cast<float4*>(&a.shs[cur.ei[u]]) = p;
                stnt4(m, &a.m_shs[cur.ei[u]]);
                stnt4(v, &a.v_shs[cur.ei[u]]);
            }
        }
    }
    if (!(SPLIT && wave == 1)) rows_pass();
    return att_sum;
}

// DqoAdamStep.block_ticket: the device-side step count advances inside the launch — every block has read it at its start, so the
// block that takes the last ticket may bump it (and hands the ticket counters back at zero for the next launch).
// Two levels (round 6): a block takes a ticket on ONE OF UP TO 64 LINES (blockIdx % lines), the last block of a line takes one of the
// `lines` tickets of word 0, the last of those advances the count.  A single counter is one address that all ~4 000 blocks of a
// 500 k map add to, and same-address returning atomics are served one per ~11 ns memory-side: 43 us of queue — hidden while the
// blocks are long and finish spread out, the kernel's floor when they are short (a mapping call that trains a tenth of the map:
// 81 -> 58 us, profiles/r06_tail_ticket.txt).
// (no fence: the only ordering needed is "read of the step count before the ticket", and that load has long been consumed)
#define DQO_TICKET_LINES 64
static_assert(DQO_TICKET_WORDS == 16 + 16 * DQO_TICKET_LINES, "DqoAdamStep.block_ticket: include/dqo_raster.h and the kernels disagree");
__device__ __forceinline__ void adam_take_ticket(const AdamArgs& a) {
    if (a.step_advance != nullptr && threadIdx.x == 0) {
        const int grid = (int)gridDim.x;
        const int lines = min(DQO_TICKET_LINES, max(1, grid / 16));
        const int l = (int)blockIdx.x % lines;
        const int on_line = (grid - l + lines - 1) / lines;  // blocks b < grid with b % lines == l
        int* const line = a.block_ticket + 16 + 16 * l;
        if (atomicAdd(line, 1) != on_line - 1) return;
        *line = 0;
        if (atomicAdd(a.block_ticket, 1) != lines - 1) return;
        *a.block_ticket = 0;
        const int next = *a.step_advance + 1;
        *a.step_advance = next;
        if (a.bias_table != nullptr) {  // the next launch's bias corrections, once, instead of once per block / wave there
            float b[7];
            adam_bias_compute(a, next, b);
#pragma unroll
            for (int i = 0; i < 7; i++) a.bias_table[i] = b[i];
            a.bias_table[7] = __int_as_float(next);
        }
    }
}
#endif  // __HIPCC__
