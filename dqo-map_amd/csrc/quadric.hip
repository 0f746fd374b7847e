// Dual-quadric pose residual for gfx950: ellipsoid -> Q* -> C* = P Q* P^T -> ellipse -> bbox -> 1 - IoU, forward + analytic
// backward, and the whole 20-step Adam loop of Object_Optimize_only in ONE launch.
//
// Replaces /root/reference/SLAM/multiprocess/quadrics.py:285-290 (bboxes_iou), 2018-2091 (Ellipse_tensor),
// 2144-2220 (Ellipsoid_tensor), 2234-2298 (Object_Optimize_only).  The reference runs ~60 eager 4x4 torch ops (+ a
// torch.linalg.eig and two host syncs) per iteration per object — pure launch latency.  Here one lane owns one
// (object, view) pair (residual) or one object (Adam loop); the 2x2 eigen-decomposition is closed form.
#include "dqo_common.h"

namespace {

struct QuadOut {
    float bbox[4];
    float loss;
    float g_axes[3], g_R[9], g_center[3];
    int valid;
};

__device__ void quadric_eval(const float* axes, const float* Rm, const float* ctr, const float* P, const float* obs, QuadOut& o) {
    // Ellipsoid_tensor.forward, quadrics.py:2178-2206:  Q = [[R A R^T - c c^T, -c], [-c^T, -1]]
    const float A[3] = {axes[0] * axes[0], axes[1] * axes[1], axes[2] * axes[2]};
    float Q[4][4];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float s = 0;
            for (int k = 0; k < 3; k++) s += Rm[3 * i + k] * A[k] * Rm[3 * j + k];
            Q[i][j] = s - ctr[i] * ctr[j];
        }
    for (int i = 0; i < 3; i++) Q[i][3] = Q[3][i] = -ctr[i];
    Q[3][3] = -1.f;
    float PQ[3][4];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += P[4 * i + k] * Q[k][j];
            PQ[i][j] = s;
        }
    float C[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += PQ[i][k] * P[4 * j + k];
            C[i][j] = s;
        }
    // Ellipse_tensor.__init__, quadrics.py:2019-2068
    float Cs[3][3], Cn[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Cs[i][j] = 0.5f * (C[i][j] + C[j][i]);
    const float nrm = -Cs[2][2];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Cn[i][j] = Cs[i][j] / nrm;
    const float mux = -Cn[0][2], muy = -Cn[1][2];
    const float p = Cn[0][0] + mux * mux, q = Cn[0][1] + mux * muy, r = Cn[1][1] + muy * muy;
    const float m = 0.5f * (p + r), hd = 0.5f * (p - r);
    const float h = sqrtf(hd * hd + q * q);
    const float l1 = m + h, l2 = m - h;
    const float a1 = fabsf(l1), a2 = fabsf(l2);
    const float k = (h > 0.f) ? hd / h : 0.f;
    const float c2 = 0.5f * (1.f + k), s2 = 0.5f * (1.f - k);
    // ComputeBbox, quadrics.py:2076-2091
    const float X2 = a1 * c2 + a2 * s2, Y2 = a1 * s2 + a2 * c2;
    const float X = sqrtf(X2), Y = sqrtf(Y2);
    o.bbox[0] = mux - X;
    o.bbox[1] = muy - Y;
    o.bbox[2] = mux + X;
    o.bbox[3] = muy + Y;
    // bboxes_iou(obs, pred) with python min/max semantics, quadrics.py:283-290
    const float* b1 = obs;
    const float* b2 = o.bbox;
    const bool r_pred = b2[2] < b1[2], l_pred = b2[0] > b1[0], b_pred = b2[3] < b1[3], t_pred = b2[1] > b1[1];
    const float xr = r_pred ? b2[2] : b1[2], xl = l_pred ? b2[0] : b1[0];
    const float yb = b_pred ? b2[3] : b1[3], yt = t_pred ? b2[1] : b1[1];
    const float iw_raw = xr - xl, ih_raw = yb - yt;
    const bool w_pos = !(0.f > iw_raw), h_pos = !(0.f > ih_raw);
    const float iw = w_pos ? iw_raw : 0.f, ih = h_pos ? ih_raw : 0.f;
    const float inter = iw * ih;
    const float area1 = (b1[2] - b1[0]) * (b1[3] - b1[1]);
    const float w2 = b2[2] - b2[0], h2 = b2[3] - b2[1];
    const float uni = area1 + w2 * h2 - inter;
    const float iou = inter / uni;
    o.loss = 1.f - iou;
    o.valid = (o.loss == 1.f) ? 0 : 1;

    // ---- backward ----
    const float d_inter = -(1.f / uni + inter / (uni * uni));
    const float d_area2 = inter / (uni * uni);
    float d_b2[4] = {0, 0, 0, 0};
    d_b2[2] += d_area2 * h2;
    d_b2[0] -= d_area2 * h2;
    d_b2[3] += d_area2 * w2;
    d_b2[1] -= d_area2 * w2;
    const float d_iw = w_pos ? d_inter * ih : 0.f, d_ih = h_pos ? d_inter * iw : 0.f;
    if (r_pred) d_b2[2] += d_iw;
    if (l_pred) d_b2[0] -= d_iw;
    if (b_pred) d_b2[3] += d_ih;
    if (t_pred) d_b2[1] -= d_ih;
    float d_mux = d_b2[0] + d_b2[2], d_muy = d_b2[1] + d_b2[3];
    const float d_X = d_b2[2] - d_b2[0], d_Y = d_b2[3] - d_b2[1];
    const float d_X2 = d_X * 0.5f / X, d_Y2 = d_Y * 0.5f / Y;
    const float d_a1 = d_X2 * c2 + d_Y2 * s2, d_a2 = d_X2 * s2 + d_Y2 * c2;
    const float d_k = 0.5f * (d_X2 * (a1 - a2) + d_Y2 * (a2 - a1));
    const float sg1 = (l1 > 0.f) ? 1.f : (l1 < 0.f ? -1.f : 0.f), sg2 = (l2 > 0.f) ? 1.f : (l2 < 0.f ? -1.f : 0.f);
    const float d_l1 = d_a1 * sg1, d_l2 = d_a2 * sg2;
    const float d_m = d_l1 + d_l2;
    float d_h = d_l1 - d_l2, d_hd = 0.f, d_q = 0.f;
    if (h > 0.f) {
        d_hd += d_k / h;
        d_h += -d_k * hd / (h * h);
        d_hd += d_h * hd / h;
        d_q += d_h * q / h;
    }
    const float d_p = 0.5f * d_m + 0.5f * d_hd, d_r = 0.5f * d_m - 0.5f * d_hd;
    float d_Cn[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    d_Cn[0][0] += d_p;
    d_Cn[1][1] += d_r;
    d_Cn[0][1] += 0.5f * d_q;
    d_Cn[1][0] += 0.5f * d_q;
    d_mux += d_p * 2.f * mux + d_q * muy;
    d_muy += d_r * 2.f * muy + d_q * mux;
    d_Cn[0][2] -= d_mux;
    d_Cn[1][2] -= d_muy;
    float d_Cs[3][3];
    float d_nrm = 0.f;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            d_Cs[i][j] = d_Cn[i][j] / nrm;
            d_nrm += -d_Cn[i][j] * Cs[i][j] / (nrm * nrm);
        }
    d_Cs[2][2] -= d_nrm;
    float d_C[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) d_C[i][j] = 0.5f * (d_Cs[i][j] + d_Cs[j][i]);
    float d_Q[4][4];
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
            float s = 0;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) s += P[4 * i + a] * d_C[i][j] * P[4 * j + b];
            d_Q[a][b] = s;
        }
    float d_Qs[4][4];
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) d_Qs[a][b] = 0.5f * (d_Q[a][b] + d_Q[b][a]);
    for (int i = 0; i < 3; i++) {
        float gsum = -(d_Qs[i][3] + d_Qs[3][i]);
        for (int j = 0; j < 3; j++) gsum += -(d_Qs[i][j] + d_Qs[j][i]) * ctr[j];
        o.g_center[i] = gsum;
    }
    for (int k2 = 0; k2 < 3; k2++) {
        float gA = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) gA += d_Qs[i][j] * Rm[3 * i + k2] * Rm[3 * j + k2];
        o.g_axes[k2] = gA * 2.f * axes[k2];
    }
    for (int i = 0; i < 3; i++)
        for (int k2 = 0; k2 < 3; k2++) {
            float gsum = 0;
            for (int j = 0; j < 3; j++) gsum += (d_Qs[i][j] + d_Qs[j][i]) * A[k2] * Rm[3 * j + k2];
            o.g_R[3 * i + k2] = gsum;
        }
    if (!o.valid) {
        for (int i = 0; i < 3; i++) o.g_axes[i] = o.g_center[i] = 0.f;
        for (int i = 0; i < 9; i++) o.g_R[i] = 0.f;
    }
}

__global__ void quadric_iou_kernel(int B, const float* __restrict__ axes, const float* __restrict__ R, const float* __restrict__ center,
                                   const float* __restrict__ P34, const float* __restrict__ obs, float* __restrict__ bbox,
                                   float* __restrict__ loss, int32_t* __restrict__ valid, float* __restrict__ g_axes,
                                   float* __restrict__ g_R, float* __restrict__ g_center) {
#pragma clang fp contract(off)
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float a[3], Rm[9], c[3], P[12], ob[4];
    for (int i = 0; i < 3; i++) a[i] = axes[3 * b + i], c[i] = center[3 * b + i];
    for (int i = 0; i < 9; i++) Rm[i] = R[9 * b + i];
    for (int i = 0; i < 12; i++) P[i] = P34[12 * b + i];
    for (int i = 0; i < 4; i++) ob[i] = obs[4 * b + i];
    QuadOut o;
    quadric_eval(a, Rm, c, P, ob, o);
    for (int i = 0; i < 4; i++) bbox[4 * b + i] = o.bbox[i];
    loss[b] = o.loss;
    valid[b] = o.valid;
    for (int i = 0; i < 3; i++) g_axes[3 * b + i] = o.g_axes[i], g_center[3 * b + i] = o.g_center[i];
    for (int i = 0; i < 9; i++) g_R[9 * b + i] = o.g_R[i];
}

// One lane per object: the whole Adam trajectory (quadrics.py:2251-2285) without leaving registers.
__global__ void quadric_adam_kernel(int n_obj, int n_iters, const int32_t* __restrict__ view_offset, const float* __restrict__ P34_views,
                                    const float* __restrict__ obs_views, const int32_t* __restrict__ view_schedule,
                                    float* __restrict__ axes, float* __restrict__ R, float* __restrict__ center,
                                    float* __restrict__ loss_hist) {
#pragma clang fp contract(off)
    const int ob = blockIdx.x * blockDim.x + threadIdx.x;
    if (ob >= n_obj) return;
    float prm[15], mm[15], vv[15];
    for (int i = 0; i < 3; i++) prm[i] = axes[3 * ob + i], prm[3 + i] = center[3 * ob + i];
    for (int i = 0; i < 9; i++) prm[6 + i] = R[9 * ob + i];
    for (int i = 0; i < 15; i++) mm[i] = 0.f, vv[i] = 0.f;
    const int v0 = view_offset[ob], nv = view_offset[ob + 1] - v0;
    double pow1 = 1.0, pow2 = 1.0;  // beta1^step, beta2^step
    for (int it = 0; it < n_iters; it++) {
        int vi = view_schedule[ob * n_iters + it];
        if (vi < 0) vi += nv;
        vi = min(max(vi, 0), nv - 1);
        float P[12], obx[4];
        for (int i = 0; i < 12; i++) P[i] = P34_views[12 * (v0 + vi) + i];
        for (int i = 0; i < 4; i++) obx[i] = obs_views[4 * (v0 + vi) + i];
        QuadOut o;
        quadric_eval(prm, prm + 6, prm + 3, P, obx, o);
        if (loss_hist) loss_hist[ob * n_iters + it] = o.loss;
        if (!o.valid) continue;  // loss == 1: the reference raises and `continue`s before backward()/step()
        pow1 *= 0.9;
        pow2 *= 0.999;
        const double bc1 = 1.0 - pow1, bc2 = 1.0 - pow2;
        const float bc2_sqrt = (float)sqrt(bc2);
        for (int i = 0; i < 15; i++) {
            const float gi = i < 3 ? o.g_axes[i] : (i < 6 ? o.g_center[i - 3] : o.g_R[i - 6]);
            const double lr = (i >= 3 && i < 6) ? 0.001 : 0.01;
            mm[i] = mm[i] + (gi - mm[i]) * (1.0f - 0.9f);
            vv[i] = vv[i] * 0.999f + (1.0f - 0.999f) * gi * gi;
            const float denom = sqrtf(vv[i]) / bc2_sqrt + 1e-15f;
            const float step_size = (float)(lr / bc1);
            prm[i] = prm[i] - step_size * (mm[i] / denom);
        }
    }
    for (int i = 0; i < 3; i++) axes[3 * ob + i] = prm[i], center[3 * ob + i] = prm[3 + i];
    for (int i = 0; i < 9; i++) R[9 * ob + i] = prm[6 + i];
}

}  // namespace

int dqo_launch_quadric_iou(int B, const float* axes, const float* R, const float* center, const float* P34, const float* obs,
                           float* bbox, float* loss, int32_t* valid, float* g_axes, float* g_R, float* g_center, hipStream_t s) {
    DQO_LAUNCH("quadric_iou_kernel", quadric_iou_kernel, dim3((B + 63) / 64), dim3(64), s, B, axes, R, center, P34, obs, bbox, loss, valid, g_axes,
                       g_R, g_center);
    return DQO_OK;
}

int dqo_launch_quadric_adam(int n_obj, int n_iters, const int32_t* view_offset, const float* P34_views, const float* obs_views,
                            const int32_t* view_schedule, float* axes, float* R, float* center, float* loss_hist, hipStream_t s) {
    DQO_LAUNCH("quadric_adam_kernel", quadric_adam_kernel, dim3((n_obj + 63) / 64), dim3(64), s, n_obj, n_iters, view_offset, P34_views, obs_views,
                       view_schedule, axes, R, center, loss_hist);
    return DQO_OK;
}
