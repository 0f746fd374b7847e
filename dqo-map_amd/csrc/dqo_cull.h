// Exact (conservative) footprint test used at binning time and inside the blend kernels.
//
// The reference puts a Gaussian into every tile of the bounding SQUARE of radius ceil(3 sigma_max) around its centre
// (auxiliary.h:49-57, forward.cu:316-326).  A list entry only ever does something for a pixel when
//     power <= 0  and  alpha = min(0.99, opacity * exp(power)) >= 1/255        (forward.cu:763-772, backward.cu:940-946)
// with power = -q/2, q = A dx^2 + 2 B dx dy + C dy^2.  An entry for which NO pixel of the tile can pass that test is dead
// weight in the tile's list: it changes no output (colour, depth, hit ids, weights, T, n_touched, gradients) — it only
// advances the reference's internal `contributor` counters.  Dropping such entries is therefore output-invariant, and
// for the thin, obliquely seen surfels of this workload it removes a large share of all instances.
//
// The test is conservative: the minimum of the convex form q over the CONTINUOUS rectangle spanned by the pixel centres
// is a lower bound of q at every pixel; the entry is dropped only if that bound exceeds the 1/255 cut-off by a margin
// that dominates fp32 evaluation error of `power` in the blend loop.  IEEE ops only (no FMA contraction) so that the
// count pass and the emit pass take bit-identical decisions.
#pragma once
#include <hip/hip_runtime.h>

// q threshold above which alpha < 1/255 for every pixel: alpha >= 1/255  <=>  q <= 2 ln(255 * opacity)
__device__ __forceinline__ float dqo_q_threshold(float opacity) {
#pragma clang fp contract(off)
    return 2.0f * logf(255.0f * fmaxf(opacity, 1e-30f));
}

// true if some pixel centre in [x0,x1] x [y0,y1] may satisfy q <= qthr (i.e. the entry must be kept)
__device__ __forceinline__ bool dqo_splat_hits_rect(float mx, float my, float A, float B, float C, float qthr, float x0, float y0,
                                                    float x1, float y1) {
#pragma clang fp contract(off)
    if (qthr < 0.f) return false;  // opacity < 1/255: alpha < 1/255 even at the centre
    const float dx0 = x0 - mx, dx1 = x1 - mx, dy0 = y0 - my, dy1 = y1 - my;
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;  // centre inside: q_min = 0
    float qmin = 3.0e38f;
    // vertical edges dx = const, dy free in [dy0, dy1]:  q = A dx^2 + 2 B dx dy + C dy^2,  dy* = -B dx / C
    {
        const float invC = 1.0f / C;
        float dy = fminf(dy1, fmaxf(dy0, -B * dx0 * invC));
        qmin = fminf(qmin, A * dx0 * dx0 + 2.0f * B * dx0 * dy + C * dy * dy);
        dy = fminf(dy1, fmaxf(dy0, -B * dx1 * invC));
        qmin = fminf(qmin, A * dx1 * dx1 + 2.0f * B * dx1 * dy + C * dy * dy);
    }
    // horizontal edges dy = const, dx free in [dx0, dx1]:  dx* = -B dy / A
    {
        const float invA = 1.0f / A;
        float dx = fminf(dx1, fmaxf(dx0, -B * dy0 * invA));
        qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * dy0 + C * dy0 * dy0);
        dx = fminf(dx1, fmaxf(dx0, -B * dy1 * invA));
        qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * dy1 + C * dy1 * dy1);
    }
    // error budget of the fp32 evaluation of `power` at the farthest corner + a relative and an absolute guard band
    const float ddx = fmaxf(fabsf(dx0), fabsf(dx1)), ddy = fmaxf(fabsf(dy0), fabsf(dy1));
    const float tmax = fabsf(A) * ddx * ddx + 2.0f * fabsf(B) * ddx * ddy + fabsf(C) * ddy * ddy;
    const float margin = 0.05f + 0.01f * qthr + 4.0e-6f * tmax;
    return !(qmin > qthr + margin);  // NaN-safe: keeps the entry
}

// The quadratic form of the blend loops, forward.cu:758-760 / backward.cu:937-939, in the reference's operation order with separate
// IEEE multiplies and adds.  For a long thin splat seen far from its centre the three terms cancel to a result ~1e4 times smaller
// than they are, so WHERE the roundings happen decides the fourth digit of alpha: evaluated like this the result is bit-identical
// to the oracle's (and to a build of the reference without FMA contraction); two extra VALU instructions per (pixel, entry) pair.
__device__ __forceinline__ float dqo_power(float A, float B, float C, float dx, float dy) {
#pragma clang fp contract(off)
    return -0.5f * (A * dx * dx + C * dy * dy) - B * dx * dy;
}

// dqo_power on a conic whose A and C arrive multiplied by -0.5 (the blend kernels pre-scale them where an entry's record goes into
// LDS): scaling by a power of two commutes with every rounding, so (A' dx) dx + (C' dy) dy - (B dx) dy is bit for bit
// -0.5 (A dx dx + C dy dy) - B dx dy, one multiply per (pixel, entry) pair shorter
__device__ __forceinline__ float dqo_power_pre(float Ah, float B, float Ch, float dx, float dy) {
#pragma clang fp contract(off)
    return (Ah * dx * dx + Ch * dy * dy) - B * dx * dy;
}

// exp(power) of the blend loops (forward.cu:770, backward.cu:943).  One v_exp_f32 (|rel err| ~1e-7 for power in [-5.6, 0],
// the only range that survives the 1/255 cut) instead of the ~15-instruction libm expf; forward and backward share it so
// the backward reproduces the forward's alpha bit for bit.
__device__ __forceinline__ float dqo_gauss(float power) { return __expf(power); }
// 1/x by v_rcp_f32 (1 ulp) for the T / (1 - alpha) recurrences of the backward (backward.cu:948, 980)
__device__ __forceinline__ float dqo_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Lane -> pixel of the blend kernels' quadrant wave (one wave64 per 8x8 quadrant, one pixel per lane): every 16-lane DPP row owns one
// 4x4 block of the quadrant — row r = lane >> 4 at ((r & 1) * 4, (r >> 1) * 4), lane s = lane & 15 of the row at (s & 3, s >> 2) inside
// it.  The forward records per list entry WHICH rows saw it (a 4-bit row code in the entry's live byte), and the backward lets every
// row walk its own sub-list (rast_backward_blend.hip): an entry costs a row a step only if one of ITS 16 pixels has work for it.
__device__ __forceinline__ uint32_t dqo_lane_x(int lane) { return (uint32_t)(((lane >> 4) & 1) * 4 + (lane & 3)); }
__device__ __forceinline__ uint32_t dqo_lane_y(int lane) { return (uint32_t)((lane >> 5) * 4 + ((lane >> 2) & 3)); }
// 64-bit lane mask -> 4-bit row code: bit r = some lane of DPP row r is set (two s_quadmask: 64 lanes -> 16 quads -> 4 rows)
__device__ __forceinline__ uint32_t dqo_row_code(unsigned long long m) {
    unsigned long long q;
    uint32_t c;
    asm("s_quadmask_b64 %0, %1" : "=s"(q) : "s"(m) : "scc");
    asm("s_quadmask_b32 %0, %1" : "=s"(c) : "s"((uint32_t)q) : "scc");
    return c;
}
