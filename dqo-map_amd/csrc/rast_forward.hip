// Forward pass of the depth-aware Gaussian rasteriser for gfx950 (MI355X).
//
// Replaces (semantics, not structure) /root/reference/submodules/diff-gaussian-rasterizer-depth/
//   cuda_rasterizer/forward.cu:238-354   preprocessCUDA            -> preprocess_kernel
//   cuda_rasterizer/rasterizer_impl.cu:70-142, 303-365 (cub scan, duplicateWithKeys, cub radix sort,
//   identifyTileRanges, host tile compaction with two D2H syncs)   -> tile_scan_kernel, emit_kernel, tile_sort_kernel
//   cuda_rasterizer/forward.cu:636-866   renderCUDA_withMask       -> blend_forward_kernel
//
// MI355X design: no host synchronisation anywhere; binning is a per-tile counting sort (atomic tile
// histogram in K1 -> one-block scan -> cursor emit) followed by an LDS sort of each tile's (depth, id) keys, so an
// instance crosses HBM once as an 8-byte key instead of 6 radix passes over 12-byte pairs; everything the blend loop
// needs per Gaussian is precomputed once into 16-byte SoA records (the reference rebuilds the quaternion rotation and
// four uncoalesced gathers per (pixel, Gaussian) pair, forward.cu:779-791).
#include "dqo_common.h"
#include "dqo_cull.h"

namespace {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ------------------------------------------------------------------------------------------------------------------
// K1: per-Gaussian preprocess.  Bit-faithful to the oracle: IEEE ops only, no FMA contraction in this kernel (it is
// bandwidth bound; contraction would only move discrete decisions such as ceil(radius) and the tile rect).
// ------------------------------------------------------------------------------------------------------------------
struct PreOut {
    int radius;      // 0 = culled
    uint32_t touch;  // unmasked tiles in rect
    int rminx, rminy, rmaxx, rmaxy;
};

constexpr int K1_THREADS = 256;
constexpr int K1_ITEMS = 4;

__device__ __forceinline__ void quat_to_R(const float4 q, float Rm[3][3]) {
#pragma clang fp contract(off)
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    Rm[0][0] = 1.f - 2.f * (y * y + z * z);
    Rm[0][1] = 2.f * (x * y - r * z);
    Rm[0][2] = 2.f * (x * z + r * y);
    Rm[1][0] = 2.f * (x * y + r * z);
    Rm[1][1] = 1.f - 2.f * (x * x + z * z);
    Rm[1][2] = 2.f * (y * z - r * x);
    Rm[2][0] = 2.f * (x * z - r * y);
    Rm[2][1] = 2.f * (y * z + r * x);
    Rm[2][2] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ int arg_min3(float a, float b, float c) { return (a <= b && a <= c) ? 0 : ((b <= a && b <= c) ? 1 : 2); }
__device__ __forceinline__ int arg_max3(float a, float b, float c) { return (a >= b && a >= c) ? 0 : ((b >= a && b >= c) ? 1 : 2); }

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

__global__ __launch_bounds__(K1_THREADS) void preprocess_kernel(const DqoView v, const float* __restrict__ means3D,
                                                                const float* __restrict__ scales,
                                                                const float* __restrict__ rotations,
                                                                const float* __restrict__ opacities,
                                                                const float* __restrict__ shs,
                                                                const float* __restrict__ colors_precomp,
                                                                const int32_t* __restrict__ tile_mask, DqoGeomLayout g,
                                                                int32_t* __restrict__ radii_out,
                                                                int32_t* __restrict__ n_touched_out) {
#pragma clang fp contract(off)
    __shared__ uint32_t s_visible;
    const int tid = threadIdx.x;
    const int P = v.P;
    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) view[i] = v.view[i], proj[i] = v.proj[i];
    const float cam0 = v.campos[0], cam1 = v.campos[1], cam2 = v.campos[2];
    if (tid == 0) s_visible = 0;
    __syncthreads();

    uint32_t nvis = 0;
#pragma unroll 1
    for (int it = 0; it < K1_ITEMS; it++) {
        const int idx = blockIdx.x * (K1_THREADS * K1_ITEMS) + it * K1_THREADS + tid;
        if (idx >= P) continue;
        int radius = 0;
        int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
        do {
            const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
            // in_frustum, auxiliary.h:139-165
            const float hx = proj[0] * px + proj[4] * py + proj[8] * pz + proj[12];
            const float hy = proj[1] * px + proj[5] * py + proj[9] * pz + proj[13];
            const float hw = proj[3] * px + proj[7] * py + proj[11] * pz + proj[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float projx = hx * p_w, projy = hy * p_w;
            const float tvx = view[0] * px + view[4] * py + view[8] * pz + view[12];
            const float tvy = view[1] * px + view[5] * py + view[9] * pz + view[13];
            const float tvz = view[2] * px + view[6] * py + view[10] * pz + view[14];
            if (tvz <= 0.2f || (double)projx < -1.3 || (double)projx > 1.3 || (double)projy < -1.3 || (double)projy > 1.3) break;
            // computeCov3D, forward.cu:202-235
            const float sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
            const float4 q = reinterpret_cast<const float4*>(rotations)[idx];
            float Rm[3][3];
            quat_to_R(q, Rm);
            const float s[3] = {v.scale_mod * sx, v.scale_mod * sy, v.scale_mod * sz};
            float Mm[3][3];
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int i = 0; i < 3; i++) Mm[k][i] = s[k] * Rm[i][k];
            float c3[6];
            {
                int o = 0;
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = i; j < 3; j++) c3[o++] = Mm[0][i] * Mm[0][j] + Mm[1][i] * Mm[1][j] + Mm[2][i] * Mm[2][j];
            }
            // computeCov2D, forward.cu:158-197
            const float limx = 1.3f * v.tanfovx, limy = 1.3f * v.tanfovy;
            const float txtz = tvx / tvz, tytz = tvy / tvz;
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * tvz;
            const float ty = fminf(limy, fmaxf(-limy, tytz)) * tvz;
            const float J00 = v.focal_x / tvz, J02 = -(v.focal_x * tx) / (tvz * tvz);
            const float J11 = v.focal_y / tvz, J12 = -(v.focal_y * ty) / (tvz * tvz);
            float A0[3], A1[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                A0[j] = J00 * view[j * 4 + 0] + J02 * view[j * 4 + 2];
                A1[j] = J11 * view[j * 4 + 1] + J12 * view[j * 4 + 2];
            }
            const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
            float VA0[3], VA1[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                VA0[i] = V[i][0] * A0[0] + V[i][1] * A0[1] + V[i][2] * A0[2];
                VA1[i] = V[i][0] * A1[0] + V[i][1] * A1[1] + V[i][2] * A1[2];
            }
            const float ca = A0[0] * VA0[0] + A0[1] * VA0[1] + A0[2] * VA0[2] + 0.3f;
            const float cb = A0[0] * VA1[0] + A0[1] * VA1[1] + A0[2] * VA1[2];
            const float cc = A1[0] * VA1[0] + A1[1] * VA1[1] + A1[2] * VA1[2] + 0.3f;
            const float det = ca * cc - cb * cb;
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float conx = cc * det_inv, cony = -cb * det_inv, conz = ca * det_inv;
            const float mid = 0.5f * (ca + cc);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(v.color_sigma * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix(v, S, c) = v * S * 0.5 + c with double intermediate, auxiliary.h:44-47
            const float pixx = (float)((double)(projx * (float)v.W) * 0.5 + (double)v.cx);
            const float pixy = (float)((double)(projy * (float)v.H) * 0.5 + (double)v.cy);
            // getRect, auxiliary.h:49-57
            const int ir = (int)my_radius;
            rminx = min(v.gx, max(0, (int)((pixx - (float)ir) / (float)DQO_TILE)));
            rminy = min(v.gy, max(0, (int)((pixy - (float)ir) / (float)DQO_TILE)));
            rmaxx = min(v.gx, max(0, (int)((pixx + (float)ir + (float)(DQO_TILE - 1)) / (float)DQO_TILE)));
            rmaxy = min(v.gy, max(0, (int)((pixy + (float)ir + (float)(DQO_TILE - 1)) / (float)DQO_TILE)));
            if ((rmaxx - rminx) * (rmaxy - rminy) == 0) break;
            // colour: computeColorFromSH, forward.cu:104-155
            float rgb[3];
            uint32_t clampbits = 0;
            if (colors_precomp == nullptr) {
                const float dxx = px - cam0, dyy = py - cam1, dzz = pz - cam2;
                const float len = sqrtf(dxx * dxx + dyy * dyy + dzz * dzz);
                const float x = dxx / len, y = dyy / len, z = dzz / len;
                const float* sh = shs + (size_t)idx * v.M * 3;
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float result = kSH_C0 * sh[ch];
                    if (v.D > 0) {
                        result = result - kSH_C1 * y * sh[3 + ch] + kSH_C1 * z * sh[6 + ch] - kSH_C1 * x * sh[9 + ch];
                        if (v.D > 1) {
                            result = result + kSH_C2[0] * xy * sh[12 + ch] + kSH_C2[1] * yz * sh[15 + ch] +
                                     kSH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + kSH_C2[3] * xz * sh[21 + ch] +
                                     kSH_C2[4] * (xx - yy) * sh[24 + ch];
                            if (v.D > 2) {
                                result = result + kSH_C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] + kSH_C3[1] * xy * z * sh[30 + ch] +
                                         kSH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                                         kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                                         kSH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + kSH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                                         kSH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                            }
                        }
                    }
                    result += 0.5f;
                    if (result < 0.f) clampbits |= 1u << ch;
                    rgb[ch] = fmaxf(result, 0.0f);
                }
            } else {
                rgb[0] = colors_precomp[3 * idx], rgb[1] = colors_precomp[3 * idx + 1], rgb[2] = colors_precomp[3 * idx + 2];
            }
            // surfel normal / camera-space point, hoisted from the blend loop (forward.cu:54-74, 779-785)
            const int naxis = arg_min3(sx, sy, sz), maxis = arg_max3(sx, sy, sz);
            const float nwx = Rm[0][naxis], nwy = Rm[1][naxis], nwz = Rm[2][naxis];
            const float smax = (maxis == 0 ? sx : (maxis == 1 ? sy : sz)) * v.scale_mod;
            const float ncx = view[0] * nwx + view[4] * nwy + view[8] * nwz;
            const float ncy = view[1] * nwx + view[5] * nwy + view[9] * nwz;
            const float ncz = view[2] * nwx + view[6] * nwy + view[10] * nwz;
            const float npc = tvx * ncx + tvy * ncy + tvz * ncz;

            radius = ir;
            g.conic_opacity[idx] = make_float4(conx, cony, conz, opacities[idx]);
            g.xy_depth[idx] = make_float4(pixx, pixy, tvz, __int_as_float(ir));
            g.rgb_smax[idx] = make_float4(rgb[0], rgb[1], rgb[2], smax);
            g.normal_c[idx] = make_float4(ncx, ncy, ncz, npc);
            g.point_c[idx] = make_float4(tvx, tvy, tvz, 0.f);
            g.clamped[idx] = (uint8_t)clampbits;
        } while (false);
        radii_out[idx] = radius;
        n_touched_out[idx] = 0;
        g.rect16[idx] = make_uint2((uint32_t)rminx | ((uint32_t)rmaxx << 16), (uint32_t)rminy | ((uint32_t)rmaxy << 16));
        nvis += radius > 0 ? 1u : 0u;
    }
    // visible count for the header (statistics only)
    if (nvis) atomicAdd(&s_visible, nvis);
    __syncthreads();
    if (tid == 0 && s_visible) atomicAdd(&g.counters[1], s_visible);
}

// ------------------------------------------------------------------------------------------------------------------
// One-block exclusive scan over the per-tile histogram: ranges, emit cursors, tile order, header.
// Replaces cub::DeviceScan + identifyTileRanges + the host compaction loop (rasterizer_impl.cu:303, 338-365).
// ------------------------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 1024;
constexpr int LPT_BUCKETS = 256;

__global__ __launch_bounds__(SCAN_THREADS) void tile_scan_kernel(int T, DqoImageLayout img, DqoGeomLayout g, int64_t capacity) {
    __shared__ uint32_t s_part[SCAN_THREADS / 64];
    __shared__ uint32_t s_part_act[SCAN_THREADS / 64];
    __shared__ uint32_t s_carry, s_carry_act, s_max;
    const int tid = threadIdx.x;
    const uint32_t lane = lane_id(), wave = tid >> 6;
    if (tid == 0) s_carry = 0, s_carry_act = 0, s_max = 0;
    __syncthreads();
    uint32_t local_max = 0;
    for (int base = 0; base < T; base += SCAN_THREADS) {
        const int t = base + tid;
        const uint32_t c = t < T ? img.tile_count[t] : 0u;
        const uint32_t a = c ? 1u : 0u;
        local_max = max(local_max, c);
        uint32_t incl = c, incl_a = a;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off), oa = __shfl_up(incl_a, off);
            if (lane >= (uint32_t)off) incl += o, incl_a += oa;
        }
        if (lane == 63) s_part[wave] = incl, s_part_act[wave] = incl_a;
        __syncthreads();
        uint32_t wbase = 0, wbase_a = 0;
        for (uint32_t w = 0; w < wave; w++) wbase += s_part[w], wbase_a += s_part_act[w];
        const uint32_t start = s_carry + wbase + incl - c;
        const uint32_t apos = s_carry_act + wbase_a + incl_a - a;
        if (t < T) {
            // clamp to capacity so later kernels never index past the binning buffer (overflow is flagged below)
            const uint32_t cs = (uint32_t)min((int64_t)start, capacity), ce = (uint32_t)min((int64_t)start + c, capacity);
            img.ranges[t] = make_uint2(c ? cs : 0u, c ? ce : 0u);  // empty tiles keep (0,0): rasterizer_impl.cu:338
            img.tile_cursor[t] = start;
            img.tile_walk[t] = 0;
            if (a) img.tile_order[apos] = (uint32_t)t;
        }
        __syncthreads();
        if (tid == SCAN_THREADS - 1) s_carry = start + c, s_carry_act = apos + a;
        __syncthreads();
    }
    atomicMax(&s_max, local_max);
    __syncthreads();
    // inactive tiles after the active ones (they still get a block: it writes the reference's initial fills)
    const uint32_t n_act = s_carry_act;
    for (int base = 0; base < T; base += SCAN_THREADS) {
        const int t = base + tid;
        const uint32_t c = t < T ? img.tile_count[t] : 1u;
        const uint32_t a = c ? 0u : 1u;
        uint32_t incl_a = a;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t oa = __shfl_up(incl_a, off);
            if (lane >= (uint32_t)off) incl_a += oa;
        }
        __syncthreads();
        if (lane == 63) s_part_act[wave] = incl_a;
        if (tid == 0 && base == 0) s_carry_act = 0;
        __syncthreads();
        uint32_t wbase_a = 0;
        for (uint32_t w = 0; w < wave; w++) wbase_a += s_part_act[w];
        const uint32_t ipos = s_carry_act + wbase_a + incl_a - a;
        if (t < T && a) img.tile_order[n_act + ipos] = (uint32_t)t;
        __syncthreads();
        if (tid == SCAN_THREADS - 1) s_carry_act = ipos + a;
        __syncthreads();
    }
    // ---- longest-processing-time-first launch order for the blend kernels: active tiles bucketed by list length (256
    // levels, descending) with an LDS counting sort — 3 barriers instead of a full sort.  Only the blockIdx -> tile mapping
    // changes: results do not depend on it. ----
    if (n_act > 1) {
        __shared__ uint32_t s_bucket[LPT_BUCKETS];
        const uint32_t mx = s_max;
        int shift = 0;
        while ((mx >> shift) >= (uint32_t)LPT_BUCKETS) shift++;
        for (int i = tid; i < LPT_BUCKETS; i += SCAN_THREADS) s_bucket[i] = 0;
        __syncthreads();
        for (int t = tid; t < T; t += SCAN_THREADS) {
            const uint32_t c = img.tile_count[t];
            if (c) atomicAdd(&s_bucket[LPT_BUCKETS - 1 - (c >> shift)], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // exclusive scan of the 256 bucket sizes by one wave (4 buckets per lane)
            uint32_t v4[LPT_BUCKETS / 64], sum = 0;
#pragma unroll
            for (int k = 0; k < LPT_BUCKETS / 64; k++) v4[k] = s_bucket[tid * (LPT_BUCKETS / 64) + k], sum += v4[k];
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off);
                if (tid >= off) incl += o;
            }
            uint32_t run = incl - sum;
#pragma unroll
            for (int k = 0; k < LPT_BUCKETS / 64; k++) s_bucket[tid * (LPT_BUCKETS / 64) + k] = run, run += v4[k];
        }
        __syncthreads();
        for (int t = tid; t < T; t += SCAN_THREADS) {
            const uint32_t c = img.tile_count[t];
            if (c) img.tile_order[atomicAdd(&s_bucket[LPT_BUCKETS - 1 - (c >> shift)], 1u)] = (uint32_t)t;
        }
    }
    if (tid == 0) {
        DqoRastHeader h;
        h.num_rendered = s_carry;
        h.num_tiles = n_act;
        h.overflow = ((int64_t)s_carry > capacity) ? 1u : 0u;
        h.max_tile_count = s_max;
        h.num_visible = g.counters[1];
        h.reserved[0] = h.reserved[1] = h.reserved[2] = 0;
        *g.header = h;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-tile sort of (depth, id) keys with the slot as payload.  Ascending-comparator bitonic network (works for any n
// without padding: elements past n behave as +inf and never move).  Lists up to SORT_LDS_CAP entries are sorted in
// LDS; longer ones in place in global memory by the same block (rare: correctness path, not a fast path).
// ------------------------------------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 256;
constexpr int SORT_LDS_CAP = 4096;

template <typename KeyPtr, typename ValPtr>
__device__ __forceinline__ void bitonic_any(KeyPtr keys, ValPtr vals, int n, int tid) {
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    const int half = n2 >> 1;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool flip = (j == (k >> 1));
            for (int t = tid; t < half; t += SORT_THREADS) {
                int i, p;
                if (flip) {
                    const int blk = t / j, off = t % j;
                    i = blk * k + off;
                    p = blk * k + k - 1 - off;
                } else {
                    i = 2 * j * (t / j) + (t % j);
                    p = i + j;
                }
                if (p < n) {
                    const uint64_t a = keys[i], b = keys[p];
                    if (a > b) {
                        keys[i] = b;
                        keys[p] = a;
                        const uint32_t va = vals[i];
                        vals[i] = vals[p];
                        vals[p] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(SORT_THREADS) void tile_sort_kernel(DqoImageLayout img, DqoBinLayout bin) {
    __shared__ uint64_t s_keys[SORT_LDS_CAP];
    __shared__ uint32_t s_vals[SORT_LDS_CAP];
    const int tile = img.tile_order[blockIdx.x];
    const uint2 rg = img.ranges[tile];
    const int n = (int)(rg.y - rg.x);
    const int tid = threadIdx.x;
    if (n <= 0) return;
    uint64_t* gk = bin.keys + rg.x;
    uint32_t* gv = bin.slots + rg.x;
    if (n <= SORT_LDS_CAP) {
        for (int i = tid; i < n; i += SORT_THREADS) s_keys[i] = gk[i], s_vals[i] = gv[i];
        __syncthreads();
        bitonic_any(s_keys, s_vals, n, tid);
        for (int i = tid; i < n; i += SORT_THREADS) {
            bin.point_list[rg.x + i] = (uint32_t)(s_keys[i] & 0xffffffffu);
            bin.slot_list[rg.x + i] = s_vals[i];
        }
    } else {
        __syncthreads();
        bitonic_any(gk, gv, n, tid);  // same block wrote / reads: __syncthreads orders global accesses within the block
        for (int i = tid; i < n; i += SORT_THREADS) {
            bin.point_list[rg.x + i] = (uint32_t)(gk[i] & 0xffffffffu);
            bin.slot_list[rg.x + i] = gv[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K6: forward blend.  One 16x16 tile per 256-thread block = 4 waves, each wave an 8x8 pixel quadrant so that whole
// waves drop out of an entry (ballot) when the splat misses their quadrant.  Per-entry data is staged in LDS as
// three 16-byte records and read back as broadcasts.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BLEND_THREADS = 256;

__device__ __forceinline__ float3 pixel_ray(uint32_t px, uint32_t py, float fx, float fy, float cx, float cy) {
#pragma clang fp contract(off)
    // ndc2ray, forward.cu:92-100
    float rx = ((float)px - cx) / fx, ry = ((float)py - cy) / fy, rz = 1.0f;
    const float n = 1.0f / sqrtf(rx * rx + ry * ry + rz * rz);
    return make_float3(rx * n, ry * n, rz * n);
}

// Ray / surfel-plane intersection of forward.cu:784-791 with its literal mixed precision: float numerator and
// denominator, `+ 1e-8` and the division in double.
struct HitEval {
    float t, den, hit_z;
};
__device__ __forceinline__ HitEval eval_hit(const float3 ray, const float4 n_np) {
#pragma clang fp contract(off)
    HitEval h;
    h.den = ray.x * n_np.x + ray.y * n_np.y + ray.z * n_np.z;
    h.t = (float)((double)n_np.w / ((double)h.den + 1e-8));
    h.hit_z = h.t * ray.z;
    return h;
}

__global__ __launch_bounds__(BLEND_THREADS) void blend_forward_kernel(const DqoView v, DqoGeomLayout g, DqoImageLayout img,
                                                                      DqoBinLayout bin, DqoRastOutputs out) {
    __shared__ float4 s_co[BLEND_THREADS];
    __shared__ float4 s_xy[BLEND_THREADS];
    __shared__ float4 s_rgb[BLEND_THREADS];
    __shared__ int s_id[BLEND_THREADS];
    __shared__ int s_cnt[BLEND_THREADS];
    __shared__ uint32_t s_qmask[BLEND_THREADS];
    __shared__ uint8_t s_list[4][BLEND_THREADS];
    __shared__ uint32_t s_walk;

    const int tile = img.tile_order[blockIdx.x];
    const int tile_x = tile % v.gx, tile_y = tile / v.gx;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const uint32_t px = tile_x * DQO_TILE + (wave & 1) * 8 + (lane & 7);
    const uint32_t py = tile_y * DQO_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < (uint32_t)v.W && py < (uint32_t)v.H;
    const size_t HW = (size_t)v.W * v.H;
    const size_t pix_id = (size_t)v.W * py + px;
    const uint2 range = img.ranges[tile];
    const int n = (int)(range.y - range.x);

    if (n == 0) {
        // masked or empty tile: the reference's torch::full initial values (rasterize_points.cu:79-89).  A tile that is
        // active in the reference but whose instances were all culled as dead is rendered with an empty list instead:
        // colour = bg, ids = -1 (forward.cu:724-725, 852-860).
        const bool rendered = img.tile_flag[tile] != 0u;
        if (inside) {
            out.out_color[pix_id] = rendered ? v.bg[0] : 0.f;
            out.out_color[HW + pix_id] = rendered ? v.bg[1] : 0.f;
            out.out_color[2 * HW + pix_id] = rendered ? v.bg[2] : 0.f;
            out.out_depth[pix_id] = 0.f;
            out.out_hit_depth[pix_id] = rendered ? -1 : 0;
            out.out_hit_color[pix_id] = rendered ? -1 : 0;
            out.out_hit_color_weight[pix_id] = 0.f;
            out.out_hit_depth_weight[pix_id] = 0.f;
            out.out_T[pix_id] = 1.f;
            img.final_T[pix_id] = 1.f;
            img.n_contrib[pix_id] = 0;
            img.hit_pos[pix_id] = 0;
        }
        return;
    }

    const float pixfx = (float)px, pixfy = (float)py;
    const float3 ray = pixel_ray(px, py, v.focal_x, v.focal_y, v.cx, v.cy);
    bool done = !inside;
    float T = 1.0f, end_T = 1.0f;
    uint32_t last_contributor = 0, hit_pos = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    float depth_ = 0.f;
    bool hit_gaussian = false;
    int hit_id = -1, hit_color_id = -1;
    float color_weight_max = -1.f, hit_color_weight = 0.f, hit_depth_weight = 0.f;

    s_cnt[tid] = 0;
    int flushed_upto = 0;  // index of the batch currently held in LDS
    if (tid == 0) s_walk = 0;
    const int rounds = (n + BLEND_THREADS - 1) / BLEND_THREADS;
    int toDo = n;
    for (int i = 0; i < rounds; i++, toDo -= BLEND_THREADS) {
        // also orders the previous round's LDS reads before this round's staging writes
        if (__syncthreads_and(done)) break;
        // flush the previous batch: n_touched counts (forward.cu:833-835: one count per pair with T' > 0.5) and the
        // 4-bit live mask (which quadrants had a lane this entry acted on) that lets the backward skip dead pairs
        if (i > 0) {
            const uint32_t wd = (uint32_t)s_cnt[tid];
            const int prev = (i - 1) * BLEND_THREADS + tid;
            if (prev < n) bin.live[range.x + prev] = (uint8_t)(wd >> 28);
            if (wd & 0x0fffffffu) atomicAdd(&out.n_touched[s_id[tid]], (int)(wd & 0x0fffffffu));
            s_cnt[tid] = 0;
        }
        flushed_upto = i;
        __syncthreads();
        const int progress = i * BLEND_THREADS + tid;
        if (progress < n) {
            const int id = (int)bin.point_list[range.x + progress];
            const float4 co = g.conic_opacity[id];
            const float4 xy = g.xy_depth[id];
            s_id[tid] = id;
            s_co[tid] = co;
            s_xy[tid] = xy;
            s_rgb[tid] = g.rgb_smax[id];
            // which of the four 8x8 quadrants (= waves) the splat can reach at all (dqo_cull.h)
            const float qthr = dqo_q_threshold(co.w);
            uint32_t qm = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float x0 = (float)(tile_x * DQO_TILE + (q & 1) * 8), y0 = (float)(tile_y * DQO_TILE + (q >> 1) * 8);
                qm |= dqo_splat_hits_rect(xy.x, xy.y, co.x, co.y, co.z, qthr, x0, y0, x0 + 7.f, y0 + 7.f) ? (1u << q) : 0u;
            }
            s_qmask[tid] = qm;
        }
        __syncthreads();
        const int batch = min(BLEND_THREADS, toDo);
        // per-wave compaction: the entries of this batch that can reach this wave's quadrant, in list order
        int cnt = 0;
#pragma unroll
        for (int base = 0; base < BLEND_THREADS; base += 64) {
            const int jj = base + lane;
            const bool lv = jj < batch && ((s_qmask[jj] >> wave) & 1u);
            const unsigned long long mm = __ballot(lv);
            if (lv) s_list[wave][cnt + (int)__popcll(mm & ((1ull << lane) - 1ull))] = (uint8_t)jj;
            cnt += (int)__popcll(mm);
        }
        // software-pipelined walk: the next entry's records are fetched from LDS while the current one is blended
        int j = cnt > 0 ? (int)s_list[wave][0] : 0;
        float4 xy = s_xy[j], co = s_co[j];
        for (int k = 0; k < cnt; k++) {
            if (__ballot(!done) == 0) break;  // whole wave finished
            const int j_cur = j;
            const float4 xy_cur = xy, co_cur = co;
            if (k + 1 < cnt) {
                j = (int)s_list[wave][k + 1];
                xy = s_xy[j];
                co = s_co[j];
            }
            // ---- predicated (branch-free) per-pixel update: nested divergent ifs cost more scalar exec-mask traffic than
            // the arithmetic they guard; the only real branches left are wave-uniform ----
            const uint32_t contributor = (uint32_t)(i * BLEND_THREADS + j_cur + 1);  // the reference's running counter
            const float dx = xy_cur.x - pixfx, dy = xy_cur.y - pixfy;
            const float power = -0.5f * (co_cur.x * dx * dx + co_cur.z * dy * dy) - co_cur.y * dx * dy;
            const float alpha = fminf(0.99f, co_cur.w * dqo_gauss(power));
            const bool valid = !done && power <= 0.0f && alpha >= 1.0f / 255.0f;  // forward.cu:763-772
            if (__ballot(valid) == 0) continue;  // no pixel of this quadrant is touched by the entry
            const float4 cs = s_rgb[j_cur];
            const bool newhit = valid && !hit_gaussian && alpha >= v.opaque_thr;
            if (__ballot(newhit)) {
                // forward.cu:792-810: first Gaussian with alpha >= opaque_threshold fixes this pixel's depth (once per pixel)
                if (newhit) {
                    const int id = s_id[j_cur];
                    const HitEval h = eval_hit(ray, g.normal_c[id]);
                    hit_id = id;
                    hit_pos = contributor;
                    hit_depth_weight = alpha * T;
                    const float angle_distance = fabsf(h.den);
                    const float depth_distance = fabsf(h.hit_z - xy_cur.z);
                    depth_ = (depth_distance <= cs.w * v.depth_thr && angle_distance >= v.normal_thr) ? h.hit_z : xy_cur.z;
                    hit_gaussian = true;
                }
            }
            const float test_T = T * (1.f - alpha);
            const bool finish = valid && test_T < v.T_thr && hit_gaussian;   // forward.cu:813-817: done, T NOT updated
            const bool blend = valid && !finish && test_T >= v.T_thr;         // forward.cu:818-840
            const float w = blend ? alpha * T : 0.f;
            C0 += cs.x * w;
            C1 += cs.y * w;
            C2 += cs.z * w;
            const bool newmax = blend && w > color_weight_max;
            color_weight_max = newmax ? w : color_weight_max;
            hit_color_id = newmax ? s_id[j_cur] : hit_color_id;
            hit_color_weight = newmax ? w : hit_color_weight;
            last_contributor = blend ? contributor : last_contributor;
            end_T = blend ? test_T : end_T;
            T = (valid && !finish) ? test_T : T;  // keeps decaying below T_thr until an opaque hit appears (forward.cu:841)
            done = done || finish;
            const bool contributes_half = blend && test_T > 0.5f;  // forward.cu:833-835 (B8)
            const bool lane_live = blend || newhit;
            // one LDS atomic per (wave, entry): low 28 bits count the T' > 0.5 pairs, bit 28+wave marks the quadrant live
            const unsigned long long m = __ballot(contributes_half);
            const unsigned long long lv = __ballot(lane_live);
            if (lv && lane == 0) atomicAdd(&s_cnt[j_cur], (int)((uint32_t)__popcll(m) | (1u << (28 + wave))));
        }
    }
    __syncthreads();
    {
        // the last batch that was staged (index flushed_upto) has not been flushed yet
        const uint32_t wd = (uint32_t)s_cnt[tid];
        const int prev = flushed_upto * BLEND_THREADS + tid;
        if (prev < n) bin.live[range.x + prev] = (uint8_t)(wd >> 28);
        if (wd & 0x0fffffffu) atomicAdd(&out.n_touched[s_id[tid]], (int)(wd & 0x0fffffffu));
    }
    if (inside) {
        const float b0 = v.bg[0], b1 = v.bg[1], b2 = v.bg[2];
        img.final_T[pix_id] = end_T;
        img.n_contrib[pix_id] = last_contributor;
        img.hit_pos[pix_id] = hit_pos;
        out.out_color[pix_id] = C0 + T * b0;  // running T, not end_T (quirk B2, forward.cu:852)
        out.out_color[HW + pix_id] = C1 + T * b1;
        out.out_color[2 * HW + pix_id] = C2 + T * b2;
        out.out_depth[pix_id] = depth_;
        out.out_hit_depth[pix_id] = hit_id;
        out.out_hit_color[pix_id] = hit_color_id;
        out.out_hit_color_weight[pix_id] = hit_color_weight;
        out.out_hit_depth_weight[pix_id] = hit_depth_weight;
        out.out_T[pix_id] = end_T;
    }
    // entries the backward has to walk for this tile
    uint32_t w = inside ? max(last_contributor, hit_pos) : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w = max(w, (uint32_t)__shfl_xor((int)w, off));
    if (lane == 0) atomicMax(&s_walk, w);
    __syncthreads();
    if (tid == 0) img.tile_walk[tile] = s_walk;
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    const float* __restrict__ proj, uint8_t* __restrict__ present) {
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    const float hx = proj[0] * px + proj[4] * py + proj[8] * pz + proj[12];
    const float hy = proj[1] * px + proj[5] * py + proj[9] * pz + proj[13];
    const float hw = proj[3] * px + proj[7] * py + proj[11] * pz + proj[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float projx = hx * p_w, projy = hy * p_w;
    const float tvz = view[2] * px + view[6] * py + view[10] * pz + view[14];
    present[idx] = !(tvz <= 0.2f || (double)projx < -1.3 || (double)projx > 1.3 || (double)projy < -1.3 || (double)projy > 1.3);
}

}  // namespace

int dqo_launch_bin_count(int P, int gx, const int32_t* tile_mask, const DqoGeomLayout& g, const DqoImageLayout& img, hipStream_t s);
int dqo_launch_bin_emit(int P, int gx, const int32_t* tile_mask, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                        int64_t capacity, hipStream_t s);

int dqo_launch_forward_prepare(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, hipStream_t s) {
    const DqoView v = dqo_make_view(p, in);
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    DqoImageLayout img = dqo_image_layout(ctx->image, p->W, p->H);
    const int T = v.gx * v.gy;
    DQO_CHECK_HIP(hipMemsetAsync(g.header, 0, 512, s));  // header + counters
    DQO_CHECK_HIP(hipMemsetAsync(img.tile_count, 0, (size_t)((char*)(img.tile_flag + T) - (char*)img.tile_count), s));
    if (p->P > 0) {
        const int per_block = K1_THREADS * K1_ITEMS;
        const int grid = (p->P + per_block - 1) / per_block;
        DQO_LAUNCH("preprocess_kernel", preprocess_kernel, dim3(grid), dim3(K1_THREADS), s, v, in->means3D, in->scales, in->rotations,
                           in->opacities, in->shs, in->colors_precomp, in->tile_mask, g, out->radii, out->n_touched);
        // per-tile histogram, tiles_touched, gaussian-major slots (forward.cu:344-353 + the cub scan of rasterizer_impl.cu:303)
        int rc = dqo_launch_bin_count(p->P, v.gx, in->tile_mask, g, img, s);
        if (rc) return rc;
    }
    return DQO_OK;
}

int dqo_launch_forward_render(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, hipStream_t s) {
    const DqoView v = dqo_make_view(p, in);
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    DqoImageLayout img = dqo_image_layout(ctx->image, p->W, p->H);
    DqoBinLayout bin = dqo_bin_layout(ctx->binning, ctx->inst_capacity);
    const int T = v.gx * v.gy;
    DQO_LAUNCH("tile_scan_kernel", tile_scan_kernel, dim3(1), dim3(SCAN_THREADS), s, T, img, g, (int64_t)ctx->inst_capacity);
    if (p->P > 0) {
        int rc = dqo_launch_bin_emit(p->P, v.gx, in->tile_mask, g, img, bin, (int64_t)ctx->inst_capacity, s);
        if (rc) return rc;
        DQO_LAUNCH("tile_sort_kernel", tile_sort_kernel, dim3(T), dim3(SORT_THREADS), s, img, bin);
    }
    DQO_LAUNCH("blend_forward_kernel", blend_forward_kernel, dim3(T), dim3(BLEND_THREADS), s, v, g, img, bin, *out);
    return DQO_OK;
}

int dqo_launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present, hipStream_t s) {
    if (P <= 0) return DQO_OK;
    DQO_LAUNCH("mark_visible_kernel", mark_visible_kernel, dim3((P + 255) / 256), dim3(256), s, P, means3D, view, proj, present);
    return DQO_OK;
}
