// Backward pass of the depth-aware Gaussian rasteriser for gfx950 (MI355X).
//
// Replaces (semantics, not structure) /root/reference/submodules/diff-gaussian-rasterizer-depth/
//   cuda_rasterizer/backward.cu:808-1066  renderCUDA_flat (+propagateRotationGrad :100-148)  -> blend_backward_kernel
//   cuda_rasterizer/backward.cu:273-422   computeCov2DCUDA                                    \
//   cuda_rasterizer/backward.cu:492-548   preprocessCUDA (+SH bwd :152-268, cov3D bwd :426-487)/ -> gaussian_backward_kernel
//
// MI355X design: the reference issues ~10 scattered global float atomics per (pixel, Gaussian) pair plus 3-7 per hit
// pixel.  On gfx950 global float atomics execute memory-side and a wave instruction whose 64 lanes hit 64 different rows
// runs ~17x below the streaming rate (MI355X_MICROARCH.md, Global float atomics), so none are used here:
//   * one wave64 owns a whole 16x16 tile (4 pixels per lane); every lane looks at the same list entry in the same
//     trip, so the per-entry gradient is a register sum over the lane's 4 pixels followed by one DPP wave reduction;
//   * the wave stores ONE 64-byte record per (tile, Gaussian) instance at the instance's gaussian-major slot;
//   * gaussian_backward_kernel sums each Gaussian's contiguous records in a fixed order (bitwise reproducible) and
//     finishes the chain rule (cov2D, projection, SH, cov3D) in the same pass.
#include "dqo_common.h"
#include "dqo_cull.h"

namespace {

// ---- wave64 sum via DPP (no LDS traffic): 4 steps inside each 16-lane row, 2 row broadcasts, result in lane 63 ----
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);       // row_half_mirror
    v += dpp_mov<0x140>(v);       // row_mirror          -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
    v += dpp_mov<0x143, 0xC>(v);  // row_bcast31 into rows 2,3 -> lanes 48..63 hold the wave total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float3 pixel_ray_b(uint32_t px, uint32_t py, float fx, float fy, float cx, float cy) {
#pragma clang fp contract(off)
    float rx = ((float)px - cx) / fx, ry = ((float)py - cy) / fy, rz = 1.0f;
    const float n = 1.0f / sqrtf(rx * rx + ry * ry + rz * rz);
    return make_float3(rx * n, ry * n, rz * n);
}

constexpr int BWD_THREADS = 64;  // one wave per tile
constexpr int PPL = 4;           // pixels per lane (the four 8x8 quadrants of the tile)

__global__ __launch_bounds__(BWD_THREADS) void blend_backward_kernel(const DqoView v, DqoGeomLayout g, DqoImageLayout img,
                                                                     DqoBinLayout bin, const float* __restrict__ scales,
                                                                     const float* __restrict__ rotations,
                                                                     const float* __restrict__ dL_dpixels,
                                                                     const float* __restrict__ dL_ddepths,
                                                                     DqoGradRec* __restrict__ recs, int64_t capacity) {
    __shared__ float4 s_co[BWD_THREADS];
    __shared__ float4 s_xy[BWD_THREADS];
    __shared__ float4 s_rgb[BWD_THREADS];
    __shared__ int s_id[BWD_THREADS];
    __shared__ uint32_t s_slot[BWD_THREADS];
    __shared__ uint32_t s_qmask[BWD_THREADS];
    __shared__ float4 s_rec[BWD_THREADS * 4];  // 64 records x 64 B

    const int tile = img.tile_order[blockIdx.x];
    const uint2 range = img.ranges[tile];
    const int n = (int)(range.y - range.x);
    if (n == 0) return;
    const int L = min((int)img.tile_walk[tile], n);
    const int lane = threadIdx.x;
    const int tile_x = tile % v.gx, tile_y = tile / v.gx;
    const size_t HW = (size_t)v.W * v.H;

    float view[12];
#pragma unroll
    for (int i = 0; i < 12; i++) view[i] = v.view[i];
    const float bg0 = v.bg[0], bg1 = v.bg[1], bg2 = v.bg[2];
    const float ddelx_dx = 0.5f * v.W, ddely_dy = 0.5f * v.H;

    // per-pixel state, 4 pixels per lane
    float pixfx[PPL], pixfy[PPL];
    float T[PPL], T_final[PPL];
    int last_contrib[PPL], hit_pos[PPL];
    float acc0[PPL], acc1[PPL], acc2[PPL], lc0[PPL], lc1[PPL], lc2[PPL], last_alpha[PPL];
    float dp0[PPL], dp1[PPL], dp2[PPL], ddep[PPL], bgdot[PPL];
    uint32_t pxs[PPL], pys[PPL];
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        const uint32_t px = tile_x * DQO_TILE + (q & 1) * 8 + (lane & 7);
        const uint32_t py = tile_y * DQO_TILE + (q >> 1) * 8 + (lane >> 3);
        const bool inside = px < (uint32_t)v.W && py < (uint32_t)v.H;
        const size_t pid = (size_t)v.W * py + px;
        pxs[q] = px, pys[q] = py;
        pixfx[q] = (float)px, pixfy[q] = (float)py;
        T_final[q] = inside ? img.final_T[pid] : 0.f;
        T[q] = T_final[q];
        last_contrib[q] = inside ? (int)img.n_contrib[pid] : 0;
        hit_pos[q] = inside ? (int)img.hit_pos[pid] : 0;
        dp0[q] = inside ? dL_dpixels[pid] : 0.f;
        dp1[q] = inside ? dL_dpixels[HW + pid] : 0.f;
        dp2[q] = inside ? dL_dpixels[2 * HW + pid] : 0.f;
        ddep[q] = inside ? dL_ddepths[pid] : 0.f;
        bgdot[q] = bg0 * dp0[q] + bg1 * dp1[q] + bg2 * dp2[q];
        acc0[q] = acc1[q] = acc2[q] = 0.f;
        lc0[q] = lc1[q] = lc2[q] = 0.f;
        last_alpha[q] = 0.f;
    }

    // entries [L, n) were never reached by any pixel of the tile: zero records
    for (int e = L * 4 + lane; e < n * 4; e += BWD_THREADS) {
        const uint32_t slot = bin.slot_list[range.x + (e >> 2)];
        if ((int64_t)slot < capacity) reinterpret_cast<float4*>(recs + slot)[e & 3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const int rounds = (L + BWD_THREADS - 1) / BWD_THREADS;
    for (int b = 0; b < rounds; b++) {
        // batch b covers list positions L-1-b*64 ... (descending); lane l stages position L-1-(b*64+l)
        __syncthreads();
        const int pos = L - 1 - (b * BWD_THREADS + lane);
        if (pos >= 0) {
            const int id = (int)bin.point_list[range.x + pos];
            const float4 co = g.conic_opacity[id];
            const float4 xy = g.xy_depth[id];
            s_id[lane] = id;
            s_slot[lane] = bin.slot_list[range.x + pos];
            s_co[lane] = co;
            s_xy[lane] = xy;
            s_rgb[lane] = g.rgb_smax[id];
            // quadrants (= the lane's four pixels) the splat can reach at all (dqo_cull.h)
            const float qthr = dqo_q_threshold(co.w);
            uint32_t qm = 0;
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                const float x0 = (float)(tile_x * DQO_TILE + (q & 1) * 8), y0 = (float)(tile_y * DQO_TILE + (q >> 1) * 8);
                qm |= dqo_splat_hits_rect(xy.x, xy.y, co.x, co.y, co.z, qthr, x0, y0, x0 + 7.f, y0 + 7.f) ? (1u << q) : 0u;
            }
            s_qmask[lane] = qm;
        }
        __syncthreads();
        const int batch = min(BWD_THREADS, L - b * BWD_THREADS);
        for (int j = 0; j < batch; j++) {
            const int c = L - 1 - (b * BWD_THREADS + j);  // 0-based list position; the reference's `contributor` after --
            const float4 co = s_co[j];
            const float4 xy = s_xy[j];
            const float4 cs = s_rgb[j];
            float r_c0 = 0.f, r_c1 = 0.f, r_c2 = 0.f, r_mx = 0.f, r_my = 0.f, r_ka = 0.f, r_kb = 0.f, r_kc = 0.f, r_op = 0.f;
            bool any_color = false, any_hit = false;
            const uint32_t qm = s_qmask[j];
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                any_hit |= (hit_pos[q] == c + 1);
                if (!((qm >> q) & 1u)) continue;  // wave-uniform: quadrant q is out of the splat's reach
                if (c < last_contrib[q]) {
                    const float dx = xy.x - pixfx[q], dy = xy.y - pixfy[q];
                    const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
                    if (power <= 0.0f) {
                        const float G = expf(power);
                        const float alpha = fminf(0.99f, co.w * G);
                        if (alpha >= 1.0f / 255.0f) {
                            any_color = true;
                            T[q] = T[q] / (1.f - alpha);
                            const float dchannel_dcolor = alpha * T[q];
                            const float la = last_alpha[q];
                            acc0[q] = la * lc0[q] + (1.f - la) * acc0[q];
                            acc1[q] = la * lc1[q] + (1.f - la) * acc1[q];
                            acc2[q] = la * lc2[q] + (1.f - la) * acc2[q];
                            lc0[q] = cs.x, lc1[q] = cs.y, lc2[q] = cs.z;
                            float dL_dalpha = (cs.x - acc0[q]) * dp0[q] + (cs.y - acc1[q]) * dp1[q] + (cs.z - acc2[q]) * dp2[q];
                            r_c0 += dchannel_dcolor * dp0[q];
                            r_c1 += dchannel_dcolor * dp1[q];
                            r_c2 += dchannel_dcolor * dp2[q];
                            dL_dalpha *= T[q];
                            last_alpha[q] = alpha;
                            dL_dalpha += (-T_final[q] / (1.f - alpha)) * bgdot[q];
                            const float dL_dG = co.w * dL_dalpha;
                            const float gdx = G * dx, gdy = G * dy;
                            const float dG_ddelx = -gdx * co.x - gdy * co.y;
                            const float dG_ddely = -gdy * co.z - gdx * co.y;
                            r_mx += dL_dG * dG_ddelx * ddelx_dx;
                            r_my += dL_dG * dG_ddely * ddely_dy;
                            r_ka += -0.5f * gdx * dx * dL_dG;
                            r_kb += -0.5f * gdx * dy * dL_dG;
                            r_kc += -0.5f * gdy * dy * dL_dG;
                            r_op += G * dL_dalpha;
                        }
                    }
                }
            }
            float t_c0 = 0.f, t_c1 = 0.f, t_c2 = 0.f, t_mx = 0.f, t_my = 0.f, t_ka = 0.f, t_kb = 0.f, t_kc = 0.f, t_op = 0.f;
            if (__ballot(any_color)) {
                t_c0 = wave_sum(r_c0);
                t_c1 = wave_sum(r_c1);
                t_c2 = wave_sum(r_c2);
                t_mx = wave_sum(r_mx);
                t_my = wave_sum(r_my);
                t_ka = wave_sum(r_ka);
                t_kb = wave_sum(r_kb);
                t_kc = wave_sum(r_kc);
                t_op = wave_sum(r_op);
            }
            float t_m0 = 0.f, t_m1 = 0.f, t_m2 = 0.f, t_q0 = 0.f, t_q1 = 0.f, t_q2 = 0.f, t_q3 = 0.f;
            if (__ballot(any_hit)) {
                // hit-Gaussian depth gradient, backward.cu:997-1065 (once per pixel, for the entry that fixed its depth)
                float h_m0 = 0.f, h_m1 = 0.f, h_m2 = 0.f, h_q0 = 0.f, h_q1 = 0.f, h_q2 = 0.f, h_q3 = 0.f;
                const int id = s_id[j];
                const float4 n_np = g.normal_c[id];
                const float4 pc = g.point_c[id];
                const float sx = scales[3 * id], sy = scales[3 * id + 1], sz = scales[3 * id + 2];
                const float4 qt = reinterpret_cast<const float4*>(rotations)[id];
                const float scale_max = fmaxf(fmaxf(sx, sy), sz);  // raw scales, quirk B6 (backward.cu:1009)
                const int axis = (sx <= sy && sx <= sz) ? 0 : ((sy <= sx && sy <= sz) ? 1 : 2);
#pragma unroll
                for (int q = 0; q < PPL; q++) {
                    if (hit_pos[q] == c + 1) {
#pragma clang fp contract(off)
                        const float3 ray = pixel_ray_b(pxs[q], pys[q], v.focal_x, v.focal_y, v.cx, v.cy);
                        const float nr_f = n_np.x * ray.x + n_np.y * ray.y + n_np.z * ray.z;
                        // hit_point.z as stored by the forward (forward.cu:784-786, 808)
                        const float den_f = ray.x * n_np.x + ray.y * n_np.y + ray.z * n_np.z;
                        const float t = (float)((double)n_np.w / ((double)den_f + 1e-8));
                        const float hit_z = t * ray.z;
                        const float angle_distance = fabsf(nr_f);
                        const float depth_distance = fabsf(hit_z - pc.z);
                        const float dL_ddi = ddep[q];
                        if (depth_distance <= v.depth_thr * scale_max && angle_distance >= v.normal_thr) {
                            const float nr = (float)((double)nr_f + 1e-8);
                            const float inv_nr = 1.f / nr, inv_nr2 = inv_nr * inv_nr;
                            const float np = n_np.x * pc.x + n_np.y * pc.y + n_np.z * pc.z;
                            const float dpx = ray.z * n_np.x * inv_nr, dpy = ray.z * n_np.y * inv_nr, dpz = ray.z * n_np.z * inv_nr;
                            h_m0 += dL_ddi * (dpx * view[0] + dpy * view[1] + dpz * view[2]);
                            h_m1 += dL_ddi * (dpx * view[4] + dpy * view[5] + dpz * view[6]);
                            h_m2 += dL_ddi * (dpx * view[8] + dpy * view[9] + dpz * view[10]);
                            const float n1c = ray.z * (nr * pc.x - np * ray.x) * inv_nr2;
                            const float n2c = ray.z * (nr * pc.y - np * ray.y) * inv_nr2;
                            const float n3c = ray.z * (nr * pc.z - np * ray.z) * inv_nr2;
                            const float n1w = n1c * view[0] + n2c * view[1] + n3c * view[2];
                            const float n2w = n1c * view[4] + n2c * view[5] + n3c * view[6];
                            const float n3w = n1c * view[8] + n2c * view[9] + n3c * view[10];
                            // propagateRotationGrad, backward.cu:100-148: d(column `axis` of R(q)) / dq
                            const float q0 = qt.x, q1 = qt.y, q2 = qt.z, q3 = qt.w;
                            float d0[3], d1[3], d2[3], d3[3];
                            if (axis == 0) {
                                d0[0] = 0, d0[1] = 2 * q3, d0[2] = -2 * q2;
                                d1[0] = 0, d1[1] = 2 * q2, d1[2] = 2 * q3;
                                d2[0] = -4 * q2, d2[1] = 2 * q1, d2[2] = -2 * q0;
                                d3[0] = -4 * q3, d3[1] = 2 * q0, d3[2] = 2 * q1;
                            } else if (axis == 1) {
                                d0[0] = -2 * q3, d0[1] = 0, d0[2] = 2 * q1;
                                d1[0] = 2 * q2, d1[1] = -4 * q1, d1[2] = 2 * q0;
                                d2[0] = 2 * q1, d2[1] = 0, d2[2] = 2 * q3;
                                d3[0] = -2 * q0, d3[1] = -4 * q3, d3[2] = 2 * q2;
                            } else {
                                d0[0] = 2 * q2, d0[1] = -2 * q1, d0[2] = 0;
                                d1[0] = 2 * q3, d1[1] = -2 * q0, d1[2] = -4 * q1;
                                d2[0] = 2 * q0, d2[1] = 2 * q3, d2[2] = -4 * q2;
                                d3[0] = 2 * q1, d3[1] = 2 * q2, d3[2] = 0;
                            }
                            h_q0 += dL_ddi * (n1w * d0[0] + n2w * d0[1] + n3w * d0[2]);
                            h_q1 += dL_ddi * (n1w * d1[0] + n2w * d1[1] + n3w * d1[2]);
                            h_q2 += dL_ddi * (n1w * d2[0] + n2w * d2[1] + n3w * d2[2]);
                            h_q3 += dL_ddi * (n1w * d3[0] + n2w * d3[1] + n3w * d3[2]);
                        } else {
                            h_m0 += dL_ddi * view[2];
                            h_m1 += dL_ddi * view[6];
                            h_m2 += dL_ddi * view[10];
                        }
                    }
                }
                t_m0 = wave_sum(h_m0);
                t_m1 = wave_sum(h_m1);
                t_m2 = wave_sum(h_m2);
                t_q0 = wave_sum(h_q0);
                t_q1 = wave_sum(h_q1);
                t_q2 = wave_sum(h_q2);
                t_q3 = wave_sum(h_q3);
            }
            if (lane == 0) {
                s_rec[j * 4 + 0] = make_float4(t_c0, t_c1, t_c2, t_mx);
                s_rec[j * 4 + 1] = make_float4(t_my, t_ka, t_kb, t_kc);
                s_rec[j * 4 + 2] = make_float4(t_op, t_m0, t_m1, t_m2);
                s_rec[j * 4 + 3] = make_float4(t_q0, t_q1, t_q2, t_q3);
            }
        }
        __syncthreads();
        // write the batch's records: 4 lanes per 64-byte record, each record one contiguous row
        for (int e = lane; e < batch * 4; e += BWD_THREADS) {
            const uint32_t slot = s_slot[e >> 2];
            if ((int64_t)slot < capacity) reinterpret_cast<float4*>(recs + slot)[e & 3] = s_rec[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-Gaussian backward: sum the instance records, then K8 (cov2D) + K9 (projection, SH, cov3D) in one pass.
// Writes every gradient element (zeros for culled Gaussians), so the caller's tensors can be torch.empty.
// ------------------------------------------------------------------------------------------------------------------
__constant__ float bSH_C0 = 0.28209479177387814f;
__constant__ float bSH_C1 = 0.4886025119029199f;
__constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

__global__ __launch_bounds__(256) void gaussian_backward_kernel(const DqoView v, DqoGeomLayout g,
                                                                const float* __restrict__ means3D, const float* __restrict__ scales,
                                                                const float* __restrict__ rotations, const float* __restrict__ shs,
                                                                const DqoGradRec* __restrict__ recs, DqoRastGrads gr) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= v.P) return;
    const int M = v.M, D = v.D;
    float* dm = gr.dL_dmeans3D + 3 * (size_t)idx;
    float* dsh = gr.dL_dsh ? gr.dL_dsh + (size_t)idx * M * 3 : nullptr;
    float* dcol = gr.dL_dcolors + 3 * (size_t)idx;
    float* dsc = gr.dL_dscales + 3 * (size_t)idx;
    float* drot = gr.dL_drotations + 4 * (size_t)idx;
    float* dcov = gr.dL_dcov3D + 6 * (size_t)idx;
    float* dm2 = gr.dL_dmeans2D + 3 * (size_t)idx;
    // radii > 0 (backward.cu:285, 513)  <=>  the forward kept a non-empty tile rect for this Gaussian
    const uint2 rc = g.rect16[idx];
    const bool visible = ((rc.x >> 16) > (rc.x & 0xffffu)) && ((rc.y >> 16) > (rc.y & 0xffffu));
    if (!visible) {
        dm[0] = dm[1] = dm[2] = 0.f;
        if (dsh)
            for (int i = 0; i < 3 * M; i++) dsh[i] = 0.f;
        dcol[0] = dcol[1] = dcol[2] = 0.f;
        gr.dL_dopacity[idx] = 0.f;
        dsc[0] = dsc[1] = dsc[2] = 0.f;
        drot[0] = drot[1] = drot[2] = drot[3] = 0.f;
        for (int i = 0; i < 6; i++) dcov[i] = 0.f;
        dm2[0] = dm2[1] = dm2[2] = 0.f;
        return;
    }
    // ---- fixed-order sum of this Gaussian's instance records ----
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = 0.f;
    {
        const uint32_t base = g.slot_base[idx], cnt = g.tiles_touched[idx];
        const float4* r4 = reinterpret_cast<const float4*>(recs + base);
        for (uint32_t k = 0; k < cnt; k++) {
            const float4 r0 = r4[4 * k], r1 = r4[4 * k + 1], r2 = r4[4 * k + 2], r3 = r4[4 * k + 3];
            a[0] += r0.x, a[1] += r0.y, a[2] += r0.z, a[3] += r0.w;
            a[4] += r1.x, a[5] += r1.y, a[6] += r1.z, a[7] += r1.w;
            a[8] += r2.x, a[9] += r2.y, a[10] += r2.z, a[11] += r2.w;
            a[12] += r3.x, a[13] += r3.y, a[14] += r3.z, a[15] += r3.w;
        }
    }
    const float dcolr[3] = {a[0], a[1], a[2]};
    const float g2x = a[3], g2y = a[4];
    const float dcx = a[5], dcy = a[6], dcz = a[7];
    float mean_g[3] = {a[9], a[10], a[11]};
    float rot_g[4] = {a[12], a[13], a[14], a[15]};
    gr.dL_dopacity[idx] = a[8];
    dcol[0] = dcolr[0], dcol[1] = dcolr[1], dcol[2] = dcolr[2];
    dm2[0] = g2x, dm2[1] = g2y, dm2[2] = 0.f;

    float view[16], proj[16];
#pragma unroll
    for (int i = 0; i < 16; i++) view[i] = v.view[i], proj[i] = v.proj[i];
    const float mx = means3D[3 * idx], my = means3D[3 * idx + 1], mz = means3D[3 * idx + 2];
    const float sx = scales[3 * idx], sy = scales[3 * idx + 1], sz = scales[3 * idx + 2];
    const float4 qt = reinterpret_cast<const float4*>(rotations)[idx];
    // The cov2D-inverse -> cov3D -> (scale, quaternion) chain is ill-conditioned for thin surfels (denom^2, b^2 by
    // cancellation): it is evaluated in fp64 here.  Per Gaussian, not per pixel: a few hundred flops, invisible next to
    // the kernel's memory traffic on MI355X, and it removes the dominant fp32 noise of the reference formulation.
    typedef double real;
    const real r = qt.x, x = qt.y, y = qt.z, z = qt.w;
    real Rm[3][3];
    Rm[0][0] = 1. - 2. * (y * y + z * z), Rm[0][1] = 2. * (x * y - r * z), Rm[0][2] = 2. * (x * z + r * y);
    Rm[1][0] = 2. * (x * y + r * z), Rm[1][1] = 1. - 2. * (x * x + z * z), Rm[1][2] = 2. * (y * z - r * x);
    Rm[2][0] = 2. * (x * z - r * y), Rm[2][1] = 2. * (y * z + r * x), Rm[2][2] = 1. - 2. * (x * x + y * y);
    const real s[3] = {(real)v.scale_mod * sx, (real)v.scale_mod * sy, (real)v.scale_mod * sz};
    real Mm[3][3];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 3; i++) Mm[k][i] = s[k] * Rm[i][k];
    real c3[6];
    {
        int o = 0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i; j < 3; j++) c3[o++] = Mm[0][i] * Mm[0][j] + Mm[1][i] * Mm[1][j] + Mm[2][i] * Mm[2][j];
    }
    // ---- K8 computeCov2DCUDA, backward.cu:273-422 ----
    const real tvx0 = (real)view[0] * mx + (real)view[4] * my + (real)view[8] * mz + view[12];
    const real tvy0 = (real)view[1] * mx + (real)view[5] * my + (real)view[9] * mz + view[13];
    const real tvz = (real)view[2] * mx + (real)view[6] * my + (real)view[10] * mz + view[14];
    const real limx = 1.3f * v.tanfovx, limy = 1.3f * v.tanfovy;
    const real txtz = tvx0 / tvz, tytz = tvy0 / tvz;
    const real tx = fmin(limx, fmax(-limx, txtz)) * tvz;
    const real ty = fmin(limy, fmax(-limy, tytz)) * tvz;
    const real x_grad_mul = (txtz < -limx || txtz > limx) ? 0. : 1.;
    const real y_grad_mul = (tytz < -limy || tytz > limy) ? 0. : 1.;
    const real fx = v.focal_x, fy = v.focal_y;
    const real J00 = fx / tvz, J02 = -(fx * tx) / (tvz * tvz);
    const real J11 = fy / tvz, J12 = -(fy * ty) / (tvz * tvz);
    real A0[3], A1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        A0[j] = J00 * view[j * 4 + 0] + J02 * view[j * 4 + 2];
        A1[j] = J11 * view[j * 4 + 1] + J12 * view[j * 4 + 2];
    }
    const real V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    real A0V[3], A1V[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        A0V[j] = A0[0] * V[j][0] + A0[1] * V[j][1] + A0[2] * V[j][2];
        A1V[j] = A1[0] * V[j][0] + A1[1] * V[j][1] + A1[2] * V[j][2];
    }
    const real ca = A0[0] * A0V[0] + A0[1] * A0V[1] + A0[2] * A0V[2] + 0.3f;
    const real cb = A0[0] * A1V[0] + A0[1] * A1V[1] + A0[2] * A1V[2];
    const real cc = A1[0] * A1V[0] + A1[1] * A1V[1] + A1[2] * A1V[2] + 0.3f;
    const real denom = ca * cc - cb * cb;
    real dL_da = 0, dL_db = 0, dL_dc = 0;
    const real denom2inv = 1.0 / ((denom * denom) + 0.0000001f);
    real dcv[6];
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
        dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
        dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
        dcv[0] = A0[0] * A0[0] * dL_da + A0[0] * A1[0] * dL_db + A1[0] * A1[0] * dL_dc;
        dcv[3] = A0[1] * A0[1] * dL_da + A0[1] * A1[1] * dL_db + A1[1] * A1[1] * dL_dc;
        dcv[5] = A0[2] * A0[2] * dL_da + A0[2] * A1[2] * dL_db + A1[2] * A1[2] * dL_dc;
        dcv[1] = 2 * A0[0] * A0[1] * dL_da + (A0[0] * A1[1] + A0[1] * A1[0]) * dL_db + 2 * A1[0] * A1[1] * dL_dc;
        dcv[2] = 2 * A0[0] * A0[2] * dL_da + (A0[0] * A1[2] + A0[2] * A1[0]) * dL_db + 2 * A1[0] * A1[2] * dL_dc;
        dcv[4] = 2 * A0[2] * A0[1] * dL_da + (A0[1] * A1[2] + A0[2] * A1[1]) * dL_db + 2 * A1[1] * A1[2] * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) dcv[i] = 0.;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) dcov[i] = (float)dcv[i];
    real dT0[3], dT1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        dT0[j] = 2 * A0V[j] * dL_da + A1V[j] * dL_db;
        dT1[j] = 2 * A1V[j] * dL_dc + A0V[j] * dL_db;
    }
    const real dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
    const real dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
    const real dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
    const real dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
    const real tzi = 1. / tvz, tz2 = tzi * tzi, tz3 = tz2 * tzi;
    const real dL_dtx = x_grad_mul * -fx * tz2 * dJ02;
    const real dL_dty = y_grad_mul * -fy * tz2 * dJ12;
    const real dL_dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * tx) * tz3 * dJ02 + (2 * fy * ty) * tz3 * dJ12;
    mean_g[0] += (float)(view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz);
    mean_g[1] += (float)(view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz);
    mean_g[2] += (float)(view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz);

    // ---- K9 preprocessCUDA backward, backward.cu:492-548 ----
    const float hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
    const float m_w = 1.0f / (hw + 0.0000001f);
    const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
    const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
    mean_g[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    mean_g[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    mean_g[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

    if (shs != nullptr && dsh != nullptr) {
        // SH backward, backward.cu:152-268
        const float dox = mx - v.campos[0], doy = my - v.campos[1], doz = mz - v.campos[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float dx = dox / len, dy = doy / len, dz = doz / len;
        const float* sh = shs + (size_t)idx * M * 3;
        const uint32_t cl = g.clamped[idx];
        const float dRGB[3] = {dcolr[0] * ((cl & 1u) ? 0.f : 1.f), dcolr[1] * ((cl & 2u) ? 0.f : 1.f), dcolr[2] * ((cl & 4u) ? 0.f : 1.f)};
        float dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
        auto setd = [&](int k, float w) {
            dsh[3 * k] = w * dRGB[0];
            dsh[3 * k + 1] = w * dRGB[1];
            dsh[3 * k + 2] = w * dRGB[2];
        };
        const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
        setd(0, bSH_C0);
        if (D > 0) {
            setd(1, -bSH_C1 * dy);
            setd(2, bSH_C1 * dz);
            setd(3, -bSH_C1 * dx);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dRGBdx[ch] = -bSH_C1 * sh[9 + ch];
                dRGBdy[ch] = -bSH_C1 * sh[3 + ch];
                dRGBdz[ch] = bSH_C1 * sh[6 + ch];
            }
            if (D > 1) {
                setd(4, bSH_C2[0] * xy);
                setd(5, bSH_C2[1] * yz);
                setd(6, bSH_C2[2] * (2.f * zz - xx - yy));
                setd(7, bSH_C2[3] * xz);
                setd(8, bSH_C2[4] * (xx - yy));
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    dRGBdx[ch] += bSH_C2[0] * dy * sh[12 + ch] + bSH_C2[2] * 2.f * -dx * sh[18 + ch] + bSH_C2[3] * dz * sh[21 + ch] +
                                  bSH_C2[4] * 2.f * dx * sh[24 + ch];
                    dRGBdy[ch] += bSH_C2[0] * dx * sh[12 + ch] + bSH_C2[1] * dz * sh[15 + ch] + bSH_C2[2] * 2.f * -dy * sh[18 + ch] +
                                  bSH_C2[4] * 2.f * -dy * sh[24 + ch];
                    dRGBdz[ch] += bSH_C2[1] * dy * sh[15 + ch] + bSH_C2[2] * 2.f * 2.f * dz * sh[18 + ch] + bSH_C2[3] * dx * sh[21 + ch];
                }
                if (D > 2) {
                    setd(9, bSH_C3[0] * dy * (3.f * xx - yy));
                    setd(10, bSH_C3[1] * xy * dz);
                    setd(11, bSH_C3[2] * dy * (4.f * zz - xx - yy));
                    setd(12, bSH_C3[3] * dz * (2.f * zz - 3.f * xx - 3.f * yy));
                    setd(13, bSH_C3[4] * dx * (4.f * zz - xx - yy));
                    setd(14, bSH_C3[5] * dz * (xx - yy));
                    setd(15, bSH_C3[6] * dx * (xx - 3.f * yy));
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        dRGBdx[ch] += (bSH_C3[0] * sh[27 + ch] * 3.f * 2.f * xy + bSH_C3[1] * sh[30 + ch] * yz +
                                       bSH_C3[2] * sh[33 + ch] * -2.f * xy + bSH_C3[3] * sh[36 + ch] * -3.f * 2.f * xz +
                                       bSH_C3[4] * sh[39 + ch] * (-3.f * xx + 4.f * zz - yy) + bSH_C3[5] * sh[42 + ch] * 2.f * xz +
                                       bSH_C3[6] * sh[45 + ch] * 3.f * (xx - yy));
                        dRGBdy[ch] += (bSH_C3[0] * sh[27 + ch] * 3.f * (xx - yy) + bSH_C3[1] * sh[30 + ch] * xz +
                                       bSH_C3[2] * sh[33 + ch] * (-3.f * yy + 4.f * zz - xx) + bSH_C3[3] * sh[36 + ch] * -3.f * 2.f * yz +
                                       bSH_C3[4] * sh[39 + ch] * -2.f * xy + bSH_C3[5] * sh[42 + ch] * -2.f * yz +
                                       bSH_C3[6] * sh[45 + ch] * -3.f * 2.f * xy);
                        dRGBdz[ch] += (bSH_C3[1] * sh[30 + ch] * xy + bSH_C3[2] * sh[33 + ch] * 4.f * 2.f * yz +
                                       bSH_C3[3] * sh[36 + ch] * 3.f * (2.f * zz - xx - yy) + bSH_C3[4] * sh[39 + ch] * 4.f * 2.f * xz +
                                       bSH_C3[5] * sh[42 + ch] * (xx - yy));
                    }
                }
            }
        }
        // coefficients above the active degree keep the reference's zero initialisation (rasterize_points.cu:204)
        const int used = (D + 1) * (D + 1);
        for (int k = used; k < M; k++) dsh[3 * k] = dsh[3 * k + 1] = dsh[3 * k + 2] = 0.f;
        const float ddx = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
        const float ddy = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
        const float ddz = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        mean_g[0] += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
        mean_g[1] += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
        mean_g[2] += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
    }
    // cov3D backward, backward.cu:426-487 (no quaternion-norm Jacobian, B1; ADDS onto the depth-hit rotation grads)
    {
        const real dS[3][3] = {{dcv[0], 0.5 * dcv[1], 0.5 * dcv[2]}, {0.5 * dcv[1], dcv[3], 0.5 * dcv[4]}, {0.5 * dcv[2], 0.5 * dcv[4], dcv[5]}};
        real dM[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                dM[k][j] = 2. * (s[k] * Rm[0][k] * dS[0][j] + s[k] * Rm[1][k] * dS[1][j] + s[k] * Rm[2][k] * dS[2][j]);
#pragma unroll
        for (int k = 0; k < 3; k++) dsc[k] = (float)(Rm[0][k] * dM[k][0] + Rm[1][k] * dM[k][1] + Rm[2][k] * dM[k][2]);
        real Mt[3][3];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 3; j++) Mt[k][j] = dM[k][j] * s[k];
        rot_g[0] += (float)(2 * z * (Mt[0][1] - Mt[1][0]) + 2 * y * (Mt[2][0] - Mt[0][2]) + 2 * x * (Mt[1][2] - Mt[2][1]));
        rot_g[1] += (float)(2 * y * (Mt[1][0] + Mt[0][1]) + 2 * z * (Mt[2][0] + Mt[0][2]) + 2 * r * (Mt[1][2] - Mt[2][1]) - 4 * x * (Mt[2][2] + Mt[1][1]));
        rot_g[2] += (float)(2 * x * (Mt[1][0] + Mt[0][1]) + 2 * r * (Mt[2][0] - Mt[0][2]) + 2 * z * (Mt[1][2] + Mt[2][1]) - 4 * y * (Mt[2][2] + Mt[0][0]));
        rot_g[3] += (float)(2 * r * (Mt[0][1] - Mt[1][0]) + 2 * x * (Mt[2][0] + Mt[0][2]) + 2 * y * (Mt[1][2] + Mt[2][1]) - 4 * z * (Mt[1][1] + Mt[0][0]));
    }
    dm[0] = mean_g[0], dm[1] = mean_g[1], dm[2] = mean_g[2];
    drot[0] = rot_g[0], drot[1] = rot_g[1], drot[2] = rot_g[2], drot[3] = rot_g[3];
}

}  // namespace

int dqo_launch_backward(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx, const float* dL_dcolor,
                        const float* dL_ddepth, const int32_t* hit_image, DqoRastGrads* gr, void* ws, size_t ws_bytes, hipStream_t s) {
    (void)hit_image;  // the hit Gaussian is recovered from its list position kept in the image context
    (void)ws_bytes;
    if (p->P <= 0) return DQO_OK;
    const DqoView v = dqo_make_view(p, in);
    DqoGeomLayout g = dqo_geom_layout(ctx->geom, p->P);
    DqoImageLayout img = dqo_image_layout(ctx->image, p->W, p->H);
    DqoBinLayout bin = dqo_bin_layout(ctx->binning, ctx->inst_capacity);
    const int T = v.gx * v.gy;
    DqoGradRec* recs = (DqoGradRec*)ws;
    DQO_LAUNCH("blend_backward_kernel", blend_backward_kernel, dim3(T), dim3(BWD_THREADS), s, v, g, img, bin, in->scales, in->rotations, dL_dcolor,
                       dL_ddepth, recs, (int64_t)ctx->inst_capacity);
    DQO_LAUNCH("gaussian_backward_kernel", gaussian_backward_kernel, dim3((p->P + 255) / 256), dim3(256), s, v, g, in->means3D, in->scales, in->rotations,
                       in->shs, recs, *gr);
    return DQO_OK;
}
