// Forward blend kernel (K6) for gfx950 — replaces /root/reference/submodules/diff-gaussian-rasterizer-depth/
// cuda_rasterizer/forward.cu:636-866 (renderCUDA_withMask) with its helpers :54-100.
//
// Structure: ONE wave64 per (tile, 8x8 quadrant) — four single-wave workgroups per 16x16 tile, no __syncthreads anywhere.
// The wave streams its tile's depth-sorted list in chunks of 64 positions (one per lane: coalesced index load two chunks ahead, three
// 16-byte record gathers one chunk ahead), tests each entry's footprint against its own quadrant (fwd_hits_quadrant), compacts the
// survivors into wave-private LDS with one ballot — one 64-byte record per entry — and blends them front to back.
// The kernel is bound by VALU issue (SQ counters: the SIMDs issue vector instructions for ~80 % of its cycles), so the entry loop is
// written for its instruction count (round 6: 54 -> 35 VALU instructions per wave step, profiles/r06_ab_fwd_mask_form.txt):
//   * every per-pixel PREDICATE of forward.cu:750-842 (unfinished, depth fixed, valid, blend, finish, new maximum) is a 64-bit lane mask
//     in a scalar register pair — compares write it there, the scalar unit combines masks, selects read them (the helpers below);
//   * an entry's record is read through ONE vector address register with immediate field offsets, its conic arrives pre-scaled;
//   * the opaque hit that fixes a pixel's depth leaves three selects in the loop; the ray / surfel-plane intersection (a double
//     division, two gathers) runs once per pixel behind the walk (finish_hit).
// A wave stops as soon as its own 64 pixels are finished (the reference keeps a whole 256-thread block alive until its
// last pixel is done, and so did the first 4-wave version of this kernel, paying a block barrier per batch).
//
// Per (quadrant, list position) the wave records a live byte: non-zero iff some unfinished pixel of the quadrant saw the entry with
// alpha >= 1/255 (bit r: some pixel of DPP row r = 4x4 block r did) — a superset of "blended it or took it as its depth hit", i.e. of the (pixel, entry) pairs the backward has
// work for; the backward walks live entries only.  The per-Gaussian surfel normal / camera-space point are read from the preprocess tables (forward.cu:779-791 rebuilds
// them from the quaternion for every (pixel, Gaussian) pair).
#include "dqo_common.h"
#include "dqo_cull.h"

namespace {

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

constexpr int FWD_THREADS = 64;

// ---- lane masks as scalar values (round 6) ----
// The entry loop below keeps every per-pixel PREDICATE of forward.cu:750-842 (pixel unfinished, depth fixed, valid, blend, finish, new
// maximum) as a 64-bit lane mask in a scalar register pair: a compare writes its mask straight into the pair (VOP3 v_cmp), masks are
// combined on the scalar unit, and a select reads the pair (v_cndmask).  Written as C++ bools / 0-1 floats the same statements cost 50
// VALU instructions per (wave, entry) step — the compiler materialises ballots as v_cndmask + v_cmp, keeps `finished` / `depth fixed`
// as floats that are multiplied in, and moves the wave-uniform LDS address into a VGPR before every read; in mask form a step is 37.
// Every value that reaches an output is produced by the same IEEE operation on the same operands as before.
typedef unsigned long long lanemask;
#define DQO_CMP_F(name, op, ca, cb)                                                                \
    __device__ __forceinline__ lanemask name(float a, float b) {                                   \
        lanemask m;                                                                                \
        asm(op " %0, %1, %2" : "=s"(m) : ca(a), cb(b));                                            \
        return m;                                                                                  \
    }
// (_sv / _vs: that operand is wave-uniform and read from a scalar register — a loop-invariant threshold then costs no v_mov per trip)
DQO_CMP_F(m_le_sv, "v_cmp_le_f32", "s", "v")    // a <= b (ordered)
DQO_CMP_F(m_ge_vs, "v_cmp_ge_f32", "v", "s")    // a >= b
DQO_CMP_F(m_gt, "v_cmp_gt_f32", "v", "v")       // a >  b
DQO_CMP_F(m_nlt_vs, "v_cmp_nlt_f32", "v", "s")  // !(a < b): true for NaN, like the C++ negation
#undef DQO_CMP_F
__device__ __forceinline__ lanemask m_ge0(float b) {  // 0 >= b
    lanemask m;
    asm("v_cmp_ge_f32 %0, 0, %1" : "=s"(m) : "v"(b));
    return m;
}
__device__ __forceinline__ lanemask m_lt_half(float b) {  // 0.5 < b
    lanemask m;
    asm("v_cmp_lt_f32 %0, 0.5, %1" : "=s"(m) : "v"(b));
    return m;
}
__device__ __forceinline__ lanemask m_ne_i(int a, int b) {
    lanemask m;
    asm("v_cmp_ne_u32 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
// m ? t : f
__device__ __forceinline__ float sel_f(lanemask m, float t, float f) {
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
__device__ __forceinline__ float sel0_f(lanemask m, float t) {  // m ? t : 0
    float r;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(t), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t sel_u(lanemask m, uint32_t t, uint32_t f) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
// LDS reads at a byte address held in a VGPR (+ an immediate offset): one address register serves a whole record
typedef float dqo_lf4 __attribute__((ext_vector_type(4)));
typedef int dqo_li2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 lds_ld4(uint32_t a) {
    const dqo_lf4 t = *reinterpret_cast<const __attribute__((address_space(3))) dqo_lf4*>((uintptr_t)a);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ int2 lds_ld2i(uint32_t a) {
    const dqo_li2 t = *reinterpret_cast<const __attribute__((address_space(3))) dqo_li2*>((uintptr_t)a);
    return make_int2(t.x, t.y);
}
__device__ __forceinline__ void lds_st1i(uint32_t a, int x) {
    *reinterpret_cast<__attribute__((address_space(3))) int*>((uintptr_t)a) = x;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
// The quadrant's own footprint test (dqo_splat_hits_rect + dqo_q_threshold, dqo_cull.h) with v_rcp_f32 / v_log_f32 in place of the two
// IEEE divisions and logf: ~40 instructions shorter per (quadrant, entry).  Still conservative — an entry is dropped only if the bound
// on q over the quadrant exceeds the 1/255 cut-off by dqo_cull.h's margin (0.05 + 1 % in q), against which one ulp in 1 / C or 1e-6 in
// the logarithm is nothing: q is stationary at the minimiser the reciprocal feeds — so every dropped entry is one the walk would have
// found `valid` for no pixel: the outputs, live bytes and n_touched cannot tell the two tests apart.  (The BINNING keeps the IEEE form:
// its two passes must take bit-identical decisions, and its lists are compared against the oracle's.)
__device__ __forceinline__ bool fwd_hits_quadrant(float mx, float my, float A, float B, float C, float opacity, float x0, float y0,
                                                  float x1, float y1) {
#pragma clang fp contract(off)
    const float qthr = 2.0f * (0.693147181f * __builtin_amdgcn_logf(255.0f * fmaxf(opacity, 1e-30f)));  // v_log_f32 = log2; the argument is a normal number
    if (qthr < 0.f) return false;  // opacity < 1/255: alpha < 1/255 even at the centre
    const float dx0 = x0 - mx, dx1 = x1 - mx, dy0 = y0 - my, dy1 = y1 - my;
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;  // centre inside: q_min = 0
    float qmin = 3.0e38f;
    {
        const float invC = __builtin_amdgcn_rcpf(C);
        float dy = fminf(dy1, fmaxf(dy0, -B * dx0 * invC));
        qmin = fminf(qmin, A * dx0 * dx0 + 2.0f * B * dx0 * dy + C * dy * dy);
        dy = fminf(dy1, fmaxf(dy0, -B * dx1 * invC));
        qmin = fminf(qmin, A * dx1 * dx1 + 2.0f * B * dx1 * dy + C * dy * dy);
    }
    {
        const float invA = __builtin_amdgcn_rcpf(A);
        float dx = fminf(dx1, fmaxf(dx0, -B * dy0 * invA));
        qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * dy0 + C * dy0 * dy0);
        dx = fminf(dx1, fmaxf(dx0, -B * dy1 * invA));
        qmin = fminf(qmin, A * dx * dx + 2.0f * B * dx * dy1 + C * dy1 * dy1);
    }
    const float ddx = fmaxf(fabsf(dx0), fabsf(dx1)), ddy = fmaxf(fabsf(dy0), fabsf(dy1));
    const float tmax = fabsf(A) * ddx * ddx + 2.0f * fabsf(B) * ddx * ddy + fabsf(C) * ddy * ddy;
    const float margin = 0.05f + 0.01f * qthr + 4.0e-6f * tmax;
    return !(qmin > qthr + margin);  // NaN-safe: keeps the entry
}

// DqoLossTap, forward half: this wave's share of the masked loss sums from the values it has just written for its 64 pixels, added to
// the frame's 64-bit counters (one set per spread line) with fire-and-forget atomics — nothing waits for them; the backward's blend
// kernel, a kernel boundary later, reads the totals (dqo_tap_totals).  A first version finished the loss here (the wave with the last
// ticket wrote loss_out and grad_scale): 13 k tickets on one address cost 140 us (same-address atomics are served one per ~11 ns), and
// with two-level tickets the returning atomics + their s_waitcnt at the end of every wave still cost what the two loss kernels had.
__device__ __forceinline__ float tap_wave_sum(float x, int lane) {
    return dqo_wave_sum_xor(x, lane);  // fixed order: reproducible (the xor butterfly 32 .. 1, without the LDS crossbar)
}
// owner: the pixel's object id (DqoObjectGate.pixel_object; only read when tap.per_object)
__device__ __forceinline__ void loss_tap_wave(const DqoTapDev& tap, const DqoGeomLayout& g, bool inside, size_t pid, size_t HW, float c0,
                                              float c1, float c2, float depth, int hit_id, int lane, int owner) {
    float e = 0.f, de = 0.f;
    bool m = false, valid = false;
    if (inside) {
        m = tap.mask ? tap.mask[pid] != 0 : true;
        const float g0 = tap.gt_color[pid], g1 = tap.gt_color[HW + pid], g2 = tap.gt_color[2 * HW + pid], gd = tap.gt_depth[pid];
        e = fabsf(c0 - g0) + fabsf(c1 - g1) + fabsf(c2 - g2);
        const float err = depth - gd;
        valid = m && hit_id != -1 && gd > 0.f && err < tap.add_depth_thres;  // mapper.py:850-856
        de = fabsf(err);
    }
    if (tap.per_object) {
        // DqoLossTap.per_object: one set of sums per object id among the wave's mask pixels (usually one or two: a wave-uniform loop)
        m = m && owner >= 0;
        valid = valid && m;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(m);
        while (todo != 0ull) {
            const int f = (int)__builtin_ctzll(todo);
            const int o = __builtin_amdgcn_readlane(owner, f);
            const bool sel = m && owner == o, selv = valid && owner == o;
            const unsigned long long sm = __builtin_amdgcn_ballot_w64(sel), sv = __builtin_amdgcn_ballot_w64(selv);
            const float se = tap_wave_sum(sel ? e : 0.f, lane), sd = tap_wave_sum(selv ? de : 0.f, lane);
            if (lane == 0) {
                unsigned long long* line = g.obj_tap + ((size_t)(blockIdx.x % DQO_OBJ_SPREAD) * DQO_GATE_OBJECTS + (size_t)(o & (DQO_GATE_OBJECTS - 1))) * 4;
                atomicAdd(&line[0], dqo_tap_fixed(se)), atomicAdd(&line[1], (unsigned long long)__popcll(sm));
                if (sv) atomicAdd(&line[2], dqo_tap_fixed(sd)), atomicAdd(&line[3], (unsigned long long)__popcll(sv));
            }
            todo &= ~sm;
        }
        return;
    }
    const unsigned long long nm = __popcll(__builtin_amdgcn_ballot_w64(m)), nv = __popcll(__builtin_amdgcn_ballot_w64(valid));
    if (nm == 0ull) return;  // (wave-uniform; valid implies m)
    const float se = tap_wave_sum(m ? e : 0.f, lane), sd = tap_wave_sum(valid ? de : 0.f, lane);
    if (lane != 0) return;
    unsigned long long* line = reinterpret_cast<unsigned long long*>(g.spread + (size_t)(blockIdx.x % DQO_SPREAD) * 64 + 8);
    // (a non-finite sum — NaN / infinite colours — is not representable in fixed point: dqo_tap_fixed)
    const unsigned long long fe = dqo_tap_fixed(se), fd = dqo_tap_fixed(sd);
    atomicAdd(&line[0], fe), atomicAdd(&line[1], nm);
    if (nv) atomicAdd(&line[2], fd), atomicAdd(&line[3], nv);
}

// GATE: DqoObjectGate — a list entry acts on a pixel only if the Gaussian's object id equals the pixel's owner id.  A template
// parameter, so that the ungated kernel (the reference's semantics, the drop-in op) keeps its instruction stream.  (The gated
// instantiation is held to 72 registers (FWD_MINW): left alone it takes more and loses waves per SIMD, which is where its time goes — the
// instruction counts of the two kernels are equal to 0.1 %; a duplicate of the entry loop without the owner comparison for the
// one-owner quadrants measured nothing.)
//
// SEGS: waves per quadrant.  1 = the wave described at the top of the file.  8 = the list-splitting variant for launches that cannot
// fill the chip (a strong-scaling shard: a few hundred tiles on 1024 SIMDs, where the time of the launch is the time of the ONE wave
// with the longest walk).  The list is walked in rounds of SEGS chunks, chunk r * SEGS + w of round r by wave w:
//   pass 1  every wave multiplies out, per pixel, the product of (1 - alpha) over the valid entries of its chunk and notes whether the
//           chunk holds an opaque hit — both independent of the state the pixel arrives with, as long as it is unfinished;
//   scan    transmittance and "depth already fixed" at the start of chunk w = the round's start state times the products / OR of the
//           chunks before it; a pixel is finished before a chunk iff T_in < T_threshold and a hit lies before it (the reference
//           finishes a pixel at the first valid entry at or behind BOTH the first entry that takes T below the threshold and the
//           first hit, forward.cu:813-817);
//   pass 2  the blend loop itself on the chunk, started from that state (a chunk entered finished ends at once);
// one block barrier per round, and the block stops after the round that leaves no pixel unfinished — the early exit of the serial
// walk at a granularity of SEGS chunks.  At the end wave 0 folds the waves' partial results by list position and writes the pixel.
// The products are grouped by chunk instead of strictly front to back, so T differs from the serial order's in the last bits (and a
// pixel sitting exactly on a threshold may fall on the other side): splitting is opt-in (DqoRastCtx.list_split), not a default.
//
// LDS: one block of FWD_BLK float4 per wave — the compacted entries of its chunk; with SEGS > 1 the same block holds the wave's
// merge record at the end (a wave only ever writes its own block), and SEGS * 64 more float4 behind the blocks the pass-1 results.
// one 64-byte record per compacted entry — conic (A, C pre-multiplied by -0.5) + opacity | xy, depth, object id | rgb, smax | id,
// list position + 1 — and an id table for the chunk's n_touched atomics: 4352 B
constexpr int FWD_BLK = 4 * FWD_THREADS + FWD_THREADS / 4;
constexpr int PART_STRIDE = FWD_BLK * 4;                    // floats per wave block
constexpr int PART_WORDS = 14;                              // merge record of one (run, pixel): 14 x 64 floats <= PART_STRIDE (word 9 unused)
static_assert(PART_WORDS * FWD_THREADS <= PART_STRIDE, "the merge record lives in the wave's own LDS block");

template <bool GATE, int SEGS, bool PF>
__device__ __forceinline__ void blend_quadrant(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                                               const DqoRastOutputs& out, const DqoTapDev& tap, const DqoGateDev& gate, const int tile,
                                               const int quad, const int wave, const int lane, float4* const lds, const int skip_over,
                                               const uint2 range) {
    float4* const s_ent = lds + wave * FWD_BLK;                                  // entry k: s_ent[4 k .. 4 k + 3]
    int* const s_idt = reinterpret_cast<int*>(s_ent + 4 * FWD_THREADS);          // entry k's Gaussian index
    const uint32_t ent_addr = lds_addr_of(s_ent);
    float* const s_part = reinterpret_cast<float*>(lds);  // run j, word k, lane l: s_part[j * PART_STRIDE + k * 64 + l]

    const int tile_x = tile % v.gx, tile_y = tile / v.gx;
    const uint32_t px = tile_x * DQO_TILE + (quad & 1) * 8 + dqo_lane_x(lane);
    const uint32_t py = tile_y * DQO_TILE + (quad >> 1) * 8 + dqo_lane_y(lane);
    const bool inside = px < (uint32_t)v.W && py < (uint32_t)v.H;
    const size_t HW = (size_t)v.W * v.H;
    const size_t pix_id = (size_t)v.W * py + px;
    const int n = (int)(range.y - range.x);  // (range: the tile's list, img.ranges[tile] — handed in by the caller, who has it with the tile id)
    if (n > skip_over) return;  // (SEGS == 1 beside a split launch: the long lists belong to the split blocks)
    // object gate: this pixel's owner, and the owners present in the quadrant as a 64-bit set of (id mod 64) — an entry whose object's
    // bit is not in the set matches no pixel of the quadrant and is culled with the entries that cannot reach it
    // (object ids lie in [0, 64), so the set is exact: in a quadrant with ONE owner — all but the object boundaries — the entries that
    // pass it are exactly the owner's, ownerless pixels simply start finished, and the per-pixel comparison is only made in `mixed`
    // quadrants; the Gaussian's object id arrives in the spare word of its xy record, no gather of its own)
    int owner = -1;
    unsigned long long present = 0ull;
    bool mixed = false;
    if (GATE) {
        if (inside) owner = gate.pobj[pix_id];
        if (owner < 0) owner = (int)0x80000000;  // "no owner": equal to no Gaussian's (non-negative) object id
        // (one trip per distinct owner — usually one or two — on the scalar unit; an OR-butterfly of a 64-bit word over the wave is
        // twelve dependent trips through the LDS crossbar)
        unsigned long long todo = __builtin_amdgcn_ballot_w64(owner >= 0);
        while (todo != 0ull) {
            const int o = __builtin_amdgcn_readlane(owner, (int)__builtin_ctzll(todo));
            present |= 1ull << (o & 63);
            todo &= ~__builtin_amdgcn_ballot_w64(owner == o);
        }
        mixed = __popcll(present) > 1;
    }

    if (n == 0) {
        // masked or empty tile: the reference's torch::full initial values (rasterize_points.cu:79-89).  A tile that is
        // active in the reference but whose instances were all culled as dead is rendered with an empty list instead:
        // colour = bg, ids = -1 (forward.cu:724-725, 852-860).
        const bool rendered = img.tile_flag[tile] != 0u;
        if (SEGS > 1 && wave != 0) return;  // (block-uniform branch: no barrier is skipped)
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
        if (lane == 0) img.walk4[tile * 4 + quad] = 0;
        if (tap.scale != nullptr)
            loss_tap_wave(tap, g, inside, pix_id, HW, rendered ? v.bg[0] : 0.f, rendered ? v.bg[1] : 0.f, rendered ? v.bg[2] : 0.f, 0.f,
                          rendered ? -1 : 0, lane, owner);
        return;
    }

    const float qx0 = (float)(tile_x * DQO_TILE + (quad & 1) * 8), qy0 = (float)(tile_y * DQO_TILE + (quad >> 1) * 8);
    const float pixfx = (float)px, pixfy = (float)py;
    const float static_gate = (inside && (!GATE || owner >= 0)) ? 1.f : 0.f;   // 0 = this pixel starts finished (outside the image; no owner)
    lanemask alive_m = __builtin_amdgcn_ballot_w64(static_gate != 0.f);        // pixels that are not finished
    lanemask fixed_m = 0ull;                                                   // pixels whose depth has been fixed by an opaque hit
    const float hit_thr = fmaxf(v.opaque_thr, 1.0f / 255.0f);
    const float thr255 = 1.0f / 255.0f, T_thr = v.T_thr;
    float T = 1.0f, end_T = 1.0f;
    uint32_t last_contributor = 0, hit_pos = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    float depth_ = 0.f;
    int hit_id = -1, hit_color_id = -1;
    float color_weight_max = -1.f, hit_depth_weight = 0.f;
    uint8_t* live = bin.live_q + (size_t)quad * (size_t)bin.list_cap + range.x;  // this quadrant's live bytes of this tile's segment

    const int chunks_all = (n + FWD_THREADS - 1) / FWD_THREADS;
    // the chunk in flight (written by the chunk loops below, read by entries())
    int cnt = 0;          // compacted entries of the chunk
    // per-entry result, word 14 of the entry's record: how many of the quadrant's pixels saw it with T' > 0.5 (n_touched,
    // forward.cu:833-835, quirk B8) | its row code << 8 (0: the entry was not live for this quadrant)
    uint32_t wmax_pos = 0;  // SEGS > 1: list position of the entry that holds color_weight_max
    auto entries = [&]() {
        const uint32_t ea0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ent_addr);
        // PF: the next entry's records are read from LDS while this one is blended (a read one past the chunk's last entry stays
        // inside the wave's own block and is never used)
        float4 xy_pf = make_float4(0.f, 0.f, 0.f, 0.f), co_pf = xy_pf;
        if (PF) co_pf = lds_ld4(ea0), xy_pf = lds_ld4(ea0 + 16);
        for (int k = 0; k < cnt && alive_m != 0ull; k++) {
            // the record address: formed on the scalar unit, moved into ONE vector register; the loads carry the field offsets
            uint32_t ea = ea0 + ((uint32_t)k << 6);
            asm("" : "+v"(ea));
            const float4 co_cur = PF ? co_pf : lds_ld4(ea), xy_cur = PF ? xy_pf : lds_ld4(ea + 16);
            if (PF) co_pf = lds_ld4(ea + 64), xy_pf = lds_ld4(ea + 80);
            // ---- per-pixel update (forward.cu:750-842) ----
            const float dx = xy_cur.x - pixfx, dy = xy_cur.y - pixfy;
            const float power = dqo_power_pre(co_cur.x, co_cur.y, co_cur.z, dx, dy);
            const float alpha = fminf(0.99f, co_cur.w * dqo_gauss(power));
            // forward.cu:763-772 (and the pixel is not finished): power <= 0, alpha >= 1/255
            lanemask valid_m = m_ge0(power) & m_le_sv(thr255, alpha) & alive_m;
            // (object gate, mixed quadrants only: the entry acts on the pixels of its own object)
            if (GATE && mixed) valid_m &= ~m_ne_i(__float_as_int(xy_cur.w), owner);
            if (valid_m != 0ull) {
                const float4 cs = lds_ld4(ea + 32);
                const int2 ip = lds_ld2i(ea + 48);
                const int gid = ip.x;
                const uint32_t contributor = (uint32_t)ip.y;
                const float a_v = sel0_f(valid_m, alpha);  // an entry that is not valid for a pixel acts on it with alpha = 0
                const lanemask hit_m = m_ge_vs(a_v, hit_thr) & ~fixed_m;  // valid, no depth yet, alpha >= opaque_threshold
                if (hit_m != 0ull) {
                    // forward.cu:792-810: the first Gaussian with alpha >= opaque_threshold fixes this pixel's depth.  Only what the step
                    // alone knows is kept here (three selects); the ray / surfel-plane intersection with its double-precision division
                    // and its two gathers runs ONCE per pixel behind the walk (finish_hit) — inside the loop it ran once per distinct
                    // hit entry of the quadrant, each time behind a vmcnt(0) wait that also drained the next chunk's prefetch
                    hit_id = (int)sel_u(hit_m, (uint32_t)gid, (uint32_t)hit_id);
                    hit_depth_weight = sel_f(hit_m, alpha * T, hit_depth_weight);
                    hit_pos = sel_u(hit_m, contributor, hit_pos);
                    fixed_m |= hit_m;
                }
                const float test_T = T * (1.f - a_v);  // == T when the entry is not valid for this pixel
                const lanemask nlt_m = m_nlt_vs(test_T, T_thr);
                const lanemask blend_m = valid_m & nlt_m;             // forward.cu:818-840
                const lanemask finish_m = valid_m & ~nlt_m & fixed_m;  // forward.cu:813-817: done, T NOT updated
                const float w = sel0_f(blend_m, a_v * T);
                C0 += cs.x * w;
                C1 += cs.y * w;
                C2 += cs.z * w;
                const lanemask newmax_m = blend_m & m_gt(w, color_weight_max);
                color_weight_max = sel_f(newmax_m, w, color_weight_max);
                hit_color_id = (int)sel_u(newmax_m, (uint32_t)gid, (uint32_t)hit_color_id);
                if (SEGS > 1) wmax_pos = sel_u(newmax_m, contributor, wmax_pos);  // (the waves' maxima are merged by list position)
                last_contributor = sel_u(blend_m, contributor, last_contributor);
                end_T = sel_f(blend_m, test_T, end_T);
                T = sel_f(finish_m, T, test_T);  // keeps decaying below T_thr until an opaque hit appears (forward.cu:841)
                alive_m &= ~finish_m;
                // live for the backward: some pixel of the quadrant saw the entry with alpha >= 1/255 while unfinished — a
                // superset of "blended it or took it as its depth hit" (equal except when every such pixel is saturated
                // below T_threshold), so the backward never misses a pair it has work for.  The per-entry results are
                // wave-uniform: n_touched count | WHICH 16-lane rows (4x4 pixel blocks, dqo_lane_x / _y) saw the entry (the backward's
                // rows walk their own sub-lists), one word of the entry's record.
                const lanemask half_m = blend_m & m_lt_half(test_T);
                lds_st1i(ea + 56, (int)__popcll(half_m) | (int)(dqo_row_code(valid_m) << 8));
            }
        }
    };

    // the chunk's cull test and compaction into LDS; returns the lane's slot among the survivors (reach: this lane's entry survived)
    auto compact = [&](int pos, int id, const float4& co, const float4& xy, const float4& cs_me, bool& reach) {
        // which entries of this chunk can reach this quadrant at all (conservative, dqo_cull.h); compact them into LDS
        reach = pos < n && fwd_hits_quadrant(xy.x, xy.y, co.x, co.y, co.z, co.w, qx0, qy0, qx0 + 7.f, qy0 + 7.f);
        if (GATE) reach = reach && ((present >> (__float_as_int(xy.w) & 63)) & 1ull) != 0ull;
        const unsigned long long rm = __builtin_amdgcn_ballot_w64(reach);
        cnt = (int)__popcll(rm);
        const int myk = (int)__popcll(rm & ((1ull << lane) - 1ull));
        if (reach) {
            float4* const e = s_ent + 4 * myk;
            e[0] = make_float4(-0.5f * co.x, co.y, -0.5f * co.z, co.w);  // (fwd_power_pre)
            e[1] = xy;
            e[2] = cs_me;
            // (the reference's running counter `contributor` = list position + 1; the result word starts at "not live")
            *reinterpret_cast<int4*>(e + 3) = make_int4(id, pos + 1, 0, 0);
            s_idt[myk] = id;
        }
        return myk;
    };
    // what the chunk leaves behind: one scattered integer atomic per touched Gaussian, and the live byte of every list position
    // (coalesced 64-byte store)
    auto chunk_results = [&](int pos, bool reach, int myk) {
        const int* const res = reinterpret_cast<const int*>(s_ent) + 14;  // entry k: res[16 k]
        const int half_k = lane < cnt ? (res[16 * lane] & 0xff) : 0;
        if (half_k > 0) atomicAdd(&out.n_touched[s_idt[lane]], half_k);
        // live byte = the entry's row code (0: no pixel of the quadrant has work for it)
        if (pos < n) live[pos] = reach ? (uint8_t)((res[16 * myk] >> 8) & 0xf) : (uint8_t)0;
    };

    // prologue: loads of the wave's first chunk (chunk `wave`; SEGS == 1: chunk 0).  The list is read TWO chunks ahead of the walk: a
    // chunk's records are gathered by Gaussian index, so its indices must have arrived before the gathers can be issued — with one
    // chunk of lookahead the wave sat on a vmcnt(0) at the head of every chunk (index load -> wait -> three gathers); now the head of
    // chunk c issues the gathers of chunk c + 1 from indices loaded a whole chunk ago, and the index load of chunk c + 2.
    constexpr int STRIDE = SEGS * FWD_THREADS;  // list positions between two chunks of this wave
    int id_nx = 0, id_n2 = 0;
    float4 co_nx = make_float4(0.f, 0.f, 0.f, 0.f), xy_nx = co_nx, cs_nx = co_nx;
    const int first = (SEGS > 1 ? wave : 0) * FWD_THREADS + lane;
    if (first < n) id_nx = (int)bin.point_list[range.x + first];
    if (first + STRIDE < n) id_n2 = (int)bin.point_list[range.x + first + STRIDE];
    if (first < n) {
        co_nx = g.conic_opacity[id_nx];
        xy_nx = g.xy_depth[id_nx];
        cs_nx = g.rgb_smax[id_nx];
    }
    // the next chunk's gathers (indices: id_n2, here since the previous chunk's head) and the index load of the chunk behind it
    auto fetch_ahead = [&](const int pos) {
        const int pn = pos + STRIDE;
        if (pn < n) {
            id_nx = id_n2;
            co_nx = g.conic_opacity[id_nx];
            xy_nx = g.xy_depth[id_nx];
            cs_nx = g.rgb_smax[id_nx];
        }
        if (pn + STRIDE < n) id_n2 = (int)bin.point_list[range.x + pn + STRIDE];
    };
    if (SEGS == 1) {
        for (int c = 0; c < chunks_all && alive_m != 0ull; c++) {  // a finished quadrant never looks at the entries further back
            const int pos = c * FWD_THREADS + lane;
            const int id = id_nx;
            const float4 co = co_nx, xy = xy_nx, cs_me = cs_nx;
            // issue the next chunk's loads now; they complete while this chunk is blended (all three records: a gather left for the
            // compaction below would sit, unhidden, between the cull test and the first entry of every chunk)
            fetch_ahead(pos);
            bool reach;
            const int myk = compact(pos, id, co, xy, cs_me, reach);
            if (cnt > 0) entries();
            chunk_results(pos, reach, myk);
        }
    } else {
        // ---- rounds of SEGS chunks, chunk r * SEGS + w to wave w ----
        float* const s_ph = reinterpret_cast<float*>(lds + SEGS * FWD_BLK);  // [round parity][wave][P | H][lane]
        float Tg = 1.f;     // transmittance at the start of the round (every wave computes the same bits)
        bool hitg = false;  // the depth was fixed before the round
        float T_last = 1.f; // running T behind the last chunk of this wave that the pixel entered unfinished
        int last_alive = -1;
        for (int r = 0; r * SEGS < chunks_all; r++) {
            const int c = r * SEGS + wave;
            const int pos = c * FWD_THREADS + lane;
            const int id = id_nx;
            const float4 co = co_nx, xy = xy_nx, cs_me = cs_nx;
            fetch_ahead(pos);  // the wave's chunk of the next round
            bool reach;
            const int myk = compact(pos, id, co, xy, cs_me, reach);  // (a chunk past the end of the list: cnt = 0)
            // pass 1: the chunk's product of (1 - alpha) and "holds an opaque hit", for a pixel that enters it unfinished
            float P = 1.f;
            bool H = false;
#pragma unroll 2
            for (int k = 0; k < cnt; k++) {
                const float4 xy_cur = s_ent[4 * k + 1], co_cur = s_ent[4 * k];
                const float dx = xy_cur.x - pixfx, dy = xy_cur.y - pixfy;
                const float power = dqo_power_pre(co_cur.x, co_cur.y, co_cur.z, dx, dy);
                const float alpha = fminf(0.99f, co_cur.w * dqo_gauss(power));
                const bool other = GATE && mixed && __float_as_int(xy_cur.w) != owner;
                const float a_g = (power <= 0.0f && !other) ? alpha * static_gate : 0.f;
                const float a_v = a_g >= 1.0f / 255.0f ? alpha : 0.f;
                P *= 1.f - a_v;
                H = H || a_v >= hit_thr;
            }
            float* const ph = s_ph + (r & 1) * SEGS * 2 * FWD_THREADS;
            ph[(wave * 2 + 0) * FWD_THREADS + lane] = P;
            ph[(wave * 2 + 1) * FWD_THREADS + lane] = H ? 1.f : 0.f;
            // (one barrier per round: the round after next writes this parity again, and the next round's barrier lies between)
            __syncthreads();
            // scan: the state this chunk starts from, and the state the next round starts from
            float T_in = Tg, T_n = Tg;
            bool h_in = hitg, h_n = hitg;
#pragma unroll
            for (int j = 0; j < SEGS; j++) {
                if (j == wave) T_in = T_n, h_in = h_n;
                T_n *= ph[(j * 2 + 0) * FWD_THREADS + lane];
                h_n = h_n || ph[(j * 2 + 1) * FWD_THREADS + lane] != 0.f;
            }
            // pass 2: the blend of the chunk from that state.  A pixel is finished before the chunk iff T_in < T_threshold and a hit
            // lies before it (forward.cu:813-817: it finishes at the first valid entry at or behind both)
            T = T_in;
            fixed_m = __builtin_amdgcn_ballot_w64(h_in);
            const bool alive_in = static_gate != 0.f && !(h_in && T_in < v.T_thr);
            alive_m = __builtin_amdgcn_ballot_w64(alive_in);
            if (cnt > 0) entries();
            chunk_results(pos, reach, myk);
            if (alive_in && c < chunks_all) T_last = T, last_alive = c;
            Tg = T_n, hitg = h_n;
            // every wave sees the same state: the block leaves together
            if (__builtin_amdgcn_ballot_w64(static_gate != 0.f && !(hitg && Tg < v.T_thr)) == 0ull) break;
        }
        // ---- merge: the waves' results folded by wave 0 ----
        float* rec = s_part + wave * PART_STRIDE + lane;
        rec[0 * FWD_THREADS] = C0, rec[1 * FWD_THREADS] = C1, rec[2 * FWD_THREADS] = C2;
        rec[3 * FWD_THREADS] = color_weight_max, rec[4 * FWD_THREADS] = __int_as_float(hit_color_id);
        rec[5 * FWD_THREADS] = __uint_as_float(last_contributor), rec[6 * FWD_THREADS] = end_T;
        rec[7 * FWD_THREADS] = __int_as_float(hit_id), rec[8 * FWD_THREADS] = __uint_as_float(hit_pos);
        rec[10 * FWD_THREADS] = hit_depth_weight;
        rec[11 * FWD_THREADS] = T_last, rec[12 * FWD_THREADS] = __int_as_float(last_alive);
        rec[13 * FWD_THREADS] = __uint_as_float(wmax_pos);
        __syncthreads();
        if (wave != 0) return;
        C0 = C1 = C2 = 0.f;
        color_weight_max = -1.f, hit_color_id = -1, wmax_pos = 0;
        last_contributor = 0, end_T = 1.f;
        hit_id = -1, hit_pos = 0, hit_depth_weight = 0.f;
        T = 1.f, last_alive = -1;
        for (int j = 0; j < SEGS; j++) {
            const float* q = s_part + j * PART_STRIDE + lane;
            C0 += q[0 * FWD_THREADS], C1 += q[1 * FWD_THREADS], C2 += q[2 * FWD_THREADS];
            const float wm = q[3 * FWD_THREADS];
            const uint32_t wp = __float_as_uint(q[13 * FWD_THREADS]);
            if (wm > color_weight_max || (wm == color_weight_max && wm >= 0.f && wp < wmax_pos))  // the first maximum in list order
                color_weight_max = wm, hit_color_id = __float_as_int(q[4 * FWD_THREADS]), wmax_pos = wp;
            const uint32_t lc = __float_as_uint(q[5 * FWD_THREADS]);
            if (lc > last_contributor) last_contributor = lc, end_T = q[6 * FWD_THREADS];
            const uint32_t hp = __float_as_uint(q[8 * FWD_THREADS]);
            if (hp != 0u && (hit_pos == 0u || (hp & 0x7fffffffu) < (hit_pos & 0x7fffffffu))) {
                // (one wave holds the hit: the chunks behind it start with the depth fixed)
                hit_pos = hp, hit_id = __float_as_int(q[7 * FWD_THREADS]);
                hit_depth_weight = q[10 * FWD_THREADS];
            }
            const int la = __float_as_int(q[12 * FWD_THREADS]);
            if (la > last_alive) last_alive = la, T = q[11 * FWD_THREADS];  // the running transmittance behind the last chunk entered
        }
    }
    // finish_hit (forward.cu:784-810, once per pixel): the hit entry's surfel plane against this pixel's ray
    if (__builtin_amdgcn_ballot_w64(hit_id != -1) != 0ull) {
        const int hid = max(hit_id, 0);  // (pixels without a hit read Gaussian 0 and discard it: one round of loads for the wave)
        const float4 n_np = g.normal_c[hid];
        const float raw_smax = g.point_c[hid].w;
        const float hit_zc = g.xy_depth[hid].z, hit_smax = g.rgb_smax[hid].w;  // the hit entry's view depth and (modified) largest scale
        if (hit_id != -1) {
            const float3 ray = pixel_ray(px, py, v.focal_x, v.focal_y, v.cx, v.cy);  // (here, not in the prologue: three registers less across the walk)
            const HitEval h = eval_hit(ray, n_np);
            const float angle_distance = fabsf(h.den);
            const float depth_distance = fabsf(h.hit_z - hit_zc);
            depth_ = (depth_distance <= hit_smax * v.depth_thr && angle_distance >= v.normal_thr) ? h.hit_z : hit_zc;
            // the backward repeats the test with the raw scales (backward.cu:1009-1016): decide it here once
            const bool plane_b = depth_distance <= v.depth_thr * raw_smax && angle_distance >= v.normal_thr;
            hit_pos |= plane_b ? 0x80000000u : 0u;
        }
    }
    const float oc0 = C0 + T * v.bg[0], oc1 = C1 + T * v.bg[1], oc2 = C2 + T * v.bg[2];  // running T, not end_T (quirk B2, forward.cu:852)
    if (inside) {
        img.final_T[pix_id] = end_T;
        img.n_contrib[pix_id] = last_contributor;
        img.hit_pos[pix_id] = hit_pos;
        out.out_color[pix_id] = oc0;
        out.out_color[HW + pix_id] = oc1;
        out.out_color[2 * HW + pix_id] = oc2;
        out.out_depth[pix_id] = depth_;
        out.out_hit_depth[pix_id] = hit_id;
        out.out_hit_color[pix_id] = hit_color_id;
        out.out_hit_color_weight[pix_id] = fmaxf(color_weight_max, 0.f);  // the running maximum IS the recorded weight (0 if none)
        out.out_hit_depth_weight[pix_id] = hit_depth_weight;
        out.out_T[pix_id] = end_T;
    }
    // list positions the backward has to walk for this quadrant
    int w = inside ? (int)max(last_contributor, hit_pos & 0x7fffffffu) : 0;
    w = (int)dqo_wave_max_u32((uint32_t)w, lane);  // (w >= 0)
    if (lane == 0) img.walk4[tile * 4 + quad] = (uint32_t)w;
    if (tap.scale != nullptr) loss_tap_wave(tap, g, inside, pix_id, HW, oc0, oc1, oc2, depth_, hit_id, lane, owner);
}

// One wave64 per (tile, quadrant), block b: XCD group x = b % 8 (blocks b and b + 8 share an XCD), within the group item
// j = b / 8 = (tile slot, quadrant).  (Both instantiations are held to FWD_MINW waves per SIMD, see below.)
#ifndef FWD_WPB
#define FWD_WPB 1  // waves (= quadrants of ONE tile) per workgroup; independent of each other either way
#endif
#ifndef FWD_PF
#define FWD_PF false  // the next entry's records read one step ahead (a register rotation)
#endif
#ifndef FWD_MINW
#define FWD_MINW 7  // waves per SIMD the register allocation leaves room for (72 VGPRs).  The kernel's time follows its occupancy —
                    // 5 / 6 / 7 waves: 121 / 112 / 109 us before, 106 / 103 us (6 / 7) after the hit entry's depth and scale left the
                    // loop's registers; at 8 (64 registers) the gated instantiation spills 15 and loses what it gains — the ungated
                    // one (the drop-in operator's) fits 64 without a spill and runs at 8; maps beyond 768 Ki Gaussians run the gated
                    // kernel at 6 (dqo_launch_blend_forward)
#endif
template <bool GATE, int MINW>
__global__ __launch_bounds__(FWD_THREADS * FWD_WPB, MINW) void blend_forward_kernel(const DqoView v, DqoGeomLayout g, DqoImageLayout img,
                                                                                            DqoBinLayout bin, DqoRastOutputs out,
                                                                                            const DqoTapDev tap, const DqoGateDev gate,
                                                                                            const int64_t header_capacity) {
    // header_capacity >= 0: the launch has ONE extra block, which forms the frame's header from the statistics lines (frames without a
    // long-list sort launch, whose first block does that otherwise: dqo_skip_long_sort); nothing in this kernel reads the header
    if (header_capacity >= 0 && blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x < 64) dqo_header_from_spread(g, header_capacity, bin.bucket, (int)threadIdx.x);
        return;
    }
    __shared__ float4 lds[FWD_BLK * FWD_WPB];
    const int wave = FWD_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int xg = blockIdx.x & 7, jg = FWD_WPB > 1 ? ((int)(blockIdx.x >> 3) * FWD_WPB + wave) : (int)(blockIdx.x >> 3);
    const int T8 = (v.gx * v.gy + 7) / 8;
    const uint4 si = img.slot_info[xg * T8 + (jg >> 2)];  // (tile, list start, list end) of the slot: one round
    const uint32_t tile_u = si.x;
    if (tile_u == 0xffffffffu) return;  // unused slot
    blend_quadrant<GATE, 1, FWD_PF>(v, g, img, bin, out, tap, gate, (int)tile_u, jg & 3, wave, (int)(threadIdx.x & 63), lds, 0x7fffffff,
                                    make_uint2(si.y, si.z));
}

// DqoRastCtx.list_split: blocks of SPLIT_RUNS waves.  The first SPLIT_GRID blocks take the long lists (longer than list_split entries: the
// queue tile_sort_wave_kernel left in img.split_tiles, longest first) one (tile, quadrant) at a time, every wave one run of the list,
// each block drawing its next item with a ticket; they are dispatched first, so the longest critical paths start first.  Every other
// block is eight independent waves of the serial kind: two tiles of one XCD band, four quadrants each, skipping the long lists.
#ifndef SPLIT_PF
#define SPLIT_PF true
#endif
constexpr int SPLIT_RUNS = 8;
// (two blocks per CU at the gated kernel's 100 registers; held to 80 for three, it spills and measures the same)
constexpr int SPLIT_GRID = 256;   // long-list blocks in front of the short-list blocks, and as many again behind them
template <bool GATE>
__global__ __launch_bounds__(FWD_THREADS * SPLIT_RUNS, 4) void blend_forward_split_kernel(const DqoView v, DqoGeomLayout g, DqoImageLayout img,
                                                                                           DqoBinLayout bin, DqoRastOutputs out,
                                                                                           const DqoTapDev tap, const DqoGateDev gate,
                                                                                           const int list_split) {
    __shared__ float4 lds[SPLIT_RUNS * (FWD_BLK + FWD_THREADS)];  // the waves' blocks, then the rounds' pass-1 results (two parities)
    __shared__ uint32_t s_item;
    // (readfirstlane: the compiler must know that everything derived from the wave index is wave-uniform, or every loop over a run's
    // bounds is built as a divergent one)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int T = v.gx * v.gy, T8 = (T + 7) / 8;
    const int short_blocks = 8 * ((T8 + 1) / 2);
    // one long-list block per CU goes first and works beside the short-list blocks; the ones behind them take the slots those leave
    if (blockIdx.x < SPLIT_GRID || (int)blockIdx.x >= SPLIT_GRID + short_blocks) {
        const uint32_t items = 4u * min(g.counters[4], (uint32_t)T);
        for (;;) {  // (ends for every wave of every block: the ticket only grows and `items` is fixed)
            if (threadIdx.x == 0) s_item = atomicAdd(&g.counters[3], 1u);
            __syncthreads();
            const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);
            if (it >= items) return;
            const int tile_s = (int)img.split_tiles[it >> 2];
            blend_quadrant<GATE, SPLIT_RUNS, SPLIT_PF>(v, g, img, bin, out, tap, gate, tile_s, (int)(it & 3u), wave, lane, lds, 0x7fffffff,
                                                       img.ranges[tile_s]);
            __syncthreads();  // wave 0 has read the other waves' merge records (and everyone this trip's ticket)
        }
    }
    const int b = (int)blockIdx.x - SPLIT_GRID;
    const int slot = (b >> 3) * 2 + (wave >> 2);  // two tile slots of band b % 8 per block
    if (slot >= T8) return;
    const uint32_t tile_u = img.tile_order[(b & 7) * T8 + slot];
    if (tile_u == 0xffffffffu) return;
    blend_quadrant<GATE, 1, SPLIT_PF>(v, g, img, bin, out, tap, gate, (int)tile_u, wave & 3, wave, lane, lds, list_split, img.ranges[tile_u]);
}

}  // namespace

int dqo_launch_blend_forward(const DqoView& v, const DqoGeomLayout& g, const DqoImageLayout& img, const DqoBinLayout& bin,
                             const DqoRastOutputs& out, int T, const DqoTapDev& tap, const DqoGateDev& gate, int list_split, hipStream_t s,
                             int64_t header_capacity) {
    const bool gt = gate.gobj != nullptr;
    if (list_split > 0) {
        const dim3 grid(2 * SPLIT_GRID + 8 * (((T + 7) / 8 + 1) / 2)), block(FWD_THREADS * SPLIT_RUNS);
        if (gt) DQO_LAUNCH("blend_forward_kernel", blend_forward_split_kernel<true>, grid, block, s, v, g, img, bin, out, tap, gate, list_split);
        else DQO_LAUNCH("blend_forward_kernel", blend_forward_split_kernel<false>, grid, block, s, v, g, img, bin, out, tap, gate, list_split);
        return DQO_OK;
    }
    const dim3 grid(8 * ((T + 7) / 8) * 4 / FWD_WPB + (header_capacity >= 0 ? 1 : 0)), block(FWD_THREADS * FWD_WPB);
    // waves per SIMD the register allocation leaves room for: what the walk gains from a seventh wave on a 500 k map (-3 us) it loses on
    // a 1 M one (+5 us: more waves gathering from a bigger set of records) — same-box A/B, profiles/r06_ab_fwd_seven_waves.txt
    const bool small_map = v.P <= 786432;
    if (gt && small_map) DQO_LAUNCH("blend_forward_kernel", (blend_forward_kernel<true, FWD_MINW>), grid, block, s, v, g, img, bin, out, tap, gate, header_capacity);
    else if (gt) DQO_LAUNCH("blend_forward_kernel", (blend_forward_kernel<true, 6>), grid, block, s, v, g, img, bin, out, tap, gate, header_capacity);
    else if (small_map) DQO_LAUNCH("blend_forward_kernel", (blend_forward_kernel<false, 8>), grid, block, s, v, g, img, bin, out, tap, gate, header_capacity);
    else DQO_LAUNCH("blend_forward_kernel", (blend_forward_kernel<false, 6>), grid, block, s, v, g, img, bin, out, tap, gate, header_capacity);
    return DQO_OK;
}
