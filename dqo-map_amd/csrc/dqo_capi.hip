// extern "C" entry points of libdqoraster.so (declared in include/dqo_raster.h): argument validation, buffer-size
// queries and launches.  No allocation, no synchronisation (except dqo_rast_read_header), no exceptions.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "dqo_common.h"

static thread_local char g_err[512] = "";

void dqo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int dqo_launch_mark_visible(int P, const float* means3D, const float* view, const float* proj, uint8_t* present, hipStream_t s);
int dqo_launch_knn3(int P, const float* xyz, float* mean_d2, int32_t* idx3, void* ws, size_t ws_bytes, hipStream_t s);
size_t dqo_knn3_ws_bytes(int P);
size_t dqo_knn3_query_ws_bytes(int Q, int R);
int dqo_launch_knn3_query(int Q, const float* q_xyz, int R, const float* r_xyz, float* dist2, int32_t* idx3, void* ws, size_t ws_bytes,
                          hipStream_t s, float bound2, const int32_t* q_group = nullptr, const int32_t* r_group = nullptr,
                          const float* group_box = nullptr);
int dqo_launch_quadric_iou(int B, const float* axes, const float* R, const float* center, const float* P34, const float* obs,
                           float* bbox, float* loss, int32_t* valid, float* g_axes, float* g_R, float* g_center, hipStream_t s);
int dqo_launch_quadric_adam(int n_obj, int n_iters, const int32_t* view_offset, const float* P34_views, const float* obs_views,
                            const int32_t* view_schedule, float* axes, float* R, float* center, float* loss_hist, hipStream_t s);

int dqo_launch_map_activate(int P, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* opacity,
                            float* scales, float* rotations, hipStream_t s);
size_t dqo_map_loss_ws_bytes(void);
int dqo_launch_map_loss(int W, int H, const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                        const float* gt_depth, const uint8_t* render_mask, float color_weight, float depth_weight, float add_depth_thres,
                        float* loss_out, float* dL_dcolor, float* dL_ddepth, void* ws, hipStream_t s);
size_t dqo_map_ssim_ws_bytes(int W, int H);
int dqo_launch_map_ssim(int W, int H, const float* img, const float* gt, float weight, float* ssim_out, float* dL_dimg, int accumulate,
                        float* loss8, void* ws, hipStream_t s);
int dqo_launch_map_adam(const DqoAdamStep* st, hipStream_t s);
size_t dqo_map_attach_ws_bytes(int P);
int dqo_launch_adam_multi(const DqoAdamTensor* ts, int n_tensors, int step, double beta1, double beta2, double eps, hipStream_t s,
                          int32_t* step_dev = nullptr, int bump = 0);
int dqo_launch_history_merge(int P, int M, float max_weight, int first_row, const uint8_t* row_flags, const float* conf0, const float* conf,
                             const float* xyz0, const float* shs0, const float* scaling0, const float* rot0_unit, float* xyz, float* shs,
                             float* scaling_raw, float* rotation_raw, hipStream_t s);
int dqo_launch_map_attach(int P, const float* scaling, const float* xyz, const float* rotation, const float* scaling0, const float* xyz0,
                          const float* rotation0, const uint8_t* mask, int attach_count, float* loss, float* g_scaling, float* g_xyz,
                          float* g_rotation, void* ws, hipStream_t s);
int dqo_launch_backward_adam(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx, const float* dL_dcolor,
                             const float* dL_ddepth, const DqoAdamStep* st, void* ws, hipStream_t s);
size_t dqo_icp_ws_bytes(void);
int dqo_launch_icp(int H, int W, const float* vertex0, const float* vertex1, const float* normal0, const float* normal1, const float* pose10,
                   float fx, float fy, float cx, float cy, float dist_thr, float normal_thr, float* JtJ, float* JtR, int32_t* valid_count,
                   void* ws, hipStream_t s);
int dqo_launch_tile_count(int W, int H, int mode, const uint8_t* mask_in, const float* T_map, uint8_t* mask_out, int32_t* tile_count,
                          int32_t* total, hipStream_t s);
int dqo_launch_tile_color_error(int W, int H, const float* render, const float* gt, float* err_px, float* tile_sum, hipStream_t s);
int dqo_launch_growth_scales(int n, const float* xyz, const int32_t* obj, const float* radius, const int32_t* i_new, const float* d2_old,
                             const int32_t* i_old, const float* extra_radius, float reach2, float min_radius, float max_radius, float* scales,
                             uint8_t* invalid, hipStream_t s);
int dqo_launch_growth_inside(int n, const float* d2, const int32_t* idx, const float* radius, uint8_t* inside, hipStream_t s);
int dqo_launch_error_maps(int64_t HW, const float* gt_color, const float* gt_depth, const float* render, const float* depth,
                          const int32_t* depth_index, const uint8_t* mask, float* color_err, float* depth_err, hipStream_t s);
int dqo_launch_attach_pixels(int n, const float* xyz, const float* V, float fx, float fy, float cx, float cy, int W, int H,
                             const int32_t* pixel_object, int32_t* lin, int32_t* sparse, unsigned long long* tile_objects, hipStream_t s);
int dqo_launch_attach_decide(int n, const float* xyz, const float* opacity, const int32_t* obj, const int32_t* lin, const int32_t* hit_index,
                             const float* hit_weight, const float* sxyz, const float* scaling_raw, const float* rotation_raw,
                             const int32_t* gobj, float plane_thr, float opacity_low, uint8_t* out, hipStream_t s);
int dqo_launch_tap_report(const DqoGeomLayout& g, const DqoTapDev& tap, hipStream_t s);
int dqo_launch_accumulate_confidence(int H, int W, int P, const int32_t* index, const float* confidence, float* gmax, float* gmin,
                                     float* gmean, int32_t* counter, hipStream_t s);
int dqo_launch_accumulate_error(int H, int W, int P, const float* color_err, const float* depth_err, const float* normal_err,
                                const int32_t* color_index, const int32_t* depth_index, float color_thr, float depth_thr,
                                float normal_thr, int check_max, float* gs_color, float* gs_depth, float* gs_normal, float* rescale,
                                int32_t* counters, hipStream_t s);

// ---- per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) -------------------------------
#include <map>
#include <mutex>
#include <string>
#include <vector>
int g_dqo_profile_on = 0;
namespace {
std::mutex g_profile_mu;  // the timing tables are process-wide: launches may come from several host threads (one per stream)
struct Pending {
    std::string name;
    hipEvent_t start, stop;
};
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_event_pool;
std::map<std::string, std::pair<double, uint32_t>> g_profile_acc;
hipEvent_t take_event() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
// before / after bracket ONE launch of the calling thread: the index of its pending entry is kept per thread, so brackets of
// different threads may interleave
static thread_local size_t t_pending_idx = 0;
void dqo_profile_before(const char* name, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_profile_mu);
    Pending p{name, take_event(), take_event()};
    (void)hipEventRecord(p.start, s);
    t_pending_idx = g_pending.size();
    g_pending.push_back(p);
}
void dqo_profile_after(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_profile_mu);
    if (t_pending_idx < g_pending.size()) (void)hipEventRecord(g_pending[t_pending_idx].stop, s);
}

extern "C" {

#define DQO_API __attribute__((visibility("default")))

DQO_API int dqo_profile_enable(int on) {
    g_dqo_profile_on = on ? 1 : 0;
    return DQO_OK;
}

// Waits for every bracketed launch recorded so far, accumulates elapsed times per kernel name and copies up to
// max_entries accumulated rows out (returns the number of rows).  reset != 0 clears the accumulators afterwards.
DQO_API int dqo_profile_collect(DqoProfileEntry* out, int max_entries, int reset) {
    std::lock_guard<std::mutex> lk(g_profile_mu);
    for (auto& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            auto& a = g_profile_acc[p.name];
            a.first += ms;
            a.second += 1;
        }
        g_event_pool.push_back(p.start);
        g_event_pool.push_back(p.stop);
    }
    g_pending.clear();
    int n = 0;
    for (auto& kv : g_profile_acc) {
        if (n >= max_entries || out == nullptr) break;
        strncpy(out[n].name, kv.first.c_str(), sizeof(out[n].name) - 1);
        out[n].name[sizeof(out[n].name) - 1] = 0;
        out[n].total_ms = kv.second.first;
        out[n].calls = kv.second.second;
        n++;
    }
    if (reset) g_profile_acc.clear();
    return n;
}

DQO_API int dqo_abi_version(void) { return DQO_ABI_VERSION; }
DQO_API size_t dqo_abi_sizeof(int32_t which) {
    static const size_t sz[] = {sizeof(DqoRastParams), sizeof(DqoRastInputs), sizeof(DqoRastOutputs), sizeof(DqoRastCtx), sizeof(DqoRastGrads),
                                sizeof(DqoRastHeader), sizeof(DqoProfileEntry), sizeof(DqoAdamStep), sizeof(DqoLossTap), sizeof(DqoObjectGate),
                                sizeof(DqoAdamTensor)};
    return (which >= 0 && which < (int32_t)(sizeof(sz) / sizeof(sz[0]))) ? sz[which] : 0;
}
DQO_API const char* dqo_last_error(void) { return g_err; }

DQO_API size_t dqo_rast_geom_bytes(int32_t P, int32_t W, int32_t H) {
    (void)W;
    (void)H;
    return dqo_geom_layout(nullptr, P < 0 ? 0 : P).total;
}
DQO_API size_t dqo_rast_image_bytes(int32_t W, int32_t H) { return dqo_image_layout(nullptr, W, H).total; }
DQO_API size_t dqo_rast_binning_bytes(int64_t cap) { return dqo_bin_layout(nullptr, cap < 0 ? 0 : cap, cap < 0 ? 0 : cap, 0).total; }
DQO_API size_t dqo_rast_binning_bytes_bucketed(int64_t cap, int32_t W, int32_t H, int32_t bucket) {
    if (cap < 0) cap = 0;
    if (bucket <= 0 || W <= 0 || H <= 0) return dqo_rast_binning_bytes(cap);
    return dqo_bin_layout(nullptr, cap, dqo_list_cap(cap, W, H, bucket), bucket).total;
}
DQO_API size_t dqo_rast_backward_workspace_bytes(int64_t cap) { return dqo_bwd_ws_bytes(cap); }

static int check_common(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx) {
    DQO_CHECK_ARG(p && in && ctx, "null params / inputs / ctx");
    DQO_CHECK_ARG(p->P >= 0 && p->W > 0 && p->H > 0, "bad sizes P=%d W=%d H=%d", p->P, p->W, p->H);
    DQO_CHECK_ARG(p->D >= 0 && p->D <= 3, "sh degree %d out of range 0..3", p->D);
    DQO_CHECK_ARG((p->W + DQO_TILE - 1) / DQO_TILE < 65536 && (p->H + DQO_TILE - 1) / DQO_TILE < 65536, "image too large");
    DQO_CHECK_ARG(in->viewmatrix && in->projmatrix && in->campos && in->bg, "viewmatrix/projmatrix/campos/bg must be given");
    if (p->P > 0) {
        DQO_CHECK_ARG(in->means3D && in->opacities, "means3D / opacities missing");
        DQO_CHECK_ARG((in->shs != nullptr) != (in->colors_precomp != nullptr),
                      "Please provide excatly one of either SHs or precomputed colors!");
        DQO_CHECK_ARG(in->cov3D_precomp == nullptr,
                      "cov3D_precomp is not supported by the depth rasteriser: its blend kernel reads scales/rotations (forward.cu:780)");
        DQO_CHECK_ARG(in->scales && in->rotations, "scales and rotations are required");
        if (in->shs) DQO_CHECK_ARG(p->M >= (p->D + 1) * (p->D + 1), "sh has %d coefficients, degree %d needs %d", p->M, p->D, (p->D + 1) * (p->D + 1));
        if (in->colors_precomp == nullptr && p->M == 0) {
            // rasterizer_impl.cu:267-270 analogue
            dqo_set_error("For non-RGB, provide precomputed Gaussian colors!");
            return DQO_ERR_INVALID_ARG;
        }
    }
    DQO_CHECK_ARG(ctx->geom && ctx->geom_bytes >= dqo_rast_geom_bytes(p->P, p->W, p->H), "geom buffer too small (%zu < %zu)",
                  ctx->geom_bytes, dqo_rast_geom_bytes(p->P, p->W, p->H));
    DQO_CHECK_ARG(ctx->image && ctx->image_bytes >= dqo_rast_image_bytes(p->W, p->H), "image buffer too small (%zu < %zu)",
                  ctx->image_bytes, dqo_rast_image_bytes(p->W, p->H));
    return DQO_OK;
}

static int check_outputs(const DqoRastParams* p, const DqoRastOutputs* o) {
    DQO_CHECK_ARG(o, "null outputs");
    DQO_CHECK_ARG(o->out_color && o->out_depth && o->out_hit_color && o->out_hit_depth && o->out_hit_color_weight &&
                      o->out_hit_depth_weight && o->out_T,
                  "null image output");
    if (p->P > 0) DQO_CHECK_ARG(o->n_touched && o->radii, "null n_touched / radii");
    return DQO_OK;
}

DQO_API int dqo_rast_forward_prepare(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, void* stream) {
    int rc = check_common(p, in, ctx);
    if (rc) return rc;
    rc = check_outputs(p, out);
    if (rc) return rc;
    rc = dqo_launch_forward_prepare(p, in, out, ctx, (hipStream_t)stream);
    if (rc) return rc;
    return dqo_launch_mark_header_stage0(p, ctx, (hipStream_t)stream);  // (a no-op unless the frame's first stage is fused into its second)
}

DQO_API int dqo_rast_read_header(const DqoRastCtx* ctx, DqoRastHeader* host_out, void* stream) {
    DQO_CHECK_ARG(ctx && ctx->geom && host_out, "null ctx / out");
    static thread_local uint32_t buf[128 + 64 * DQO_SPREAD];  // header | counters | spread statistics counters
    DQO_CHECK_HIP(hipMemcpyAsync(buf, ctx->geom, sizeof(buf), hipMemcpyDeviceToHost, (hipStream_t)stream));
    DQO_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    memcpy(host_out, buf, sizeof(DqoRastHeader));
    // Stage 2 has written the frame's header (tile_scan_kernel / the sort kernel's header_from_spread): report it as it is.  The
    // counters and statistics lines it was formed from are NOT a second source: bucket mode allocates slots per region (counters[0]
    // stays 0) and dqo_rast_backward_adam's tail clears them for the next frame.
    if (host_out->stage == 2u) return host_out->overflow ? DQO_ERR_OVERFLOW : DQO_OK;
    // after `prepare` only the statistics lines are valid: visible Gaussians and the candidate count (>= N) a caller sizes its
    // binning buffer from; everything else is not known yet
    uint32_t nv = 0, nc = 0;
    for (int j = 0; j < DQO_SPREAD; j++) nv += buf[128 + 64 * j], nc += buf[128 + 64 * j + 1];
    memset(host_out, 0, sizeof(*host_out));
    host_out->stage = 1u;
    host_out->num_visible = nv;
    host_out->num_candidates = nc;
    return host_out->overflow ? DQO_ERR_OVERFLOW : DQO_OK;
}

static int check_render(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx) {
    int rc = check_common(p, in, ctx);
    if (rc) return rc;
    rc = check_outputs(p, out);
    if (rc) return rc;
    DQO_CHECK_ARG(ctx->inst_capacity >= 0 && ctx->inst_capacity < (int64_t)0xffffffffll, "bad inst_capacity");
    DQO_CHECK_ARG(ctx->inst_capacity == 0 || ctx->binning, "null binning buffer");
    DQO_CHECK_ARG(ctx->tile_bucket_capacity >= 0, "negative tile_bucket_capacity");
    DQO_CHECK_ARG(ctx->loss_tap == nullptr || (ctx->loss_tap->gt_color && ctx->loss_tap->gt_depth && ctx->loss_tap->loss_out &&
                                               ctx->loss_tap->grad_scale), "loss tap with a null pointer");
    DQO_CHECK_ARG(ctx->object_gate == nullptr || (ctx->object_gate->gaussian_object && ctx->object_gate->pixel_object),
                  "object gate with a null pointer");
    DQO_CHECK_ARG(ctx->loss_tap == nullptr || !ctx->loss_tap->per_object || ctx->object_gate, "a per-object loss tap needs the object gate");
    if (ctx->binning_bytes < dqo_rast_binning_bytes_bucketed(ctx->inst_capacity, p->W, p->H, ctx->tile_bucket_capacity)) {
        dqo_set_error("binning buffer too small (%zu < %zu)", ctx->binning_bytes,
                      dqo_rast_binning_bytes_bucketed(ctx->inst_capacity, p->W, p->H, ctx->tile_bucket_capacity));
        return DQO_ERR_WORKSPACE;
    }
    return DQO_OK;
}

DQO_API int dqo_rast_forward_render(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, void* stream) {
    const int rc = check_render(p, in, out, ctx);
    if (rc) return rc;
    return dqo_launch_forward_render(p, in, out, ctx, (hipStream_t)stream);
}

DQO_API int dqo_rast_forward(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx, void* stream) {
    int rc = check_render(p, in, out, ctx);  // (covers the checks of the first stage)
    if (rc) return rc;
    rc = dqo_launch_forward_prepare(p, in, out, ctx, (hipStream_t)stream);
    if (rc) return rc;
    return dqo_launch_forward_render(p, in, out, ctx, (hipStream_t)stream);
}

DQO_API int dqo_rast_forward_async(const DqoRastParams* p, const DqoRastInputs* in, DqoRastOutputs* out, DqoRastCtx* ctx,
                                   DqoRastHeader* header_host, void* header_event, void* stream) {
    int rc = check_render(p, in, out, ctx);  // (covers the checks of the first stage)
    if (rc) return rc;
    rc = dqo_launch_forward_prepare(p, in, out, ctx, (hipStream_t)stream);
    if (rc) return rc;
    return dqo_launch_forward_render(p, in, out, ctx, (hipStream_t)stream, header_host, (hipEvent_t)header_event);
}

DQO_API int dqo_rast_backward(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx, const float* dL_dcolor,
                              const float* dL_ddepth, const int32_t* hit_image, DqoRastGrads* g, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(p, in, ctx);
    if (rc) return rc;
    DQO_CHECK_ARG(g, "null grads");
    if (p->P == 0) {  // rasterize_points.cu:208 (with a loss tap the loss of the background-only frame is still reported)
        if (ctx->loss_tap == nullptr) return DQO_OK;
        return dqo_launch_tap_report(dqo_geom_layout(ctx->geom, 0), dqo_tap_dev(ctx->loss_tap), (hipStream_t)stream);
    }
    DQO_CHECK_ARG((dL_dcolor && dL_ddepth) || ctx->loss_tap, "null upstream gradients");
    DQO_CHECK_ARG(ctx->loss_tap == nullptr || (ctx->loss_tap->gt_color && ctx->loss_tap->gt_depth && ctx->loss_tap->out_color &&
                                               ctx->loss_tap->out_depth && ctx->loss_tap->loss_out && ctx->loss_tap->grad_scale),
                  "loss tap with a null pointer");
    DQO_CHECK_ARG(g->dL_dmeans3D && g->dL_dopacity && g->dL_dscales && g->dL_drotations, "null gradient output");
    DQO_CHECK_ARG(p->M == 0 || g->dL_dsh, "null dL_dsh");
    DQO_CHECK_ARG(ctx->binning || ctx->inst_capacity == 0, "null binning buffer");
    if (ws_bytes < dqo_rast_backward_workspace_bytes(ctx->inst_capacity) || (ws == nullptr && ctx->inst_capacity > 0)) {
        dqo_set_error("backward workspace too small (%zu < %zu)", ws_bytes, dqo_rast_backward_workspace_bytes(ctx->inst_capacity));
        return DQO_ERR_WORKSPACE;
    }
    return dqo_launch_backward(p, in, ctx, dL_dcolor, dL_ddepth, hit_image, g, ws, ws_bytes, (hipStream_t)stream);
}

static int check_adam_step(const DqoAdamStep* st, bool need_grads) {
    DQO_CHECK_ARG(st, "null step");
    DQO_CHECK_ARG(st->P >= 0 && st->M >= 1 && (st->step >= 1 || st->step_dev != nullptr), "bad P / M / step");
    if (st->P == 0) return DQO_OK;
    DQO_CHECK_ARG(st->xyz && st->shs && st->opacity_raw && st->scaling_raw && st->rotation_raw, "null parameter");
    if (need_grads) DQO_CHECK_ARG(st->g_means3D && st->g_sh && st->g_opacity && st->g_scales && st->g_rotations, "null gradient");
    DQO_CHECK_ARG(st->m_xyz && st->m_shs && st->m_opacity && st->m_scaling && st->m_rotation && st->v_xyz && st->v_shs &&
                      st->v_opacity && st->v_scaling && st->v_rotation,
                  "null optimiser state");
    return DQO_OK;
}

DQO_API int dqo_rast_backward_adam(const DqoRastParams* p, const DqoRastInputs* in, const DqoRastCtx* ctx, const float* dL_dcolor,
                                   const float* dL_ddepth, const DqoAdamStep* st, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_common(p, in, ctx);
    if (rc) return rc;
    rc = check_adam_step(st, false);
    if (rc) return rc;
    DQO_CHECK_ARG(st->P == p->P && st->M == p->M, "step is for P=%d M=%d, the rasteriser call for P=%d M=%d", st->P, st->M, p->P, p->M);
    if (p->P == 0) {
        if (ctx->loss_tap == nullptr) return DQO_OK;
        return dqo_launch_tap_report(dqo_geom_layout(ctx->geom, 0), dqo_tap_dev(ctx->loss_tap), (hipStream_t)stream);
    }
    DQO_CHECK_ARG(in->shs != nullptr, "the fused backward + Adam step needs SH colours (precomputed colours have no parameter group)");
    DQO_CHECK_ARG(p->M <= 16, "the fused backward + Adam step holds gradient rows of at most 16 SH coefficients (M = %d)", p->M);
    DQO_CHECK_ARG((dL_dcolor && dL_ddepth) || ctx->loss_tap, "null upstream gradients");
    DQO_CHECK_ARG(ctx->loss_tap == nullptr || (ctx->loss_tap->gt_color && ctx->loss_tap->gt_depth && ctx->loss_tap->out_color &&
                                               ctx->loss_tap->out_depth && ctx->loss_tap->loss_out && ctx->loss_tap->grad_scale),
                  "loss tap with a null pointer");
    DQO_CHECK_ARG(ctx->binning || ctx->inst_capacity == 0, "null binning buffer");
    if (ws_bytes < dqo_rast_backward_workspace_bytes(ctx->inst_capacity) || (ws == nullptr && ctx->inst_capacity > 0)) {
        dqo_set_error("backward workspace too small (%zu < %zu)", ws_bytes, dqo_rast_backward_workspace_bytes(ctx->inst_capacity));
        return DQO_ERR_WORKSPACE;
    }
    return dqo_launch_backward_adam(p, in, ctx, dL_dcolor, dL_ddepth, st, ws, (hipStream_t)stream);
}

DQO_API int dqo_mark_visible(int32_t P, const float* means3D, const float* view, const float* proj, uint8_t* present, void* stream) {
    DQO_CHECK_ARG(P >= 0, "bad P");
    if (P == 0) return DQO_OK;
    DQO_CHECK_ARG(means3D && view && proj && present, "null pointer");
    return dqo_launch_mark_visible(P, means3D, view, proj, present, (hipStream_t)stream);
}

DQO_API size_t dqo_knn3_query_workspace_bytes(int32_t Q, int32_t R) { return dqo_knn3_query_ws_bytes(Q < 1 ? 1 : Q, R < 1 ? 1 : R); }

static int knn3_query(int32_t Q, const float* q_xyz, int32_t R, const float* r_xyz, float max_dist, float* dist2, int32_t* idx3, void* ws,
                      size_t ws_bytes, void* stream, const int32_t* q_group = nullptr, const int32_t* r_group = nullptr,
                      const float* group_box = nullptr) {
    DQO_CHECK_ARG(Q >= 0 && R >= 0, "negative size");
    DQO_CHECK_ARG(max_dist > 0.f, "max_dist must be positive");
    if (Q == 0) return DQO_OK;
    DQO_CHECK_ARG(q_xyz && dist2 && idx3, "null query / output");
    DQO_CHECK_ARG(R > 0 && r_xyz, "empty reference set");
    if (ws == nullptr || ws_bytes < dqo_knn3_query_ws_bytes(Q, R)) {
        dqo_set_error("knn query workspace too small (%zu < %zu)", ws_bytes, dqo_knn3_query_ws_bytes(Q, R));
        return DQO_ERR_WORKSPACE;
    }
    const float bound2 = max_dist < 1.8e19f ? max_dist * max_dist : 3.402823466e+38f;
    return dqo_launch_knn3_query(Q, q_xyz, R, r_xyz, dist2, idx3, ws, ws_bytes, (hipStream_t)stream, bound2, q_group, r_group, group_box);
}

DQO_API int dqo_knn3_query(int32_t Q, const float* q_xyz, int32_t R, const float* r_xyz, float* dist2, int32_t* idx3, void* ws, size_t ws_bytes,
                           void* stream) {
    return knn3_query(Q, q_xyz, R, r_xyz, 3.402823466e+38f, dist2, idx3, ws, ws_bytes, stream);
}

DQO_API int dqo_knn3_query_within(int32_t Q, const float* q_xyz, int32_t R, const float* r_xyz, float max_dist, float* dist2, int32_t* idx3,
                                  void* ws, size_t ws_bytes, void* stream) {
    return knn3_query(Q, q_xyz, R, r_xyz, max_dist, dist2, idx3, ws, ws_bytes, stream);
}

DQO_API int dqo_knn3_query_grouped(int32_t Q, const float* q_xyz, const int32_t* q_group, int32_t R, const float* r_xyz, const int32_t* r_group,
                                   const float* group_box, float max_dist, float* dist2, int32_t* idx3, void* ws, size_t ws_bytes, void* stream) {
    DQO_CHECK_ARG(Q == 0 || (q_group && r_group), "null group ids");
    DQO_CHECK_ARG(R < (1 << 25), "the grouped search carries the group id in the index word: at most 2^25 - 1 reference points");
    return knn3_query(Q, q_xyz, R, r_xyz, max_dist, dist2, idx3, ws, ws_bytes, stream, q_group, r_group, group_box);
}

DQO_API size_t dqo_knn3_workspace_bytes(int32_t P) { return dqo_knn3_ws_bytes(P < 0 ? 0 : P); }

DQO_API int dqo_knn3(int32_t P, const float* xyz, float* mean_d2, int32_t* idx3, void* ws, size_t ws_bytes, void* stream) {
    DQO_CHECK_ARG(P >= 0, "bad P");
    if (P == 0) return DQO_OK;
    DQO_CHECK_ARG(xyz && mean_d2 && idx3, "null pointer");
    if (ws == nullptr || ws_bytes < dqo_knn3_ws_bytes(P)) {
        dqo_set_error("knn workspace too small (%zu < %zu)", ws_bytes, dqo_knn3_ws_bytes(P));
        return DQO_ERR_WORKSPACE;
    }
    return dqo_launch_knn3(P, xyz, mean_d2, idx3, ws, ws_bytes, (hipStream_t)stream);
}

DQO_API int dqo_quadric_iou_fwd_bwd(int32_t B, const float* axes, const float* R, const float* center, const float* P34, const float* obs,
                                    float* bbox, float* loss, int32_t* valid, float* g_axes, float* g_R, float* g_center, void* stream) {
    DQO_CHECK_ARG(B >= 0, "bad B");
    if (B == 0) return DQO_OK;
    DQO_CHECK_ARG(axes && R && center && P34 && obs && bbox && loss && valid && g_axes && g_R && g_center, "null pointer");
    return dqo_launch_quadric_iou(B, axes, R, center, P34, obs, bbox, loss, valid, g_axes, g_R, g_center, (hipStream_t)stream);
}

DQO_API int dqo_map_activate(int32_t P, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* opacity,
                             float* scales, float* rotations, void* stream) {
    DQO_CHECK_ARG(P >= 0, "bad P");
    if (P == 0) return DQO_OK;
    DQO_CHECK_ARG(opacity_raw && scaling_raw && rotation_raw && opacity && scales && rotations, "null pointer");
    return dqo_launch_map_activate(P, opacity_raw, scaling_raw, rotation_raw, opacity, scales, rotations, (hipStream_t)stream);
}

DQO_API size_t dqo_map_loss_workspace_bytes(void) { return dqo_map_loss_ws_bytes(); }

DQO_API int dqo_map_loss_fwd_bwd(int32_t W, int32_t H, const float* color, const float* depth, const int32_t* depth_index,
                                 const float* gt_color, const float* gt_depth, const uint8_t* render_mask, float color_weight,
                                 float depth_weight, float add_depth_thres, float* loss_out, float* dL_dcolor, float* dL_ddepth, void* ws,
                                 size_t ws_bytes, void* stream) {
    DQO_CHECK_ARG(W > 0 && H > 0, "bad image size");
    DQO_CHECK_ARG(color && depth && depth_index && gt_color && gt_depth && loss_out && dL_dcolor && dL_ddepth, "null pointer");
    if (ws == nullptr || ws_bytes < dqo_map_loss_ws_bytes()) {
        dqo_set_error("loss workspace too small (%zu < %zu)", ws_bytes, dqo_map_loss_ws_bytes());
        return DQO_ERR_WORKSPACE;
    }
    return dqo_launch_map_loss(W, H, color, depth, depth_index, gt_color, gt_depth, render_mask, color_weight, depth_weight,
                               add_depth_thres, loss_out, dL_dcolor, dL_ddepth, ws, (hipStream_t)stream);
}

DQO_API size_t dqo_map_ssim_workspace_bytes(int32_t W, int32_t H) { return (W > 0 && H > 0) ? dqo_map_ssim_ws_bytes(W, H) : 0; }

DQO_API int dqo_map_ssim_fwd_bwd(int32_t W, int32_t H, const float* image, const float* gt_image, float weight, float* ssim_out,
                                 float* dL_dimage, int32_t accumulate, float* loss_out8, void* ws, size_t ws_bytes, void* stream) {
    DQO_CHECK_ARG(W > 0 && H > 0, "bad image size");
    DQO_CHECK_ARG(image && gt_image && ssim_out, "null pointer");
    if (ws == nullptr || ws_bytes < dqo_map_ssim_ws_bytes(W, H)) {
        dqo_set_error("ssim workspace too small (%zu < %zu)", ws_bytes, dqo_map_ssim_ws_bytes(W, H));
        return DQO_ERR_WORKSPACE;
    }
    return dqo_launch_map_ssim(W, H, image, gt_image, weight, ssim_out, dL_dimage, accumulate != 0, loss_out8, ws, (hipStream_t)stream);
}

DQO_API size_t dqo_map_attach_workspace_bytes(int32_t P) { return dqo_map_attach_ws_bytes(P < 0 ? 0 : P); }

DQO_API int dqo_map_history_merge(int32_t P, int32_t M, float max_weight, int32_t first_row, const uint8_t* row_flags, const float* conf0,
                                  const float* conf, const float* xyz0, const float* shs0, const float* scaling0, const float* rot0_unit,
                                  float* xyz, float* shs, float* scaling_raw, float* rotation_raw, void* stream) {
    DQO_CHECK_ARG(P >= 0 && M >= 1 && M * 3 < 256, "bad P / M");
    if (P == 0) return DQO_OK;
    DQO_CHECK_ARG(first_row >= 0 && first_row < P, "first_row %d outside [0, %d)", first_row, P);
    DQO_CHECK_ARG(conf0 && conf && xyz0 && shs0 && scaling0 && rot0_unit && xyz && shs && scaling_raw && rotation_raw, "null tensor");
    return dqo_launch_history_merge(P, M, max_weight, first_row, row_flags, conf0, conf, xyz0, shs0, scaling0, rot0_unit, xyz, shs,
                                    scaling_raw, rotation_raw, (hipStream_t)stream);
}

DQO_API int dqo_map_attach_loss_fwd_bwd(int32_t P, const float* scaling_raw, const float* xyz, const float* rotation_raw,
                                        const float* init_scaling_raw, const float* init_xyz, const float* init_rotation_raw,
                                        const uint8_t* attach_mask, int32_t attach_count, float* loss, float* g_scaling_raw, float* g_xyz,
                                        float* g_rotation_raw, void* ws, size_t ws_bytes, void* stream) {
    DQO_CHECK_ARG(P >= 0 && attach_count >= 0, "bad P / attach_count");
    DQO_CHECK_ARG(loss, "null loss");
    if (P == 0) return dqo_launch_zero_words(reinterpret_cast<uint32_t*>(loss), 1, (hipStream_t)stream);
    DQO_CHECK_ARG(scaling_raw && xyz && rotation_raw && init_scaling_raw && init_xyz && init_rotation_raw && attach_mask && g_scaling_raw &&
                      g_xyz && g_rotation_raw, "null pointer");
    if (ws == nullptr || ws_bytes < dqo_map_attach_ws_bytes(P)) {
        dqo_set_error("attach-loss workspace too small (%zu < %zu)", ws_bytes, dqo_map_attach_ws_bytes(P));
        return DQO_ERR_WORKSPACE;
    }
    return dqo_launch_map_attach(P, scaling_raw, xyz, rotation_raw, init_scaling_raw, init_xyz, init_rotation_raw, attach_mask, attach_count,
                                 loss, g_scaling_raw, g_xyz, g_rotation_raw, ws, (hipStream_t)stream);
}

static int check_adam_multi(const DqoAdamTensor* ts, int32_t n_tensors, double beta1, double beta2, double eps) {
    DQO_CHECK_ARG(n_tensors >= 0 && n_tensors <= DQO_ADAM_MULTI_MAX, "n_tensors out of range");
    DQO_CHECK_ARG(n_tensors == 0 || ts != nullptr, "null tensor list");
    DQO_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "bad betas / eps");
    int64_t total = 0;
    for (int t = 0; t < n_tensors; t++) {
        DQO_CHECK_ARG(ts[t].n >= 0, "negative element count");
        DQO_CHECK_ARG(ts[t].n == 0 || (ts[t].p && ts[t].g && ts[t].m && ts[t].v), "null pointer in the tensor list");
        total += ts[t].n;
    }
    DQO_CHECK_ARG(total < ((int64_t)1 << 40), "too many elements");
    return DQO_OK;
}

DQO_API int dqo_adam_multi(const DqoAdamTensor* ts, int32_t n_tensors, int32_t step, double beta1, double beta2, double eps, void* stream) {
    DQO_CHECK_ARG(step >= 1, "step is the 1-based count of this update");
    const int rc = check_adam_multi(ts, n_tensors, beta1, beta2, eps);
    if (rc) return rc;
    return dqo_launch_adam_multi(ts, n_tensors, step, beta1, beta2, eps, (hipStream_t)stream);
}

DQO_API int dqo_adam_multi_dev(const DqoAdamTensor* ts, int32_t n_tensors, int32_t* step_dev, int32_t advance, double beta1, double beta2,
                               double eps, void* stream) {
    DQO_CHECK_ARG(step_dev != nullptr, "null step counter");
    const int rc = check_adam_multi(ts, n_tensors, beta1, beta2, eps);
    if (rc) return rc;
    return dqo_launch_adam_multi(ts, n_tensors, 1, beta1, beta2, eps, (hipStream_t)stream, step_dev, advance != 0);
}

DQO_API int dqo_accumulate_gaussian_error(int32_t H, int32_t W, int32_t P, const float* ce, const float* de, const float* ne,
                                          const int32_t* ci, const int32_t* di, float color_thr, float depth_thr, float normal_thr,
                                          int32_t check_max, float* gs_color, float* gs_depth, float* gs_normal, float* rescale,
                                          int32_t* counters, void* stream) {
    DQO_CHECK_ARG(H > 0 && W > 0 && P >= 0, "bad sizes");
    if (P == 0) return DQO_OK;
    DQO_CHECK_ARG(ce && de && ne && ci && di && gs_color && gs_depth && gs_normal && rescale, "null pointer");
    DQO_CHECK_ARG(check_max || counters, "mean mode needs the counters scratch buffer");
    return dqo_launch_accumulate_error(H, W, P, ce, de, ne, ci, di, color_thr, depth_thr, normal_thr, check_max, gs_color, gs_depth,
                                       gs_normal, rescale, counters, (hipStream_t)stream);
}

DQO_API int dqo_accumulate_gaussian_confidence(int32_t H, int32_t W, int32_t P, const int32_t* gaussian_index_map,
                                               const float* gaussian_confidence_map, float* gs_max, float* gs_min, float* gs_mean,
                                               int32_t* counter, void* stream) {
    DQO_CHECK_ARG(H > 0 && W > 0 && P >= 0, "bad sizes");
    if (P == 0) return DQO_OK;
    DQO_CHECK_ARG(gaussian_index_map && gaussian_confidence_map && gs_max && gs_min && gs_mean && counter, "null pointer");
    return dqo_launch_accumulate_confidence(H, W, P, gaussian_index_map, gaussian_confidence_map, gs_max, gs_min, gs_mean, counter,
                                            (hipStream_t)stream);
}

DQO_API int dqo_tile_count_mask(int32_t W, int32_t H, const uint8_t* pixel_mask, int32_t* tile_count, void* stream) {
    DQO_CHECK_ARG(W > 0 && H > 0 && pixel_mask && tile_count, "bad size / null pointer");
    return dqo_launch_tile_count(W, H, 0, pixel_mask, nullptr, nullptr, tile_count, nullptr, (hipStream_t)stream);
}

DQO_API int dqo_transmission_mask(int32_t W, int32_t H, const float* T_map, uint8_t* render_mask, int32_t* tile_count, int32_t* total,
                                  void* stream) {
    DQO_CHECK_ARG(W > 0 && H > 0 && T_map && tile_count, "bad size / null pointer");
    return dqo_launch_tile_count(W, H, 1, nullptr, T_map, render_mask, tile_count, total, (hipStream_t)stream);
}

DQO_API int dqo_tile_color_error(int32_t W, int32_t H, const float* render, const float* gt, float* color_error, float* tile_sum,
                                 void* stream) {
    DQO_CHECK_ARG(W > 0 && H > 0 && render && gt && tile_sum, "bad size / null pointer");
    return dqo_launch_tile_color_error(W, H, render, gt, color_error, tile_sum, (hipStream_t)stream);
}

DQO_API int dqo_attach_pixels(int32_t n, const float* temp_xyz, const float* viewmatrix, float fx, float fy, float cx, float cy, int32_t W,
                              int32_t H, const int32_t* pixel_object, int32_t* lin, int32_t* sparse_pixel_object, uint64_t* tile_objects,
                              void* stream) {
    DQO_CHECK_ARG(n >= 0 && W > 0 && H > 0 && (int64_t)W * H < (1ll << 31), "bad size");
    DQO_CHECK_ARG(viewmatrix && pixel_object && sparse_pixel_object && tile_objects && (n == 0 || (temp_xyz && lin)), "null pointer");
    return dqo_launch_attach_pixels(n, temp_xyz, viewmatrix, fx, fy, cx, cy, W, H, pixel_object, lin, sparse_pixel_object,
                                    reinterpret_cast<unsigned long long*>(tile_objects), (hipStream_t)stream);
}

DQO_API int dqo_attach_decide(int32_t n, const float* temp_xyz, const float* temp_opacity, const int32_t* temp_object, const int32_t* lin,
                              const int32_t* hit_index, const float* hit_weight, const float* xyz, const float* scaling_raw,
                              const float* rotation_raw, const int32_t* gaussian_object, float plane_thr, float opacity_low, uint8_t* out,
                              void* stream) {
    DQO_CHECK_ARG(n >= 0, "bad size");
    if (n == 0) return DQO_OK;
    DQO_CHECK_ARG(temp_xyz && temp_opacity && temp_object && lin && hit_index && hit_weight && xyz && scaling_raw && rotation_raw &&
                      gaussian_object && out, "null pointer");
    return dqo_launch_attach_decide(n, temp_xyz, temp_opacity, temp_object, lin, hit_index, hit_weight, xyz, scaling_raw, rotation_raw,
                                    gaussian_object, plane_thr, opacity_low, out, (hipStream_t)stream);
}

DQO_API int dqo_growth_scales(int32_t n, const float* xyz, const int32_t* object, const float* radius, const int32_t* i_new,
                              const float* d2_old, const int32_t* i_old, const float* extra_radius, float reach2, float min_radius,
                              float max_radius, float* scales, uint8_t* invalid, void* stream) {
    DQO_CHECK_ARG(n >= 0, "bad size");
    if (n == 0) return DQO_OK;
    DQO_CHECK_ARG(xyz && object && radius && scales && invalid, "null pointer");
    DQO_CHECK_ARG(i_old == nullptr || (d2_old && extra_radius), "i_old without d2_old / extra_radius");
    return dqo_launch_growth_scales(n, xyz, object, radius, i_new, d2_old, i_old, extra_radius, reach2, min_radius, max_radius, scales, invalid,
                                    (hipStream_t)stream);
}

DQO_API int dqo_growth_inside(int32_t n, const float* d2, const int32_t* idx, const float* radius, uint8_t* inside, void* stream) {
    DQO_CHECK_ARG(n >= 0, "bad size");
    if (n == 0) return DQO_OK;
    DQO_CHECK_ARG(d2 && idx && radius && inside, "null pointer");
    return dqo_launch_growth_inside(n, d2, idx, radius, inside, (hipStream_t)stream);
}

DQO_API int dqo_error_maps(int32_t H, int32_t W, const float* gt_color, const float* gt_depth, const float* render, const float* depth,
                           const int32_t* depth_index, const uint8_t* mask, float* color_err, float* depth_err, void* stream) {
    DQO_CHECK_ARG(W > 0 && H > 0 && gt_color && gt_depth && render && depth && depth_index && color_err && depth_err, "bad size / null pointer");
    return dqo_launch_error_maps((int64_t)W * H, gt_color, gt_depth, render, depth, depth_index, mask, color_err, depth_err, (hipStream_t)stream);
}

DQO_API size_t dqo_icp_workspace_bytes(void) { return dqo_icp_ws_bytes(); }

DQO_API int dqo_icp_normal_equations(int32_t H, int32_t W, const float* vertex0, const float* vertex1, const float* normal0,
                                     const float* normal1, const float* pose10, float fx, float fy, float cx, float cy, float dist_thr,
                                     float normal_thr, float* JtJ, float* JtR, int32_t* valid_count, void* ws, size_t ws_bytes,
                                     void* stream) {
    DQO_CHECK_ARG(H > 1 && W > 1 && (int64_t)H * W < (int64_t)0x7fffffff / 3, "bad image size");
    DQO_CHECK_ARG(vertex0 && vertex1 && normal0 && normal1 && pose10 && JtJ && JtR && valid_count, "null pointer");
    if (ws == nullptr || ws_bytes < dqo_icp_ws_bytes()) {
        dqo_set_error("icp workspace too small (%zu < %zu)", ws_bytes, dqo_icp_ws_bytes());
        return DQO_ERR_WORKSPACE;
    }
    return dqo_launch_icp(H, W, vertex0, vertex1, normal0, normal1, pose10, fx, fy, cx, cy, dist_thr, normal_thr, JtJ, JtR, valid_count, ws,
                          (hipStream_t)stream);
}

DQO_API int dqo_map_adam_step(const DqoAdamStep* st, void* stream) {
    const int rc = check_adam_step(st, true);
    if (rc) return rc;
    if (st->P == 0) return DQO_OK;
    return dqo_launch_map_adam(st, (hipStream_t)stream);
}

DQO_API int dqo_quadric_adam(int32_t n_obj, int32_t n_iters, const int32_t* view_offset, const float* P34_views, const float* obs_views,
                             const int32_t* view_schedule, float* axes, float* R, float* center, float* loss_hist, void* stream) {
    DQO_CHECK_ARG(n_obj >= 0 && n_iters >= 0, "bad sizes");
    if (n_obj == 0 || n_iters == 0) return DQO_OK;
    DQO_CHECK_ARG(view_offset && P34_views && obs_views && view_schedule && axes && R && center, "null pointer");
    return dqo_launch_quadric_adam(n_obj, n_iters, view_offset, P34_views, obs_views, view_schedule, axes, R, center, loss_hist,
                                   (hipStream_t)stream);
}

}  // extern "C"
