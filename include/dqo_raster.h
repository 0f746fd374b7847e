/*
 * dqo_raster.h — C ABI of libdqoraster.so: the MI355X (gfx950) drop-in for DQO-MAP's mapping hot path.
 *
 * Every entry point replaces one binding of the reference's three CUDA extensions (paths relative to
 * /root/reference):
 *
 *   dqo_rast_forward_*   <- _C_depth.rasterize_gaussians           submodules/diff-gaussian-rasterizer-depth/ext.cpp:16,
 *                                                                  rasterize_points.cu:37-155, cuda_rasterizer/rasterizer.h:35-73
 *   dqo_rast_backward    <- _C_depth.rasterize_gaussians_backward  ext.cpp:17, rasterize_points.cu:157-249, rasterizer.h:75-105
 *   dqo_mark_visible     <- _C_depth.mark_visible                  ext.cpp:18, rasterize_points.cu:251-270, rasterizer.h:27-33
 *   dqo_knn3             <- simple_knn._C.distCUDA2                submodules/simple-knn/ext.cpp:15-17, spatial.cu:15-28
 *   dqo_quadric_*        <- Ellipsoid_tensor.forward + bboxes_iou + the Adam loop of Object_Optimize_only
 *                                                                  SLAM/multiprocess/quadrics.py:285-290, 2018-2091, 2144-2220, 2234-2298
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (tensor.data_ptr()), sizes, no torch / C++ types; all memory is caller-owned.
 *   - every launch goes to the caller's stream (`hipStream_t` passed as void*); no hipMalloc/hipFree, no host
 *     synchronisation inside any entry point except dqo_rast_read_header (explicitly a D2H read).
 *   - return value: 0 on success, negative DqoStatus on error; dqo_last_error() gives the message (thread-local).
 *     The reference throws C++ exceptions -> Python RuntimeError (rasterize_points.cu:67-70); the Python shim does the same.
 *   - fp32 everywhere, matrices in the reference's row-vector convention (flat 16 floats, auxiliary.h:59-77).
 */
#ifndef DQO_RASTER_H_
#define DQO_RASTER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQO_ABI_VERSION 5

typedef enum DqoStatus {
    DQO_OK = 0,
    DQO_ERR_INVALID_ARG = -1,  /* bad shape / null pointer / unsupported combination */
    DQO_ERR_WORKSPACE = -2,    /* a caller-provided buffer is too small */
    DQO_ERR_LAUNCH = -3,       /* hipLaunch / hipMemsetAsync failed (message holds hipGetErrorString) */
    DQO_ERR_OVERFLOW = -4      /* instance capacity exceeded (reported by dqo_rast_read_header) */
} DqoStatus;

/* Scalar settings: GaussianRasterizationSettings, diff_gaussian_rasterization_depth/__init__.py:288-307. */
typedef struct DqoRastParams {
    int32_t P;              /* number of Gaussians (means3D.size(0)) */
    int32_t D;              /* active SH degree 0..3 */
    int32_t M;              /* SH coefficients per Gaussian (sh.size(1)); 0 when colours are precomputed */
    int32_t W, H;           /* image size */
    int32_t prefiltered;    /* accepted, unused (reference only traps on it) */
    int32_t debug;          /* accepted, unused */
    float tanfovx, tanfovy;
    float cx, cy;           /* principal point in pixels */
    float scale_modifier;
    float color_sigma;      /* radius = ceil(color_sigma * sqrt(lambda_max)) */
    float opaque_threshold; /* alpha of the first Gaussian that fixes the depth */
    float depth_threshold;  /* hit_depth_threshold */
    float normal_threshold; /* hit_normal_threshold = cos(angle) */
    float T_threshold;
} DqoRastParams;

/* Borrowed device inputs.  Exactly one of (shs, colors_precomp) is non-NULL.  scales and rotations are required: the
 * reference's blend kernel dereferences them unconditionally (forward.cu:780), so cov3D_precomp-only calls are rejected
 * with DQO_ERR_INVALID_ARG instead of faulting. */
typedef struct DqoRastInputs {
    const float* bg;             /* [3] */
    const float* means3D;        /* [P,3] */
    const float* shs;            /* [P,M,3] or NULL */
    const float* colors_precomp; /* [P,3] or NULL */
    const float* opacities;      /* [P] */
    const float* scales;         /* [P,3] */
    const float* rotations;      /* [P,4] (r,x,y,z), already normalised by the caller */
    const float* cov3D_precomp;  /* must be NULL */
    const float* viewmatrix;     /* [16] */
    const float* projmatrix;     /* [16] */
    const float* campos;         /* [3] */
    const int32_t* tile_mask;    /* [ceil(H/16) * ceil(W/16)], non-zero = render; NULL = all ones */
    /* ABI 5.  NULL (the drop-in behaviour: every row is rendered) or one byte per Gaussian; a row with DQO_ROW_HIDDEN set is treated
     * as culled by the per-Gaussian forward (radii 0, no list entry, no gradient) — the reference's two clouds kept in ONE map:
     * Mapping.global_optimization renders the stable cloud alone (SLAM/multiprocess/mapper.py:1199-1204, stable_params), local_optimize
     * renders cat(unstable, stable) (:578, :1810-1840).  Device memory, read when the launch runs: a captured iteration follows a
     * caller that rewrites the bytes in place between two mapping calls.  The same array serves DqoAdamStep.row_flags. */
    const uint8_t* row_flags;
} DqoRastInputs;
#define DQO_ROW_FROZEN 1u /* DqoAdamStep.row_flags: the row is not trained (no gradient row, no Adam, no confidence count) */
#define DQO_ROW_HIDDEN 2u /* DqoRastInputs.row_flags: the row is not rendered */

/* Caller-allocated outputs; the forward writes EVERY element (masked / empty tiles get the reference's initial
 * fills: colour 0, depth 0, ids 0, weights 0, T 1 — rasterize_points.cu:79-89), so torch.empty is enough. */
typedef struct DqoRastOutputs {
    float* out_color;            /* [3,H,W] */
    float* out_depth;            /* [1,H,W] */
    int32_t* out_hit_color;      /* [1,H,W] id of the max-weight colour contributor, -1 if none */
    int32_t* out_hit_depth;      /* [1,H,W] id of the Gaussian that fixed the depth, -1 if none */
    float* out_hit_color_weight; /* [1,H,W] */
    float* out_hit_depth_weight; /* [1,H,W] */
    float* out_T;                /* [1,H,W] */
    int32_t* n_touched;          /* [P] */
    int32_t* radii;              /* [P] */
} DqoRastOutputs;

/* Forward->backward context: the three opaque byte buffers the reference keeps as geomBuffer / binningBuffer /
 * imgBuffer (rasterize_points.cu:93-98).  Layout is private to this library; sizes come from the functions below. */
/* Optional loss tap of the mapping iteration (row f2; SLAM/multiprocess/mapper.py:836-875 with a render mask): the masked
 * 0.8 L1 colour + 1.0 depth L1 loss is evaluated where its inputs are produced and consumed instead of in two passes over the
 * images between forward and backward.  The forward's blend kernel adds up, per 8x8-pixel wave, |colour error| over the mask and
 * |depth error| over the valid depth pixels (fixed point: exact, order-independent, reproducible) and the two pixel counts.  The
 * backward's blend kernel reads the totals, forms each pixel's dL/dcolour and dL/ddepth from the rendered and the ground-truth
 * images as sign(error) x {color_weight / (3 n_colour), depth_weight / n_depth} — bit for bit what dqo_map_loss_fwd_bwd would have
 * written — so its dL_dout_color / dL_dout_depth arguments are not read (may be NULL), and writes loss_out[8] (same layout as
 * dqo_map_loss_fwd_bwd) and grad_scale[2] (the two factors): the loss is available after the BACKWARD call.  All pointers are
 * device pointers and must stay valid from the forward to the backward; out_color / out_depth are the forward's own outputs. */
typedef struct DqoLossTap {
    const float* gt_color;      /* [3,H,W] */
    const float* gt_depth;      /* [1,H,W] */
    const uint8_t* render_mask; /* [H,W] or NULL (= all pixels) */
    const float* out_color;     /* [3,H,W] = DqoRastOutputs.out_color of the forward (read by the backward) */
    const float* out_depth;     /* [1,H,W] = DqoRastOutputs.out_depth */
    float color_weight, depth_weight, add_depth_thres;
    float* loss_out;            /* [8] */
    float* grad_scale;          /* [2] */
    /* ABI 3.  0: ONE masked loss over all mask pixels (the reference's loss_update with a render mask, mapper.py:836-875).
     * non-zero, with DqoRastCtx.object_gate: the loss of the per-object job of SURVEY.md §8(e),  L = sum_k L_k,  L_k = the same masked
     * loss evaluated on object k's pixels alone (pixel_object == k inside the mask), each term normalised by ITS OWN pixel counts — so
     * that the loss (and every gradient) of an object does not depend on which other objects the caller renders with it: shards of one
     * map add up to the unsharded job exactly.  loss_out[0..2] = sum_k of (total, colour, depth); [4..7] the raw sums over all objects;
     * grad_scale is not written (the scales are per object).  Object ids must lie in [0, 64). */
    int32_t per_object;
} DqoLossTap;

/* Optional object gate (ABI 3; not a feature of the reference, whose renders composite every Gaussian of a ray — F3).  With it a list
 * entry acts on a pixel only if gaussian_object[id] == pixel_object[pixel] (a negative pixel id: no entry acts), in the forward and in
 * the backward: every pixel sees the render of its own object alone, whatever other Gaussians the call holds.  This is what makes the
 * per-object job of SURVEY.md §8(e) ONE function for every number of shards: a shard that owns some of the objects computes exactly its
 * objects' pixels of the unsharded render.  List positions (n_contrib, the hit position) count gated entries like skipped ones.  Both
 * arrays are device pointers and must stay valid from the forward to the backward. */
typedef struct DqoObjectGate {
    const int32_t* gaussian_object; /* [P], ids in [0, 64) */
    const int32_t* pixel_object;    /* [H*W] owner of every pixel, < 0 = none */
    /* Optional (NULL = not used): per 16x16 tile the set of owners among its pixels, bit k = some pixel of the tile has owner k
     * ([ceil(H/16) * ceil(W/16)] 64-bit words, derived from pixel_object by the caller — it only changes when pixel_object does).
     * The binning then drops a (Gaussian, tile) instance whose object owns no pixel of the tile: it could act on none. */
    const uint64_t* tile_objects;
} DqoObjectGate;

typedef struct DqoRastCtx {
    void* geom;
    size_t geom_bytes;
    void* binning;
    size_t binning_bytes;
    void* image;
    size_t image_bytes;
    int64_t inst_capacity; /* number of (Gaussian, tile) instances `binning` can hold */
    /* 0 (default): the per-tile lists are packed back to back by a prefix scan — any list length, `binning` sized by
     * dqo_rast_binning_bytes(inst_capacity).  > 0: every tile owns a fixed bucket of this many list entries (`binning` sized by
     * dqo_rast_binning_bytes_bucketed): the binning kernel writes an instance straight to tile * bucket + rank, no scan and no
     * placement pass sit between it and the sort.  A tile that outgrows its bucket raises the same overflow flag as running out
     * of inst_capacity (outputs invalid, nothing written out of bounds) — for callers that can re-run, like the captured mapping
     * iteration.  Same lists, same order, same results as the packed mode.  Capacity in bucket mode: the instance slots are handed
     * out by up to 64 regional allocators, each owning inst_capacity / regions slots, so a frame is also flagged invalid when ONE
     * region of the map outgrows its share although the total fits — size inst_capacity with headroom (the captured iteration
     * uses 3 x the measured N), or use the packed mode when the capacity is tight. */
    int32_t tile_bucket_capacity;
    /* Bucket mode only.  Non-zero: keep the tile launch order (longest list first, XCD bands) that an earlier forward left in
     * `image` instead of recomputing it — the caller guarantees that a forward with this flag at 0 has run on this `image` buffer
     * for the same W x H (the order only decides which workgroup renders which tile, never a result, so an order computed for a
     * slightly different state of the map is as good).  The one-block scan kernel between binning and sort then disappears: list
     * ranges follow from the tile counters, the header from per-line statistics.  For a captured iteration that is replayed. */
    int32_t keep_tile_order;
    /* NULL (default, the drop-in behaviour) or the loss tap described above; read by the forward and by the backward. */
    const DqoLossTap* loss_tap;
    /* NULL (default: the reference's semantics) or the object gate described above; read by the forward and by the backward. */
    const DqoObjectGate* object_gate;
    /* 0 (default): every 8x8-pixel quadrant of a tile blends its list front to back (and walks it back to front) in ONE wave, the
     * reference's order of arithmetic.  n > 0: lists longer than n entries (n < 64 counts as 64) are shared between eight waves — rounds
     * of eight chunks of 64 entries: per chunk the composed map of the per-pixel state (forward: the product of (1 - alpha) and "holds an
     * opaque hit"; backward: the scale of T and the affine map of the colour blended behind), a scan over the round, then the blend /
     * walk of every chunk from its scanned start state — for launches that cannot fill the GPU, a strong-scaling shard of a few hundred
     * tiles, whose time is the time of the one longest list.  Read by the forward AND by the backward call: the forward builds the
     * queue of long lists, the backward (given n > 0 too) walks that queue with eight waves per quadrant (and every shorter list exactly
     * like a backward given 0: same kernel code, same bits); a backward given 0 walks every list in one wave whatever the forward did.  The state is grouped by chunk, so results on those lists differ from the serial order
     * in the last bits (1e-7 relative; a pixel exactly on T_threshold may finish one entry earlier or later); shorter lists are treated
     * as with 0. */
    int32_t list_split;
    /* ABI 4.  Non-zero: the caller guarantees that the per-frame scalars of ctx.geom (slot allocator, queue counters, statistics lines,
     * loss-tap counters) are zero — as dqo_rast_backward_adam leaves them: its per-Gaussian kernel, the LAST consumer of a frame's
     * counters, clears them for the next frame on the way — so the forward does not launch its zero-fill kernel.  For a captured iteration that is
     * replayed back to back (forward, dqo_rast_backward_adam, forward, ...) on one context: one launch less per iteration.  Ignored when
     * P == 0.  The device header (num_rendered ... overflow) is never cleared: every frame rewrites it.
     * Round 5: that kernel also clears the tile histogram and tile flags of ctx.image and leaves a stamp behind; with per-tile buckets
     * (tile_bucket_capacity > 0) a pre-zeroed frame then has no per-Gaussian preprocess launch either — its statements run at the head of
     * the binning kernel (csrc/dqo_k1_early.h) — and, with buckets of at most 1024 entries and keep_tile_order, no long-list sort launch.
     * A frame that finds no stamp (the promise was broken: a forward-only render, dqo_rast_backward or an error return in between) is
     * flagged in header.overflow and trains nothing; the frame after it is valid again.  Between dqo_rast_forward_prepare and
     * dqo_rast_forward_render of such a frame the stage-1 statistics are not available yet (dqo_rast_read_header reports zeros): use
     * dqo_rast_forward / dqo_rast_forward_async.
     * Round 6: dqo_rast_backward reads the field too.  Non-zero there (and list_split == 0): its per-Gaussian kernel — the last
     * consumer of the frame's counters in that call — clears the same words and leaves the same stamp, so the NEXT forward on the
     * context may be given frame_prezeroed: the drop-in operator's pooled contexts (forward, dqo_rast_backward, forward, ...).  A second
     * dqo_rast_backward over the same forward (retain_graph) is fine: nothing it reads is cleared.  0 (default): nothing is cleared. */
    int32_t frame_prezeroed;
} DqoRastCtx;

/* Gradients (all caller-allocated, fully written by the backward; rasterize_points.cu:198-206).  dL_dcolors, dL_dcov3D and
 * dL_dmeans2D may be NULL when the caller has no use for them (the fused mapping step: no precomputed colours or covariances,
 * no densification statistics) — three scattered partial-line stores per visible Gaussian less. */
typedef struct DqoRastGrads {
    float* dL_dmeans3D;   /* [P,3] */
    float* dL_dsh;        /* [P,M,3] (NULL when M == 0) */
    float* dL_dcolors;    /* [P,3] or NULL */
    float* dL_dopacity;   /* [P,1] */
    float* dL_dscales;    /* [P,3] */
    float* dL_drotations; /* [P,4] */
    float* dL_dcov3D;     /* [P,6] or NULL */
    float* dL_dmeans2D;   /* [P,3] (x,y used; z = 0) or NULL */
    /* 0 (the drop-in behaviour): every row is written, zeros for the Gaussians the forward culled (radii == 0).
     * non-zero: those all-zero rows are left unwritten — for a consumer that looks at radii itself
     * (dqo_map_adam_step with DqoAdamStep.radii) this saves writing and re-reading 236 B per culled Gaussian. */
    int32_t skip_culled_rows;
} DqoRastGrads;

/* Host-visible copy of the device header kept at the start of ctx.geom. */
typedef struct DqoRastHeader {
    uint32_t num_rendered;   /* N: (Gaussian, tile) instances kept in the tile lists (valid after stage 2) */
    uint32_t num_tiles;      /* active (non-empty, unmasked) tiles */
    uint32_t overflow;       /* non-zero: N exceeded inst_capacity, results of this forward are invalid */
    uint32_t max_tile_count; /* longest per-tile list */
    uint32_t num_visible;    /* Gaussians with radius > 0 */
    uint32_t num_candidates; /* (Gaussian, tile) pairs inside the tile rects = the reference's num_rendered
                                (rasterizer_impl.cu:303-309); an upper bound of N, valid after stage 1 */
    uint32_t stage;          /* 1 after stage 1 (only num_visible / num_candidates are meaningful), 2 once stage 2 has written the
                                frame's header; dqo_rast_read_header reports the device header as it is at stage 2 */
    uint32_t reserved;
} DqoRastHeader;

int dqo_abi_version(void);
const char* dqo_last_error(void);
/* sizeof() of the ABI structs as this library was compiled (a binding checks its own struct definitions against it):
 * 0 DqoRastParams, 1 DqoRastInputs, 2 DqoRastOutputs, 3 DqoRastCtx, 4 DqoRastGrads, 5 DqoRastHeader, 6 DqoProfileEntry,
 * 7 DqoAdamStep, 8 DqoLossTap, 9 DqoObjectGate, 10 DqoAdamTensor; 0 for any other index. */
size_t dqo_abi_sizeof(int32_t which);

/* Optional per-kernel timing (measurement only; the reference has nothing comparable — it times whole frames with
 * time.time(), utils/monitor.py:22-37).  While enabled every kernel launch of this library is bracketed by HIP events
 * recorded on the launch stream; dqo_profile_collect waits for them and returns accumulated milliseconds per kernel. */
typedef struct DqoProfileEntry {
    char name[48];
    double total_ms;
    uint32_t calls;
} DqoProfileEntry;
int dqo_profile_enable(int on);
int dqo_profile_collect(DqoProfileEntry* out, int max_entries, int reset);

size_t dqo_rast_geom_bytes(int32_t P, int32_t W, int32_t H);
size_t dqo_rast_image_bytes(int32_t W, int32_t H);
size_t dqo_rast_binning_bytes(int64_t inst_capacity);
size_t dqo_rast_binning_bytes_bucketed(int64_t inst_capacity, int32_t W, int32_t H, int32_t tile_bucket_capacity);
size_t dqo_rast_backward_workspace_bytes(int64_t inst_capacity);

/* Stage 1 (per-Gaussian preprocess).  Needs ctx.geom and ctx.image; leaves num_candidates (>= N) in the device
 * header: the capacity a caller that wants a guaranteed fit allocates.  rasterizer_impl.cu:272-307 (K1, K2). */
int dqo_rast_forward_prepare(const DqoRastParams*, const DqoRastInputs*, DqoRastOutputs*, DqoRastCtx*, void* hipStream);
/* D2H read of the header (the ONLY synchronising call; replaces the reference's cudaMemcpy at rasterizer_impl.cu:307).  After
 * stage 1 (header.stage == 1): num_visible and num_candidates, everything else 0.  After stage 2 (stage == 2): the device header as
 * the frame's sort kernels wrote it — still valid after dqo_rast_backward / dqo_rast_backward_adam, which clear the counters the
 * header was formed from but not the header. */
int dqo_rast_read_header(const DqoRastCtx*, DqoRastHeader* host_out, void* hipStream);
/* Stage 2 (footprint test + tile binning, per-tile sort, blend).  Needs ctx.binning with inst_capacity >= N, otherwise
 * the device header's overflow flag is raised, nothing is written out of bounds and the outputs are invalid.
 * Stage 2 reads the inputs again (means3D, scales, rotations, shs / colors_precomp: the colour / surfel-normal part of the
 * per-Gaussian forward rides in the sort launches): a two-stage caller passes the SAME, unmodified, still-alive input arrays to
 * both stages.  With ctx.frame_prezeroed the caller promises that the previous frame on this ctx ended in dqo_rast_backward_adam
 * (the only call that restores the zeroed per-frame scalars); any other sequence must leave the flag at 0.
 * rasterizer_impl.cu:309-440 (K3-K6). */
int dqo_rast_forward_render(const DqoRastParams*, const DqoRastInputs*, DqoRastOutputs*, DqoRastCtx*, void* hipStream);
/* Both stages back to back, no host synchronisation (caller guarantees / later checks capacity). */
int dqo_rast_forward(const DqoRastParams*, const DqoRastInputs*, DqoRastOutputs*, DqoRastCtx*, void* hipStream);

/* Both stages in ONE call for a caller that carries its instance capacity over from earlier frames and checks it afterwards (ABI 4; the
 * drop-in op's 'lazy' / 'deferred' modes): as dqo_rast_forward, plus — where the frame's header is final, behind the sort kernels and
 * BEFORE the blend kernel — an asynchronous copy of the 32-byte device header to `header_host` (pinned host memory; NULL: no copy) and
 * hipEventRecord(header_event) on the launch stream (a hipEvent_t; NULL: none).  When the event has completed, header_host->overflow says
 * whether ctx.inst_capacity held the frame (num_rendered, num_tiles, max_tile_count, num_visible and num_candidates are final
 * too) — about one blend kernel earlier than a copy issued behind the call would.  Replaces the reference's blocking
 * cudaMemcpy of rasterizer_impl.cu:307 like dqo_rast_read_header does, without the host wait. */
int dqo_rast_forward_async(const DqoRastParams*, const DqoRastInputs*, DqoRastOutputs*, DqoRastCtx*, DqoRastHeader* header_host,
                           void* header_event, void* hipStream);

/* rasterizer_impl.cu:445-564 (K7-K9).  `hit_image` is out_hit_depth of the forward ([H*W]); workspace holds the
 * per-instance gradient records. */
int dqo_rast_backward(const DqoRastParams*, const DqoRastInputs*, const DqoRastCtx*, const float* dL_dout_color,
                      const float* dL_dout_depth, const int32_t* hit_image, DqoRastGrads*, void* workspace,
                      size_t workspace_bytes, void* hipStream);

int dqo_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* hipStream);

/* Exact 3-NN: mean of the 3 smallest squared distances and the 3 neighbour indices (ascending distance, ties by
 * Morton order, INT_MAX / FLT_MAX when fewer than 3 other points exist). */
size_t dqo_knn3_workspace_bytes(int32_t P);
int dqo_knn3(int32_t P, const float* xyz, float* mean_d2, int32_t* idx3, void* workspace, size_t workspace_bytes,
             void* hipStream);

/* Row f3: exact K = 3 nearest REFERENCE points of every QUERY point (Mapping.temp_points_filter, SLAM/multiprocess/mapper.py:
 * 1351-1380 -> pytorch3d.ops.knn_points(K=3, norm=2)): dist2[Q,3] squared L2 distances ascending, idx3[Q,3] indices into the
 * reference set (FLT_MAX / -1 when it has fewer than 3 points).  Ties between equally distant references are resolved
 * arbitrarily. */
size_t dqo_knn3_query_workspace_bytes(int32_t Q, int32_t R);
int dqo_knn3_query(int32_t Q, const float* query_xyz, int32_t R, const float* ref_xyz, float* dist2, int32_t* idx3, void* workspace,
                   size_t workspace_bytes, void* hipStream);
/* ABI 4: the same search restricted to references closer than max_dist (> 0): a slot no such reference fills comes back as FLT_MAX / -1.
 * For callers whose decision saturates beyond a distance (the growth step's per-object decisions, dqo_mapgrowth: a neighbour further than
 * 0.087 m + 3 x its radius clips the scale like a missing one): the search starts from that bound instead of an open one, so a query with
 * no reference nearby prunes the whole map at once.  Same workspace as dqo_knn3_query. */
int dqo_knn3_query_within(int32_t Q, const float* query_xyz, int32_t R, const float* ref_xyz, float max_dist, float* dist2, int32_t* idx3,
                          void* workspace, size_t workspace_bytes, void* hipStream);
/* Row f3, the per-object job (not a reference feature): the same search where a reference point only counts for a
 * query of the SAME GROUP (int32 ids in [0, 64) — the Gaussians' object ids; any other value: the point belongs to no group and is
 * neither found nor finds anything) and, when `group_box` is given ([64][6] floats: lo xyz, hi xyz per group), only if it lies strictly
 * inside its group's box (SLAM/utils.py:801-808 bbox_filter, per object).  One search over the map as it is stored: no shifted copies,
 * no gathered subsets.  At most 2^25 - 1 reference points.  Same workspace as dqo_knn3_query. */
int dqo_knn3_query_grouped(int32_t Q, const float* query_xyz, const int32_t* query_group, int32_t R, const float* ref_xyz,
                           const int32_t* ref_group, const float* group_box, float max_dist, float* dist2, int32_t* idx3, void* workspace,
                           size_t workspace_bytes, void* hipStream);

/* Row f, the per-object job's temp_points_attach (SLAM/multiprocess/mapper.py:1384-1430; not a reference binding: the reference runs the
 * decision as a chain of torch ops) in two launches around the gated render of the stable cloud.
 *   dqo_attach_pixels: candidate n -> lin[n] = its pixel v * W + u (get_uv, scene/cameras.py:207-214: trunc(K (R x + t) / z); -1 = outside
 *     the image), and the sparse object gate of that render, both fully written: sparse_pixel_object [H*W] = pixel_object where a candidate
 *     projects, -1 elsewhere; tile_objects [ceil(H/16) * ceil(W/16)] = per tile the 64-bit set of the owners among those pixels
 *     (DqoObjectGate.pixel_object / .tile_objects).  viewmatrix: the 16 floats DqoRastInputs.viewmatrix takes; fx, fy, cx, cy: pixels.
 *   dqo_attach_decide: out[n] = 1 iff opacity[n] > opacity_low, the candidate is inside the image, hit_index at its pixel names a
 *     Gaussian s >= 0 (index 0 with hit_weight 0 = the op's zero fill: no hit) of the candidate's object, and
 *     |(xyz[s] - temp_xyz[n]) . normal(s)| < plane_thr (normal: SLAM/gaussian_pointcloud.py:780-791 from the raw quaternion and raw
 *     scales).  hit_index / hit_weight: DqoRastOutputs.out_hit_color / out_hit_color_weight of the gated render. */
int dqo_attach_pixels(int32_t n, const float* temp_xyz, const float* viewmatrix, float fx, float fy, float cx, float cy, int32_t W, int32_t H,
                      const int32_t* pixel_object, int32_t* lin, int32_t* sparse_pixel_object, uint64_t* tile_objects, void* hipStream);
int dqo_attach_decide(int32_t n, const float* temp_xyz, const float* temp_opacity, const int32_t* temp_object, const int32_t* lin,
                      const int32_t* hit_index, const float* hit_weight, const float* xyz, const float* scaling_raw,
                      const float* rotation_raw, const int32_t* gaussian_object, float plane_thr, float opacity_low, uint8_t* out,
                      void* hipStream);

/* Row f, small per-point kernels of the growth step (each replaces a chain of element-wise torch ops of the reference's callers; the
 * step is bound by the host's op issue rate):
 *   dqo_growth_scales: GaussianPointCloud.update_geometry's scale initialisation (SLAM/gaussian_pointcloud.py:519-556) behind its two
 *     searches, per-object job: for new point i the candidates are i_new[i][0..2] (indices among the n new points from dqo_knn3 on
 *     object-shifted coordinates; >= n: none; a neighbour counts if it has i's object id and lies within sqrt(reach2) — the distance is
 *     recomputed from xyz) and (d2_old, i_old)[i][0..2] (dqo_knn3_query_grouped against the existing map; i_old < 0: none); the three
 *     nearest give gaps g = dist - 3 * radius (radius / extra_radius of the neighbour); invalid = any g < 0; scale =
 *     clip(sqrt(mean g^2), min_radius, max_radius).  i_new or i_old may be NULL (no such search).
 *   dqo_growth_inside: Mapping.temp_points_filter's decision (SLAM/multiprocess/mapper.py:1372-1380): inside[i] = some idx[i][k] >= 0
 *     with sqrt(d2[i][k]) < 0.6 * radius[idx[i][k]].
 *   dqo_error_maps: the per-pixel error images of mapper.py:1016-1033 for dqo_accumulate_gaussian_error: depth_err = max(gt_depth -
 *     depth, 0), color_err = sum over channels |gt_color - render|; both 0 where gt_depth == 0 or mask == 0 (mask NULL = all), depth_err
 *     also where depth_index == -1.  Images are [C, H*W] planes. */
int dqo_growth_scales(int32_t n, const float* xyz, const int32_t* object, const float* radius, const int32_t* i_new, const float* d2_old,
                      const int32_t* i_old, const float* extra_radius, float reach2, float min_radius, float max_radius, float* scales,
                      uint8_t* invalid, void* hipStream);
int dqo_growth_inside(int32_t n, const float* d2, const int32_t* idx, const float* radius, uint8_t* inside, void* hipStream);
int dqo_error_maps(int32_t H, int32_t W, const float* gt_color, const float* gt_depth, const float* render, const float* depth,
                   const int32_t* depth_index, const uint8_t* mask, float* color_err, float* depth_err, void* hipStream);

/* Batched dual-quadric residual over B independent (object, view) pairs: loss = 1 - IoU(obs, bbox(ellipsoid, P34)),
 * with gradients.  valid[b] = 0 when loss == 1 (the reference skips that Adam step). */
int dqo_quadric_iou_fwd_bwd(int32_t B, const float* axes, const float* R, const float* center, const float* P34,
                            const float* obs_bbox, float* bbox, float* loss, int32_t* valid, float* g_axes, float* g_R,
                            float* g_center, void* hipStream);
/* Object_Optimize_only inner loops for n_obj objects in ONE launch: n_iters Adam steps each (lr .01/.001/.01,
 * betas .9/.999, eps 1e-15), view picked by view_schedule[obj][it] (negative = from the end).  Parameters are updated
 * in place; loss_hist [n_obj, n_iters] may be NULL.  Views of object o are P34_views / obs_views rows
 * [view_offset[o], view_offset[o+1]). */
int dqo_quadric_adam(int32_t n_obj, int32_t n_iters, const int32_t* view_offset, const float* P34_views,
                     const float* obs_views, const int32_t* view_schedule, float* axes, float* R, float* center,
                     float* loss_hist, void* hipStream);

/* ---- fused helpers around the rasteriser for one mapping iteration (optional) ------------------
 * They replace eager torch op sequences of the reference's callers, not a CUDA binding:
 *   dqo_map_activate      <- SLAM/gaussian_pointcloud.py:732-733, 746-747 (sigmoid / exp / F.normalize)
 *   dqo_map_loss_fwd_bwd  <- SLAM/multiprocess/mapper.py:836-875 with a render mask (masked L1 colour + masked depth L1; SSIM
 *                            is skipped in that case, B14) and its autograd backward
 *   dqo_map_ssim_fwd_bwd  <- SLAM/multiprocess/mapper.py:839-845 without a render mask: the SSIM term (utils/loss_utils.py:41-100)
 *   dqo_map_adam_step     <- autograd through the activations + torch.optim.Adam(eps=1e-15) over the six parameter groups
 *                            (SLAM/gaussian_pointcloud.py:331-378, mapper.py:548) */
int dqo_map_activate(int32_t P, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* opacity,
                     float* scales, float* rotations, void* hipStream);

size_t dqo_map_loss_workspace_bytes(void);
/* color [3,H,W], depth [1,H,W], depth_index int32 [1,H,W] (the op's hit_depth), render_mask uint8 [H,W] or NULL (= all).
 * loss_out[8] = {total, colour, depth, 0, sum |colour error| over the mask, mask pixels, sum |depth error| over the valid depth
 * pixels, valid depth pixels} — the four sums are what object shards of one map add up (one packed all-reduce) to report the loss
 * over all objects; writes dL_dcolor [3,H,W] and dL_ddepth [1,H,W] of `total`. */
int dqo_map_loss_fwd_bwd(int32_t W, int32_t H, const float* color, const float* depth, const int32_t* depth_index,
                         const float* gt_color, const float* gt_depth, const uint8_t* render_mask, float color_weight,
                         float depth_weight, float add_depth_thres, float* loss_out, float* dL_dcolor, float* dL_ddepth,
                         void* workspace, size_t workspace_bytes, void* hipStream);

/* The SSIM term of Mapping.loss_update's unmasked branch (SLAM/multiprocess/mapper.py:839-845: loss += 0.2 * (1 - ssim(image, gt)),
 * ssim = utils/loss_utils.py:41-100: 11 x 11 Gaussian window, sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2, mean over all
 * channels and pixels) with its gradient, in three launches in place of the five conv2d calls, ~25 elementwise ops and their autograd
 * backward.  image, gt_image [3,H,W]; ssim_out[2] = {ssim, weight * (1 - ssim)}; dL_dimage [3,H,W] (may be NULL: value only) receives
 * d(weight * (1 - ssim)) / d image — written when accumulate == 0, added onto what it holds otherwise (e.g. onto the L1 gradient
 * image dqo_map_loss_fwd_bwd wrote).  loss_out8 (may be NULL) = the loss_out of a dqo_map_loss_fwd_bwd call issued before on the same
 * stream: its total [0] grows by the term and slot [3] receives 1 - ssim, so the eight floats read like Mapping.loss_update's report.
 * The sum over pixels is formed per 16 x 16 tile and then over tiles in a fixed order: reproducible. */
size_t dqo_map_ssim_workspace_bytes(int32_t W, int32_t H);
int dqo_map_ssim_fwd_bwd(int32_t W, int32_t H, const float* image, const float* gt_image, float weight, float* ssim_out,
                         float* dL_dimage, int32_t accumulate, float* loss_out8, void* workspace, size_t workspace_bytes,
                         void* hipStream);

/* The attach loss of Mapping.loss_update (SLAM/multiprocess/mapper.py:812-829) with its gradient, for callers that keep their own
 * optimiser (ABI 3):  loss[0] = 1000 * (mse(scaling[a], scaling0[a]) + mse(xyz[a], xyz0[a]) + mse(rotation[a], rotation0[a])),  a =
 * attach_mask != 0 with |a| = attach_count (0: loss and gradients are zero), on the RAW parameters; g_* ([P,3], [P,3], [P,4]) are fully
 * written (zeros outside a).  Two launches in place of the ~12 eager torch ops of the forward and the ~25 of their autograd backward. */
size_t dqo_map_attach_workspace_bytes(int32_t P);
int dqo_map_attach_loss_fwd_bwd(int32_t P, const float* scaling_raw, const float* xyz, const float* rotation_raw,
                                const float* init_scaling_raw, const float* init_xyz, const float* init_rotation_raw,
                                const uint8_t* attach_mask, int32_t attach_count, float* loss, float* g_scaling_raw, float* g_xyz,
                                float* g_rotation_raw, void* workspace, size_t workspace_bytes, void* hipStream);

/* torch.optim.Adam's step (weight_decay = 0, amsgrad = False, maximize = False) over up to DQO_ADAM_MULTI_MAX dense fp32 tensors in
 * ONE launch, for callers that keep the reference's parameter tensors and autograd (ABI 3): per tensor p, its gradient g (w.r.t. p
 * itself — no activation Jacobian is applied here, unlike dqo_map_adam_step), exp_avg m, exp_avg_sq v, element count n and the
 * group's learning rate; `step` is the 1-based step count of THIS update, common to the tensors of the call (gaussian_pointcloud.py:
 * 331-378 builds ONE Adam over six groups with eps = 1e-15, so they share it).  The element update is the library's adam1 — the
 * statement order of torch's single- / multi-tensor paths; betas and eps arrive as doubles and 1 - beta, beta^t, lr / (1 - beta1^t) are
 * formed in double and rounded once, as torch forms them from python floats.  In place
 * of torch.optim.Adam.step()'s per-group launches (and their host time: six groups = 0.28 ms per iteration on the drop-in path). */
#define DQO_ADAM_MULTI_MAX 16
typedef struct DqoAdamTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
    double lr;  /* the group's learning rate as the python float it is: lr / (1 - beta1^t) is formed in double and rounded once, as torch does */
} DqoAdamTensor;
int dqo_adam_multi(const DqoAdamTensor* tensors, int32_t n_tensors, int32_t step, double beta1, double beta2, double eps, void* hipStream);
/* The same with the step count on the device, for a step that is captured into a hipGraph (a replay must take the count of ITS step,
 * not the capture's): step_dev[0] = steps taken so far; the launch applies step step_dev[0] + 1 — its bias corrections are formed by the
 * kernel with the host path's double-precision expressions (torch's capturable Adam forms them in float) — and, with advance != 0, a
 * one-thread launch behind it stores the new count (advance = 0 on all but the last call when one step takes several calls). */
int dqo_adam_multi_dev(const DqoAdamTensor* tensors, int32_t n_tensors, int32_t* step_dev, int32_t advance, double beta1, double beta2,
                       double eps, void* hipStream);

/* dqo_accumulate_gaussian_error <- cuda_utils._C.accumulate_gaussian_error (submodules/cuda_utils/ext.cpp, cuda_utils.cu:17-62,
 * map_process.cu:33-245; caller SLAM/multiprocess/mapper.py:1034-1047).  Maps are [H*W]; outputs [P] are fully written
 * (zero-initialised inside).  check_max != 0: per-Gaussian maximum of the errors (the mode DQO-MAP uses); 0: mean, which needs
 * `counters` (int32 [2*P] scratch).  rescale_counter counts pixels whose error exceeds the thresholds. */
int dqo_accumulate_gaussian_error(int32_t H, int32_t W, int32_t P, const float* screen_color_error, const float* screen_depth_error,
                                  const float* screen_normal_error, const int32_t* screen_color_index,
                                  const int32_t* screen_depth_index, float color_threshold, float depth_threshold,
                                  float normal_threshold, int32_t check_max, float* gs_color_error, float* gs_depth_error,
                                  float* gs_normal_error, float* gs_rescale_counter, int32_t* counters, void* hipStream);

/* dqo_accumulate_gaussian_confidence <- cuda_utils._C.accumulate_gaussian_confidence (submodules/cuda_utils/ext.cpp:7,
 * cuda_utils.cu:62-83, map_process.cu:247-360; exported by the reference's extension, no Python caller in the reference).  Maps are
 * [H*W]; per Gaussian named by gaussian_index_map (entries outside [0, P) are skipped): maximum, minimum and mean of the confidence
 * over its pixels, 0 / 0 / 0 for a Gaussian no pixel names.  Outputs [P] fully written; `counter` is int32 [P] scratch. */
int dqo_accumulate_gaussian_confidence(int32_t H, int32_t W, int32_t P, const int32_t* gaussian_index_map,
                                       const float* gaussian_confidence_map, float* gs_confidence_max, float* gs_confidence_min,
                                       float* gs_confidence_mean, int32_t* counter, void* hipStream);

/* Row f3 — tile-mask producers of the mapping loop (SLAM/utils.py:720-799, SLAM/multiprocess/mapper.py:930-988), 16x16 tiles
 * with the reference's zero padding (a tile is always divided by 256).  All pointers are device pointers.
 *   dqo_tile_count_mask:   tile_count[t] = number of non-zero pixels of a uint8 pixel mask in tile t
 *                          (pixelmask2tilemask = count > 0; transmission2tilemask = count / 256 > ratio)
 *   dqo_transmission_mask: render_mask = (T_map != 1) (written if non-NULL), tile_count as above, *total = its pixel count
 *                          (evaluate_render_range: render_mask, tile_mask, render_ratio in one pass)
 *   dqo_tile_color_error:  color_error = sum_c |render - gt|, zeroed where the rendered colour sums to 0 (written if
 *                          non-NULL); tile_sum[t] = its sum over tile t (meanpool / colorerror2tilemask = tile_sum / 256) */
int dqo_tile_count_mask(int32_t W, int32_t H, const uint8_t* pixel_mask, int32_t* tile_count, void* hipStream);
int dqo_transmission_mask(int32_t W, int32_t H, const float* T_map, uint8_t* render_mask, int32_t* tile_count, int32_t* total,
                          void* hipStream);
int dqo_tile_color_error(int32_t W, int32_t H, const float* render, const float* gt, float* color_error, float* tile_sum,
                         void* hipStream);

/* Row f4 — normal equations of one Gauss-Newton iteration of the point-to-plane ICP tracker (SLAM/icp.py:51-123:
 * compute_residuals_jacobian + compute_jtj + compute_jtr).  vertex / normal maps are [H, W, 3] fp32, pose10 a row-major 4x4
 * (maps frame-0 points into frame 1), normal_threshold the cosine.  Out: JtJ [6,6] (rotation block first), JtR [6],
 * valid_count [1] = pixels that passed every test.  The 6x6 solve / se(3) update stay with the caller. */
size_t dqo_icp_workspace_bytes(void);
int dqo_icp_normal_equations(int32_t H, int32_t W, const float* vertex0, const float* vertex1, const float* normal0, const float* normal1,
                             const float* pose10, float fx, float fy, float cx, float cy, float distance_threshold,
                             float normal_threshold, float* JtJ, float* JtR, int32_t* valid_count, void* workspace,
                             size_t workspace_bytes, void* hipStream);

#define DQO_TICKET_WORDS (16 + 16 * 64) /* int32 words behind DqoAdamStep.block_ticket */
typedef struct DqoAdamStep {
    int32_t P, M;      /* Gaussians, SH coefficients per Gaussian (f_dc = coefficient 0, f_rest = the others) */
    int32_t step;      /* 1-based Adam step count */
    /* betas travel as floats; 1 - beta and beta^t are formed in double from the decimal the float was written as (rounded to seven
     * decimals: 0.9, 0.999 and every other short decimal come back exactly; a beta that is not a short decimal is used to seven
     * decimals — dqo_adam_multi takes doubles and has no such limit) */
    float beta1, beta2, eps;
    float lr_xyz, lr_f_dc, lr_f_rest, lr_opacity, lr_scaling, lr_rotation;
    float *xyz, *shs, *opacity_raw, *scaling_raw, *rotation_raw;            /* raw parameters, updated in place */
    const float *g_means3D, *g_sh, *g_opacity, *g_scales, *g_rotations;      /* dqo_rast_backward outputs (w.r.t. activated) */
    float *m_xyz, *m_shs, *m_opacity, *m_scaling, *m_rotation;               /* exp_avg, same shapes as the parameters */
    float *v_xyz, *v_shs, *v_opacity, *v_scaling, *v_rotation;               /* exp_avg_sq */
    /* Optional (NULL = skip): the activated values of the UPDATED parameters, exactly what dqo_map_activate would
     * compute from them — saves that launch in the next iteration. */
    float *act_opacity, *act_scales, *act_rotations;
    /* Optional (NULL = every gradient row is read): the forward's radii of this step.  Gradient rows of culled Gaussians
     * (radii == 0) are taken as zero without being read (pair with DqoRastGrads.skip_culled_rows); their parameters and
     * moments are still updated exactly as dense Adam does with a zero gradient. */
    const int32_t* radii;
    /* Optional (NULL = use `step`): device scalar holding the 1-based step count of THIS launch; the launch is followed
     * by a one-thread kernel that increments it.  With it a whole mapping iteration has no host-side per-iteration
     * argument and can be captured once in a hipGraph and replayed. */
    int32_t* step_dev;
    /* Optional (NULL = dense): one byte per Gaussian, 0 = both moment rows of the Gaussian are identically zero (the state of a
     * freshly built optimiser, which the reference builds per mapping call, mapper.py:548).  Such a Gaussian with no gradient
     * (radii == 0) is a fixed point of Adam — m, v stay 0 and p - step * 0 / (0 + eps) = p bit for bit — so its rows are neither
     * read nor written; the first gradient sets its byte to 1.  Needs `radii`.  Results are identical to the dense update. */
    uint8_t* moment_live;
    /* Optional (NULL = no such term): the attach loss of Mapping.loss_update (SLAM/multiprocess/mapper.py:812-829),
     *   1000 * (mse(_scaling[a], scaling0[a]) + mse(_xyz[a], xyz0[a]) + mse(_rotation[a], rotation0[a])),
     * a = Gaussians whose opacity at the start of the mapping call (init_stat, mapper.py:533-545) was below 0.9.  Its gradient is
     * elementwise on the RAW parameters and is added here, after the activation Jacobians: attach_mask [P] (1 = in a),
     * init_* = the raw parameters at the start of the call, attach_count = |a| (the means divide by 3 |a|, 3 |a|, 4 |a|).
     * attach_partial (optional, [ceil(P / 256)]): per-block sum of 1000 * (d_scaling^2 / (3|a|) + d_xyz^2 / (3|a|) + d_rot^2 /
     * (4|a|)) over the block's Gaussians at the PRE-update parameters — their sum is the reported "scale_loss" of this
     * iteration (fixed order, reproducible).  A Gaussian that has never moved contributes an exact zero, so the exact sparse
     * mode stays exact. */
    const uint8_t* attach_mask;
    const float *init_xyz, *init_scaling_raw, *init_rotation_raw;
    int32_t attach_count;
    float* attach_partial;
    /* Optional (NULL = always step): device header of the forward whose gradients this step consumes (start of ctx.geom).  If
     * its overflow flag is set — the frame was invalid: lists emptied, all gradients zero — the launch is a no-op: parameters,
     * moments, moment_live and the device step count stay as they were, so the caller can re-capture with a larger capacity and
     * continue from a clean optimiser state. */
    const DqoRastHeader* frame_header;
    /* Optional, with step_dev: DQO_TICKET_WORDS int32 (ABI 5; one word until ABI 4), zero before the first launch (they are zero again
     * after every launch).  The block that finishes last advances *step_dev inside the Adam launch itself — tickets are taken on up
     * to 64 lines and then on word 0, so that no single address is hit by every block; NULL = a separate one-thread kernel does it
     * afterwards. */
    int32_t* block_ticket;
    /* Optional, with step_dev and block_ticket (ABI 3): eight floats, zero before the first launch.  The bias corrections of a step
     * (two double-precision pow() calls + six divisions) are then computed ONCE per step — by the block that advances *step_dev, for the
     * step it advances to — instead of once per block of every launch; a launch whose step the table does not hold (the first one,
     * or after the caller rewrote *step_dev) computes them itself.  Same function either way: same bits. */
    float* bias_table;
    /* Optional (ABI 3): two device floats { 2000 / (3 |a|), 2000 / (4 |a|) } (computed in double, rounded to float), read when the
     * launch RUNS instead of deriving them from attach_count when it is issued — for a captured launch (hipGraph) whose attach set
     * changes between replays: the caller rewrites attach_mask, init_* and these two numbers in place at the start of a mapping call
     * and replays the same graph.  With it, the attach term is on whenever attach_mask is given (|a| = 0: write two zeros). */
    const float* attach_gains;
    /* ABI 5 — the reference trains ONE of its two clouds per mapping call while it renders both (local_optimize: pointcloud.parametrize,
     * SLAM/multiprocess/mapper.py:533; global_optimization: stable_pointcloud.parametrize, :1119).  Optional (NULL = every row trains):
     * one byte per Gaussian; a row with DQO_ROW_FROZEN set is no parameter of the optimiser: its gradient row is not formed, its
     * parameters, moments, moment_live byte and confidence stay bit for bit as they are, it is no member of the attach set whatever
     * attach_mask says.  Device memory, read when the launch runs (see DqoRastInputs.row_flags). */
    const uint8_t* row_flags;
    /* Optional (NULL = not counted): float [P], the reference's per-Gaussian confidence (SLAM/gaussian_pointcloud.py:42, :836).  A trained
     * row whose f_dc gradient of this step has a non-zero element gains 1 — Mapping.loss_update's
     *     grad_mask = (pointcloud._features_dc.grad.abs() != 0).any(dim=-1);  pointcloud._confidence[grad_mask] += 1   (mapper.py:908-910)
     * — inside the launch that consumes the gradient (the fused tail never writes the row to HBM).  An invalid frame counts nothing. */
    float* confidence;
    /* Optional (NULL = the six lr_* fields above): six device floats { xyz, f_dc, f_rest, opacity, scaling, rotation }, read when the
     * launch runs — global_optimization rescales the groups' learning rates per call (mapper.py:1120-1131: xyz 0, the others x 0.1 or
     * x their *_lr_coef); a caller that rewrites the table (and zeroes bias_table) between two mapping calls keeps its captured graph. */
    const float* lr_table;
} DqoAdamStep;
int dqo_map_adam_step(const DqoAdamStep*, void* hipStream);

/* Mapping.history_merge (SLAM/multiprocess/mapper.py:607-650), the statement that closes every local_optimize call: the trained cloud is
 * pulled back towards its state at the start of the call (`history_stat`, :535-545), weighted by how much of its confidence is old:
 *     w[i] = max_weight * conf0[i] / (conf[i] + 1e-6)
 *     xyz[i]      = xyz0[i] * w[i] + (1 - w[i]) * xyz[i]
 *     f_dc, f_rest, scaling: the same lerp with w[first_row] FOR EVERY ROW — the reference indexes `history_weight[0]` (:620-637), a [1]
 *                  tensor that broadcasts: the first Gaussian's weight serves the whole cloud (reproduced, not fixed)
 *     rotation[i] = slerp(rot0_unit[i], normalize(rotation[i]), 1 - w[i])     (SLAM/utils.py:650-709: |dot| > 0.9995 or NaN -> torch.lerp,
 *                  otherwise sin-weighted; no shortest-arc flip, no renormalisation) — the result replaces the RAW quaternion.
 * One launch in place of ~40 eager torch ops.  All tensors are updated in place; rot0_unit = the ACTIVATED (normalised) rotation at the
 * start of the call (`history_stat["rotation"]` = get_rotation), shs0 / shs are [P, M, 3] (coefficient 0 = f_dc, the rest f_rest).  rows:
 * optional row_flags (NULL = all rows): rows with DQO_ROW_FROZEN are left untouched (they belong to the cloud the call did not train);
 * first_row = the row whose weight serves the broadcast quirk (row 0 of the reference's trained cloud).  max_weight <= 0: no-op (:608-609). */
int dqo_map_history_merge(int32_t P, int32_t M, float max_weight, int32_t first_row, const uint8_t* row_flags, const float* conf0,
                          const float* conf, const float* xyz0, const float* shs0, const float* scaling0, const float* rot0_unit, float* xyz,
                          float* shs, float* scaling_raw, float* rotation_raw, void* hipStream);

/* Fused mapping iteration, backward half (ABI 3): the blend kernel of dqo_rast_backward followed by ONE kernel that, per block of 256
 * Gaussians, sums the per-instance gradient records, runs the per-Gaussian backward (rasterizer_impl.cu:445-564's K8 + K9,
 * backward.cu:273-548) and applies dqo_map_adam_step's update (SLAM/gaussian_pointcloud.py:331-378, SLAM/multiprocess/mapper.py:548,
 * 812-829) with the gradient rows held in LDS: neither the 59-float gradient rows nor the summed records reach HBM, two launches
 * fewer.  Parameters, moments, activations, moment_live and the device step count come out BIT-IDENTICAL to
 *     dqo_rast_backward(grads with skip_culled_rows = 1)  +  dqo_map_adam_step(step with radii = the forward's radii)
 * (same statements on the same operands).  Only step->attach_partial differs: here it must hold 4 * ceil(P / 256) floats — one
 * partial sum per wave of 64 Gaussians instead of one per block of 256 — whose total agrees with dqo_map_adam_step's to rounding.
 * `step`: the g_* and radii fields are not read (visibility comes from ctx); frame_header is taken from ctx (an overflowed frame is a
 * no-op, as in dqo_map_adam_step).  inputs->shs is required (precomputed colours have no Adam group) with params->M == step->M <= 16
 * and params->P == step->P.  workspace as for dqo_rast_backward.  With a loss tap in ctx, dL_dout_* may be NULL. */
int dqo_rast_backward_adam(const DqoRastParams*, const DqoRastInputs*, const DqoRastCtx*, const float* dL_dout_color,
                           const float* dL_dout_depth, const DqoAdamStep* step, void* workspace, size_t workspace_bytes,
                           void* hipStream);

#ifdef __cplusplus
}
#endif
#endif /* DQO_RASTER_H_ */
