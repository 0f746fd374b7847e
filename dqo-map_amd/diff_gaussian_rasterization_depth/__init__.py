"""Drop-in replacement for DQO-MAP's `diff_gaussian_rasterization_depth` package on MI355X.

Same public surface as /root/reference/submodules/diff-gaussian-rasterizer-depth/diff_gaussian_rasterization_depth/
__init__.py: `GaussianRasterizationSettings` (:288-307, same field names / order / defaults), `GaussianRasterizer`
(:310-376, `.forward(...)`, `.markVisible(...)`), `rasterize_gaussians` (:29-50) and the autograd Function
`_RasterizeGaussians` (:53-285) with the same 9-tuple of outputs and the same gradient slots, so SLAM/render.py:8-13
imports and calls it unchanged.  Below the Python surface the pybind module `_C_depth` (ext.cpp:15-19) is replaced by
the C ABI of libdqoraster.so (include/dqo_raster.h) called through ctypes with raw device pointers.

Differences that are not visible to callers:
  * the three context tensors (geomBuffer, binningBuffer, imgBuffer) keep their role (opaque uint8 tensors saved for
    backward) but have this library's layout;
  * the forward does ONE 4-byte device->host read (the instance count N, to size the binning buffer exactly as the
    reference does at rasterizer_impl.cu:307) instead of the reference's two reads plus a host loop over all tiles;
    `set_sync_mode("lazy")` removes even that one (capacity carried over from the previous call, overflow detected
    and raised at the next synchronisation point — never silent; the backward waits for its forward's header so that no gradient of
    an invalid frame is ever produced), `set_sync_mode("deferred")` also drops that wait (an overflow is then raised by a later call,
    after at most a few iterations whose gradients were zero: see set_sync_mode);
  * in the 'lazy' / 'deferred' modes the three context tensors come from a pool per (device, stream, P, W, H) instead of being
    allocated per call (rasterize_points.cu:37-155 allocates them per call): see `_pool` below — four launches less per forward +
    backward pair, the same bits;
  * `tile_mask=None` is accepted and means "all tiles" (the reference requires a tensor).
There is no CPU path: tensors must live on the GPU and the HIP library must be built.
"""
import ctypes
from typing import NamedTuple

import torch
import torch.nn as nn

import _dqo_native as N

# Operator state.  It lives at module level on purpose: the reference's caller builds a NEW GaussianRasterizer for every render call
# (SLAM/render.py:163), so state hung off the module instance would not survive from one call to the next — and the lazy mode's capacity
# hint has to.  One lock guards it (tracker and mapper threads of one process may render concurrently).
import threading
import weakref

_lock = threading.RLock()
_list_split = 0         # DqoRastCtx.list_split of the forwards issued through this module (set_list_split)
_sync_mode = "exact"   # "exact": read N after the preprocess stage;  "lazy" / "deferred" / "graph": reuse the previous capacity, no sync
_cap_hint = {}
_pending = []          # lazy mode: (event, pinned header tensor, key, capacity, pooled context?) of forwards not yet verified
_last = {"num_rendered": None, "num_visible": None, "header": None}   # header: (event, pinned host copy) or a weak reference to the geometry buffer
_ring = {}             # per device: pinned 32-byte header buffers + their events, reused round robin (no pin_memory() / Event() per call)
_RING = 64
# Context pool ('lazy' / 'deferred' modes: the modes that carry state from call to call).  The reference allocates its three context
# buffers anew in every forward (rasterize_points.cu:37-155: resizeFunctional on fresh tensors); so did this op, and a fresh context knows
# nothing: its counters must be zeroed, its tile launch order computed, its lists packed by a scan.  A pooled context keeps, per
# (device, stream, P, W, H), what the fused mapping step keeps for its captured frames: per-tile list buckets sized from the headers of
# earlier frames (no scan, no placement pass), the tile launch order of the previous frame on the same buffers (no tile_scan launch),
# and counters that the previous BACKWARD on the context cleared on its way out (no zero-fill launch, the per-Gaussian preprocess at
# the head of the binning kernel): DqoRastCtx.tile_bucket_capacity / keep_tile_order / frame_prezeroed, include/dqo_raster.h.
# Who may reuse a context: the buffers are the autograd context of a call whose graph may still be alive (retain_graph=True, outputs
# held by the caller), so every forward takes a LEASE that lives in its autograd ctx object and ends with it; a context is handed out
# again only when its lease is gone.  A loop that holds the previous iteration's outputs across the next forward (DQO-MAP's does)
# simply alternates between two contexts.
_pool = {}             # (device index, stream handle, P, W, H) -> [_CtxSet]
_pool_on = True        # set_context_pool
_POOL_SETS = 3         # contexts per key; a forward that finds all of them leased takes the unpooled path
_POOL_KEYS = 4         # shapes with pooled contexts, most recently used last: a map that grows changes P with every keyframe, and the
                       # contexts of the sizes it has left behind are dropped (one still leased lives on with its graph)
_shape_hint = {}       # (device index, P, W, H) -> [candidate pairs, longest tile list] of earlier frames (max over them)


class _CtxSet:
    __slots__ = ("geom", "img", "binning", "cap", "bucket", "order_valid", "clean", "leased",
                 "captured")  # captured: (keep_tile_order, frame_prezeroed) of the captured forward that pinned it ('graph' mode)


class _Lease:
    """Held by the autograd ctx of the forward that uses a pooled context; the context is free again when the ctx object dies (the
    graph was freed, or the forward ran without grad)."""
    __slots__ = ("set",)

    def __init__(self, s):
        self.set = s
        s.leased = True

    def __del__(self):
        self.set.leased = False


class _Pinned:
    """'graph' mode: the context a CAPTURED forward runs on belongs to the graph for good — its buffers are baked into the captured
    launches — so it leaves the pool and is never handed out again.  It is kept alive by `_captured` until the owner of the graph takes
    it over (take_captured_contexts)."""
    __slots__ = ("set",)

    def __init__(self, s):
        self.set = s
        s.leased = True


_captured = []         # contexts pinned by captured forwards whose owner has not called take_captured_contexts() yet


def take_captured_contexts():
    """The contexts that forwards captured in 'graph' mode have pinned since the last call.  Whoever owns the torch.cuda.CUDAGraph keeps
    the returned list for as long as the graph may be replayed and drops it with the graph (fused_ops.CapturedIteration does); until
    somebody takes them the module keeps them alive — a caller that never calls this leaks one context per captured forward
    instead of replaying a graph on freed memory."""
    with _lock:
        out = list(_captured)
        del _captured[:]
    return out


def set_context_pool(on):
    """True (default): forwards in 'lazy' / 'deferred' mode run on pooled contexts once the shape's list statistics are known (from
    the second call of a shape on).  False: every forward allocates its context anew, like the reference.  Results are the same bits
    either way (the lists, their order and every kernel that reads them are the same); switching it off also frees the pool."""
    global _pool_on
    with _lock:
        _pool_on = bool(on)
        if not _pool_on:
            _pool.clear()


def _bucket_for(longest):
    """Per-tile bucket for lists up to `longest` entries in earlier frames: a power of two, at least twice that (FusedMapper.capture's
    rule, and its exception: 1024 — what the per-tile sort reaches, so the long-list sort launch disappears — if 1.3 x longest fits)."""
    b = 256
    while b < 2 * longest:
        b *= 2
    if b == 2048 and 1.3 * longest <= 1024:
        b = 1024
    return b


def _pooled_set(lib, key, stream, dev, W, H, pin=False):
    """A free pooled context for this shape with room for the shape's hints, or None (pool off, statistics not known yet, every
    context leased).  Called under the module lock.
    pin ('graph' mode, a forward that is being captured): the context leaves the pool for good (_Pinned); a free one of ANY stream will
    do — the warm-up iterations ran on a side stream and are complete (a capture starts from a synchronised device) — and is the one to
    have: it holds a tile order and the counters the warm-up's last backward cleared, so the captured launches are the short sequence."""
    if not _pool_on:
        return None
    hint = _shape_hint.get(key)
    if hint is None or hint[1] <= 0:
        return None
    cap, bucket = int(hint[0] * 1.25) + 4096, _bucket_for(hint[1])
    if pin:
        for pkey, sets in _pool.items():
            if (pkey[0],) + pkey[2:] != key:
                continue
            for cs in sets:
                if not cs.leased and cs.cap >= cap and cs.bucket >= bucket:
                    sets.remove(cs)
                    return cs
    pkey = (key[0], stream) + key[1:]
    sets = _pool.pop(pkey, [])
    _pool[pkey] = sets  # (dicts keep insertion order: the shape just used is the last one)
    while len(_pool) > _POOL_KEYS:
        del _pool[next(iter(_pool))]
    for cs in sets:
        if not cs.leased and cs.cap >= cap and cs.bucket >= bucket:
            return cs
    sets[:] = [cs for cs in sets if cs.leased or (cs.cap >= cap and cs.bucket >= bucket)]  # (outgrown free contexts go)
    if len(sets) >= _POOL_SETS:
        return None
    u8 = dict(dtype=torch.uint8, device=dev)
    cs = _CtxSet()
    cs.cap, cs.bucket = int(cap * 1.2), bucket  # (headroom: a context is replaced when the hints outgrow it)
    cs.geom = torch.empty((lib.dqo_rast_geom_bytes(key[1], W, H),), **u8)
    cs.img = torch.empty((lib.dqo_rast_image_bytes(W, H),), **u8)
    cs.binning = torch.empty((lib.dqo_rast_binning_bytes_bucketed(cs.cap, W, H, cs.bucket),), **u8)
    cs.order_valid = cs.clean = cs.leased = False
    if not pin:
        sets.append(cs)
    return cs


def _ring_slot(dev_index):
    """(pinned 8-int32 host buffer, event, host pointer, raw event handle) from the device's ring (allocated on first use; the event is
    recorded once at creation — on the device it will serve — so that its handle exists: dqo_rast_forward_async records it from C).
    A slot is reused after _RING later forwards; a pending lazy check older than that has long been verified (lazy mode verifies at
    every forward).  Called under the module lock."""
    ring, pos = _ring.setdefault(dev_index, ([], [0]))
    if len(ring) < _RING:
        host, ev = torch.empty((8,), dtype=torch.int32).pin_memory(), torch.cuda.Event()
        host.zero_()
        ev.record()
        ring.append((host, ev, host.data_ptr(), ev.cuda_event))
        return ring[-1]
    pos[0] = (pos[0] + 1) % _RING
    slot = ring[pos[0]]
    if any(p[1] is slot[0] for p in _pending):  # (never in practice: the GPU would be _RING frames behind)
        _verify_pending(block=True)
    return slot


def last_header():
    """Device header of the most recent forward (waits for its asynchronous 32-byte copy): dict with num_rendered = instances kept
    in the tile lists, num_candidates = the reference's num_rendered, num_tiles, max_tile_count, num_visible, overflow."""
    hd = _last["header"]
    if hd is None:
        return None
    if torch.is_tensor(hd):  # exact mode: read on demand from the forward's geometry buffer
        h = hd[:32].view(torch.int32).cpu().tolist()
        return dict(num_rendered=h[0], num_tiles=h[1], overflow=h[2], max_tile_count=h[3], num_visible=h[4], num_candidates=h[5])
    ev, host = hd
    ev.synchronize()
    h = host.tolist()
    return dict(num_rendered=h[0], num_tiles=h[1], overflow=h[2], max_tile_count=h[3], num_visible=h[4], num_candidates=h[5])


def last_num_rendered():
    """Instance count N of the most recent forward that ran in 'exact' mode (statistics for benchmarks)."""
    return _last["num_rendered"]


def set_list_split(longer_than):
    """0 (default): the reference's front-to-back order of arithmetic in every tile list.  n > 0: the forward shares every list longer
    than n entries between eight waves (DqoRastCtx.list_split, include/dqo_raster.h) — for renders of a few hundred tiles, a
    strong-scaling shard, which cannot fill the GPU; results on those lists differ from the serial order in the last bits."""
    global _list_split
    if int(longer_than) < 0:
        raise ValueError("list_split is 0 (off) or a list length")
    _list_split = int(longer_than)


def set_sync_mode(mode):
    """'exact' (default): one 4-byte D2H read per forward.  'lazy': no host synchronisation in the forward — the instance capacity is
    carried over from earlier calls, the device header of every forward is copied back asynchronously and checked later; the BACKWARD
    waits for its own forward's header first (the host then idles until the forward has run: 0.3 ms per iteration on config 3), so a
    frame that outgrew the capacity raises before any gradient of it exists.  'deferred': like 'lazy' without that wait — the host
    never blocks; an overflowed frame (empty lists: its outputs are background, its gradients exact zeros, nothing is written out of
    bounds) raises at a later forward / backward / verify_pending() call, typically one iteration later, and the optimiser steps taken
    in between saw zero gradients.  For loops that can tolerate that (or call verify_pending() where it matters).
    'graph': for callers that capture their iteration with torch.cuda.graph (the op, the caller's loss, autograd and a capturable
    optimiser in ONE graph replay — on config 3 the host time of the ~130 eager launches of DQO-MAP's loop is what bounds it): nothing
    the op does touches the host — no header copy, no event, no check; the capacity is the one earlier calls measured (run at least one
    iteration in 'lazy' mode first — the usual warm-up before a capture — or call set_capacity(P, W, H, instances)), and the frame's
    header stays on the device: last_header() (after the replay) reads it; `overflow` there means the frame was invalid (background
    outputs, zero gradients) and the graph must be captured again with a larger capacity.  The gated op's object ids are not range-checked
    in this mode (the check reads scalars back): run one gated call in another mode first, as the warm-up does."""
    global _sync_mode
    if mode not in ("exact", "lazy", "deferred", "graph"):
        raise ValueError(mode)
    if _pending:
        _verify_pending(block=True)  # forwards issued in lazy mode are still checked (raises if one of them overflowed)
    _sync_mode = mode
    if mode == "exact":
        with _lock:
            _pool.clear()  # (only the other modes use pooled contexts; one still leased lives on with its graph)


def set_capacity(P, W, H, instances, device_index=None):
    """Instance capacity (Gaussian-tile pairs) of later forwards of this shape in 'lazy' / 'deferred' / 'graph' mode; it only grows."""
    dev_index = torch.cuda.current_device() if device_index is None else int(device_index)
    key = (dev_index, int(P), int(W), int(H))
    with _lock:
        _cap_hint[key] = max(_cap_hint.get(key, 0), int(instances))


def verify_pending():
    """Wait for the headers of every forward issued in 'lazy' / 'deferred' mode and raise if one of them overflowed its capacity."""
    _verify_pending(block=True)


def _verify_pending(block):
    """Lazy mode: check the headers of earlier forwards; raise if one of them overflowed its instance capacity.  The whole pass runs
    under the module lock (forwards of other threads append to `_pending` and read `_cap_hint`; the autograd engine calls this from
    its own thread); the error is raised after the lock is released."""
    err = None
    with _lock:
        keep = []
        for i, (ev, host, key, cap, pooled) in enumerate(_pending):
            if not block and not ev.query():
                keep.append((ev, host, key, cap, pooled))
                continue
            ev.synchronize()
            n, overflow = int(host[0]), int(host[2])
            # the hint only grows: one key serves calls with different tile masks / camera poses, whose N differ
            _cap_hint[key] = max(_cap_hint.get(key, 0), int(n * 1.25) + 4096)
            sh = _shape_hint.setdefault(key, [0, 0])  # (the context pool's sizing: candidate pairs and the longest list)
            sh[0], sh[1] = max(sh[0], int(host[5])), max(sh[1], int(host[3]))
            if overflow and pooled:
                # a region of the map may have outgrown its share of the instance slots although the bucket and the total fitted
                # (include/dqo_raster.h, tile_bucket_capacity) -> more instance slots next time (cap = 1.25 x this + 4096)
                sh[0] = max(sh[0], int(cap * 1.2))
            if overflow:
                keep.extend(_pending[i + 1:])  # (the later forwards stay pending: they are checked by the next call)
                err = (n, cap, pooled)
                break
        _pending[:] = keep
    if err is not None and err[2]:
        raise RuntimeError(f"diff_gaussian_rasterization_depth (lazy mode, pooled context): a previous forward produced {err[0]} "
                           f"Gaussian-tile instances or a tile list that did not fit its context ({err[1]} instance slots in per-tile "
                           "buckets); its outputs are invalid. The sizes have been raised — re-run that iteration (or use "
                           "set_sync_mode('exact')).")
    if err is not None:
        raise RuntimeError(f"diff_gaussian_rasterization_depth (lazy mode): a previous forward produced {err[0]} Gaussian-tile "
                           f"instances but only {err[1]} fitted its binning buffer; its outputs are invalid. The capacity has "
                           "been raised — re-run that iteration (or use set_sync_mode('exact')).")


_gate_checked = {}  # id(tensor) -> (weak reference, version) of object-id tensors whose value range has been checked


def _check_gate_ids(gaussian_object, pixel_object):
    """The object gate's ids must lie in [0, 64) (pixel ids: < 64, negative = no owner): the kernels keep a quadrant's owners as a
    64-bit set and skip the exact id comparison where a quadrant has one owner, so an id of 69 would act on the pixels of object 5
    (DqoObjectGate, include/dqo_raster.h).  Checked once per tensor OBJECT (and again when it is modified in place): the cache is
    keyed on a weak reference, so a new tensor that the caching allocator places at a freed tensor's address is checked again.  The
    check reads two scalars back (a host synchronisation): the 'graph' mode, whose contract is no host touch, skips it — its caller
    validates the ids once before capturing (set_sync_mode's docstring)."""
    if _sync_mode == "graph":
        return
    for t, lo_ok in ((gaussian_object, False), (pixel_object, True)):
        hit = _gate_checked.get(id(t))
        if hit is not None and hit[0]() is t and hit[1] == t._version:
            continue
        if t.numel():
            mn, mx = int(t.min().item()), int(t.max().item())
            if mx > 63 or (mn < 0 and not lo_ok):
                raise RuntimeError("object gate: object ids must lie in [0, 64) (negative pixel ids = no owner); got "
                                   f"[{mn}, {mx}] in {'pixel_object' if lo_ok else 'gaussian_object'}")
        key = id(t)
        _gate_checked[key] = (weakref.ref(t, lambda _r, k=key: _gate_checked.pop(k, None)), t._version)


def gate_ids_checked(t):
    """The caller vouches for an object-id tensor it has just (re)written from values that were checked before (e.g. a subset of a
    checked pixel_object with -1 elsewhere): _check_gate_ids will not read it back — no host synchronisation — until it is modified
    again.  A wrong promise has the consequence _check_gate_ids describes."""
    key = id(t)
    _gate_checked[key] = (weakref.ref(t, lambda _r, k=key: _gate_checked.pop(k, None)), t._version)


def _f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {t.dtype} ({name})")
    return t.contiguous()


def rasterize_gaussians(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask, raster_settings):
    return _RasterizeGaussians.apply(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask,
                                     raster_settings)


def rasterize_gaussians_gated(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask, raster_settings,
                              gaussian_object, pixel_object, tile_objects=None):
    """NOT part of the reference's surface: the same op with libdqoraster's object gate (include/dqo_raster.h, DqoObjectGate) — a list
    entry acts on a pixel only if gaussian_object[id] == pixel_object[pixel] (int32 [P] / [H, W]; a negative pixel id: nothing
    acts), in the forward and in the backward.  The per-object render of the sharded mapping job: every pixel sees its
    own object alone, so a shard that holds some of the objects computes exactly its pixels of the unsharded render.
    tile_objects (optional, DqoObjectGate.tile_objects): int64 [tiles], per 16x16 tile the 64-bit set of the objects that own one of
    its pixels — the binning then drops a (Gaussian, tile) pair whose object owns no pixel of the tile (it could act on none); must
    be the sets of THIS pixel_object (dqo_harness.fused_mapping.tile_object_sets)."""
    return _RasterizeGaussians.apply(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask,
                                     raster_settings, gaussian_object, pixel_object, tile_objects)


def _params(rs, P, M):
    return N.DqoRastParams(P=P, D=int(rs.sh_degree), M=M, W=int(rs.image_width), H=int(rs.image_height),
                           prefiltered=int(bool(rs.prefiltered)), debug=int(bool(rs.debug)), tanfovx=float(rs.tanfovx),
                           tanfovy=float(rs.tanfovy), cx=float(rs.cx), cy=float(rs.cy), scale_modifier=float(rs.scale_modifier),
                           color_sigma=float(rs.color_sigma), opaque_threshold=float(rs.opaque_threshold),
                           depth_threshold=float(rs.depth_threshold), normal_threshold=float(rs.normal_threshold),
                           T_threshold=float(rs.T_threshold))


def _inputs(rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask, row_flags=None):
    """row_flags: DqoRastInputs.row_flags (uint8 [P], DQO_ROW_HIDDEN rows are not rendered) — None for the drop-in operator, whose
    reference counterpart renders every row it is given."""
    return N.DqoRastInputs(bg=N.ptr(rs.bg), means3D=N.ptr(means3D), shs=N.ptr(sh), colors_precomp=N.ptr(colors_precomp),
                           opacities=N.ptr(opacities), scales=N.ptr(scales), rotations=N.ptr(rotations),
                           cov3D_precomp=N.ptr(cov3Ds_precomp), viewmatrix=N.ptr(rs.viewmatrix), projmatrix=N.ptr(rs.projmatrix),
                           campos=N.ptr(rs.campos), tile_mask=N.ptr(tile_mask), row_flags=N.ptr(row_flags))


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask, raster_settings,
                gaussian_object=None, pixel_object=None, tile_objects=None):
        rs = raster_settings
        lib = N.lib()
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:67-70
        N.require_gpu(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask, rs.bg, rs.viewmatrix,
                      rs.projmatrix, rs.campos)
        if not means3D.is_cuda:
            raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
        dev = means3D.device
        means3D, opacities = _f32(means3D, "means3D"), _f32(opacities, "opacities")
        sh, colors_precomp = _f32(sh, "sh"), _f32(colors_precomp, "colors_precomp")
        scales, rotations, cov3Ds_precomp = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(cov3Ds_precomp, "cov3D_precomp")
        bg, view, proj, campos = (_f32(rs.bg, "bg"), _f32(rs.viewmatrix, "viewmatrix"), _f32(rs.projmatrix, "projmatrix"),
                                  _f32(rs.campos, "campos"))
        rs = rs._replace(bg=bg, viewmatrix=view, projmatrix=proj, campos=campos)
        if tile_mask is not None:
            if tile_mask.dtype != torch.int32:
                raise RuntimeError(f"expected scalar type Int but found {tile_mask.dtype} (tile_mask)")
            tile_mask = tile_mask.contiguous()
        P = means3D.size(0)
        H, W = int(rs.image_height), int(rs.image_width)
        M = sh.size(1) if sh.numel() != 0 else 0  # rasterize_points.cu:105-109
        gate = None
        if (gaussian_object is None) != (pixel_object is None):
            raise RuntimeError("object gate: gaussian_object and pixel_object go together")
        if gaussian_object is not None:
            N.require_gpu(gaussian_object, pixel_object)
            if gaussian_object.dtype != torch.int32 or pixel_object.dtype != torch.int32:
                raise RuntimeError("expected scalar type Int (object gate)")
            if gaussian_object.numel() != P or pixel_object.numel() != H * W:
                raise RuntimeError("object gate: gaussian_object must have num_points elements, pixel_object H x W")
            gaussian_object, pixel_object = gaussian_object.contiguous(), pixel_object.contiguous()
            _check_gate_ids(gaussian_object, pixel_object)
            gate = N.DqoObjectGate(gaussian_object=N.ptr(gaussian_object), pixel_object=N.ptr(pixel_object))
            if tile_objects is not None:
                N.require_gpu(tile_objects)
                if tile_objects.dtype != torch.int64 or tile_objects.numel() != ((H + 15) // 16) * ((W + 15) // 16):
                    raise RuntimeError("object gate: tile_objects must be int64 with ceil(H/16) x ceil(W/16) elements")
                tile_objects = tile_objects.contiguous()
                gate.tile_objects = N.ptr(tile_objects)
        elif tile_objects is not None:
            raise RuntimeError("object gate: tile_objects without gaussian_object / pixel_object")
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            stream = N.current_stream()
            color = torch.empty((3, H, W), **f32)
            depth = torch.empty((1, H, W), **f32)
            hit_color = torch.empty((1, H, W), **i32)
            hit_depth = torch.empty((1, H, W), **i32)
            hit_color_weight = torch.empty((1, H, W), **f32)
            hit_depth_weight = torch.empty((1, H, W), **f32)
            T_map = torch.empty((1, H, W), **f32)
            n_touched = torch.empty((P,), **i32)
            radii = torch.empty((P,), **i32)
            geomBuffer = torch.empty((lib.dqo_rast_geom_bytes(P, W, H),), **u8)
            imgBuffer = torch.empty((lib.dqo_rast_image_bytes(W, H),), **u8)
            params = _params(rs, P, M)
            inputs = _inputs(rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask)
            outputs = N.DqoRastOutputs(out_color=color.data_ptr(), out_depth=depth.data_ptr(), out_hit_color=hit_color.data_ptr(),
                                       out_hit_depth=hit_depth.data_ptr(), out_hit_color_weight=hit_color_weight.data_ptr(),
                                       out_hit_depth_weight=hit_depth_weight.data_ptr(), out_T=T_map.data_ptr(),
                                       n_touched=N.ptr(n_touched), radii=N.ptr(radii))
            cctx = N.DqoRastCtx(geom=geomBuffer.data_ptr(), geom_bytes=geomBuffer.numel(), binning=None, binning_bytes=0,
                                image=imgBuffer.data_ptr(), image_bytes=imgBuffer.numel(), inst_capacity=0, list_split=_list_split)
            if gate is not None:
                cctx.object_gate = ctypes.addressof(gate)
            key = (dev.index, P, W, H)
            cap = None
            lease = None
            if _sync_mode == "graph":
                with _lock:
                    cap = _cap_hint.get(key)
                if cap is None:
                    raise RuntimeError("diff_gaussian_rasterization_depth ('graph' mode): no instance capacity known for "
                                       f"P={P}, {W}x{H}: run one forward in 'lazy' mode first (the warm-up before the capture) or call "
                                       "set_capacity(P, W, H, instances)")
                # nothing below touches the host: the launches of both stages, no header copy, no event
                num_rendered = -1
                with _lock:
                    # (only a forward that IS being captured pins a context: an eager call in this mode keeps contexts of its own)
                    capturing = P > 0 and torch.cuda.is_current_stream_capturing()
                    pooled = _pooled_set(lib, key, stream, dev, W, H, pin=True) if capturing else None
                    if pooled is not None:
                        # the context belongs to the graph from here on (_Pinned).  The counters' promise holds at every replay only if
                        # the captured iteration also holds this frame's backward, which clears them: a forward that needs no gradient
                        # is captured with its own zero fill
                        lease = _Pinned(pooled)
                        _captured.append(pooled)
                        geomBuffer, imgBuffer, binningBuffer, cap = pooled.geom, pooled.img, pooled.binning, pooled.cap
                        cctx.geom, cctx.geom_bytes = geomBuffer.data_ptr(), geomBuffer.numel()
                        cctx.image, cctx.image_bytes = imgBuffer.data_ptr(), imgBuffer.numel()
                        cctx.tile_bucket_capacity = pooled.bucket
                        cctx.keep_tile_order = 1 if pooled.order_valid else 0
                        cctx.frame_prezeroed = 1 if (pooled.clean and _list_split == 0 and any(ctx.needs_input_grad)) else 0
                        pooled.clean = False
                        pooled.captured = (int(cctx.keep_tile_order), int(cctx.frame_prezeroed))
                    else:
                        binningBuffer = torch.empty((lib.dqo_rast_binning_bytes(cap),), **u8)
                cctx.binning, cctx.binning_bytes, cctx.inst_capacity = binningBuffer.data_ptr(), binningBuffer.numel(), cap
                N.check(lib.dqo_rast_forward_async(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(outputs),
                                                   ctypes.byref(cctx), None, None, stream))
                _last["header"] = geomBuffer  # (a captured forward: the graph's own buffer, valid after every replay)
            elif _sync_mode != "exact":
                _verify_pending(block=False)
                with _lock:
                    cap = _cap_hint.get(key)
            if _sync_mode == "graph":
                pass
            elif cap is None:
                # 'exact', or the first call for this shape in 'lazy' / 'deferred' (measure once): stage 1, the one D2H read, stage 2
                N.check(lib.dqo_rast_forward_prepare(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(outputs),
                                                     ctypes.byref(cctx), stream))
                hdr = N.DqoRastHeader()
                N.check(lib.dqo_rast_read_header(ctypes.byref(cctx), ctypes.byref(hdr), stream))
                # (Gaussian, tile) pairs in the tile rects: the reference's num_rendered, and an upper bound of the instances
                # the binning keeps after its footprint test — a capacity that always fits
                num_rendered = int(hdr.num_candidates)
                cap = max(num_rendered, 1)
                if _sync_mode == "exact":
                    _last["num_rendered"], _last["num_visible"] = num_rendered, int(hdr.num_visible)
                else:
                    cap = num_rendered + 4096  # later calls keep max(this, 1.25 N)
                    with _lock:
                        cap = _cap_hint[key] = max(_cap_hint.get(key, 0), cap)
                    num_rendered = -1
                binningBuffer = torch.empty((lib.dqo_rast_binning_bytes(cap),), **u8)
                cctx.binning, cctx.binning_bytes, cctx.inst_capacity = binningBuffer.data_ptr(), binningBuffer.numel(), cap
                N.check(lib.dqo_rast_forward_render(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(outputs),
                                                    ctypes.byref(cctx), stream))
                if _sync_mode == "exact":
                    # exact mode has read what it needs already: nothing per call; last_header() reads the rest on demand from the
                    # geometry buffer, which therefore stays referenced until the next forward (~160 B per Gaussian, one call longer)
                    _last["header"] = geomBuffer
                else:
                    with _lock:
                        host, ev, _, _ = _ring_slot(dev.index)
                        host.copy_(geomBuffer[:32].view(torch.int32), non_blocking=True)
                        ev.record()
                        _pending.append((ev, host, key, cap, False))
                        _last["header"] = (ev, host)
            else:
                # 'lazy' / 'deferred' with a carried-over capacity: ONE call — both stages, and between the sort and the blend kernel
                # (where the frame's header is final) its asynchronous copy into a pinned ring slot + the slot's event: the deferred
                # capacity check, available a blend kernel before the forward's end
                num_rendered = -1
                with _lock:
                    pooled = _pooled_set(lib, key, stream, dev, W, H) if P > 0 else None
                    if pooled is not None:
                        # a pooled context (see _pool): its buffers instead of the fresh ones, lists in per-tile buckets, the tile
                        # order of the previous frame on it, and — if that frame's backward cleared the counters — no zero fill and
                        # no preprocess launch
                        lease = _Lease(pooled)
                        geomBuffer, imgBuffer, binningBuffer, cap = pooled.geom, pooled.img, pooled.binning, pooled.cap
                        cctx.geom, cctx.geom_bytes = geomBuffer.data_ptr(), geomBuffer.numel()
                        cctx.image, cctx.image_bytes = imgBuffer.data_ptr(), imgBuffer.numel()
                        cctx.tile_bucket_capacity = pooled.bucket
                        cctx.keep_tile_order = 1 if pooled.order_valid else 0
                        cctx.frame_prezeroed = 1 if (pooled.clean and _list_split == 0) else 0
                        pooled.clean = False  # (whatever happens below: this frame's counters are in use)
                    else:
                        binningBuffer = torch.empty((lib.dqo_rast_binning_bytes(cap),), **u8)
                    cctx.binning, cctx.binning_bytes, cctx.inst_capacity = binningBuffer.data_ptr(), binningBuffer.numel(), cap
                    host, ev, host_ptr, ev_handle = _ring_slot(dev.index)
                    N.check(lib.dqo_rast_forward_async(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(outputs),
                                                       ctypes.byref(cctx), host_ptr, ev_handle, stream))
                    if pooled is not None:
                        pooled.order_valid = True
                    _pending.append((ev, host, key, cap, pooled is not None))
                    _last["header"] = (ev, host)
        # Only grad_out_color and grad_out_depth are consumed by the backward (like the reference's, __init__.py:176-238): the engine need not
        # fill zero images for the three float outputs nobody differentiated through (hit_color_weight, hit_depth_weight, T_map) — three
        # fill launches per backward; an undefined colour / depth gradient arrives as None and is replaced by zeros there
        if hasattr(ctx, "set_materialize_grads"):  # (callers that drive forward / backward by hand pass a plain object)
            ctx.set_materialize_grads(False)
        ctx.pooled = lease  # (None: a context of this call's own)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.inst_capacity = cap
        ctx.list_split = _list_split  # (the backward shares the same lists between eight waves: the queue is the forward's)
        ctx.M = M
        ctx.object_gate = None if gate is None else (gaussian_object, pixel_object)
        ctx.save_for_backward(colors_precomp, hit_depth, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, opacities, tile_mask if tile_mask is not None else torch.empty(0))
        ctx.mark_non_differentiable(hit_color, hit_depth, n_touched, radii)
        return (color, depth, hit_color, hit_depth, hit_color_weight, hit_depth_weight, T_map, n_touched, radii)

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, grad_hit_color, grad_hit_depth, grad_hit_color_weight,
                 grad_hit_depth_weight, grad_T_map, grad_n_touched, _):
        # only grad_out_color and grad_out_depth are consumed, exactly like the reference (__init__.py:176-238; F6)
        rs = ctx.raster_settings
        lib = N.lib()
        (colors_precomp, hit_depth, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer,
         opacities, tile_mask) = ctx.saved_tensors
        if tile_mask.numel() == 0:
            tile_mask = None
        if _sync_mode in ("lazy", "deferred"):
            _verify_pending(block=(_sync_mode == "lazy"))  # (lazy: no gradient of an invalid frame; deferred: never wait, see set_sync_mode)
        P, M = means3D.size(0), ctx.M
        H, W = int(rs.image_height), int(rs.image_width)
        dev = means3D.device
        f32 = dict(dtype=torch.float32, device=dev)
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, H, W), **f32)
        if grad_out_depth is None:
            grad_out_depth = torch.zeros((1, H, W), **f32)
        grad_out_color, grad_out_depth = _f32(grad_out_color, "grad_out_color"), _f32(grad_out_depth, "grad_out_depth")
        with torch.cuda.device(dev):
            stream = N.current_stream()
            # dL/dmeans2D is computed by the reference's C++ and dropped by its Python (__init__.py:247-283 returns no slot for it);
            # dL/dcolors_precomp and dL/dcov3D_precomp only have a receiver when those inputs were given: rows nobody reads are not
            # written (NULL = the per-Gaussian kernel skips the tensor: 24 of its 142 MB of stores on config 3)
            want_colors, want_cov = colors_precomp.numel() != 0, cov3Ds_precomp.numel() != 0
            g_means3D = torch.empty((P, 3), **f32)
            g_colors = torch.empty((P, 3), **f32) if want_colors else None
            g_opacity = torch.empty((P, 1), **f32)
            g_cov3D = torch.empty((P, 6), **f32) if want_cov else None
            g_sh = torch.empty((P, M, 3), **f32)
            g_scales = torch.empty((P, 3), **f32)
            g_rot = torch.empty((P, 4), **f32)
            if P > 0:
                cap = ctx.inst_capacity
                ws = torch.empty((lib.dqo_rast_backward_workspace_bytes(cap),), dtype=torch.uint8, device=dev)
                params = _params(rs, P, M)
                inputs = _inputs(rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, tile_mask)
                cctx = N.DqoRastCtx(geom=geomBuffer.data_ptr(), geom_bytes=geomBuffer.numel(), binning=binningBuffer.data_ptr(),
                                    binning_bytes=binningBuffer.numel(), image=imgBuffer.data_ptr(), image_bytes=imgBuffer.numel(),
                                    inst_capacity=cap, list_split=getattr(ctx, "list_split", 0))
                lease = getattr(ctx, "pooled", None)
                if lease is not None:
                    # a pooled context: its lists live in buckets, and this call — the last consumer of the frame's counters — clears
                    # them for the next forward on the context (DqoRastCtx.frame_prezeroed as dqo_rast_backward reads it)
                    cctx.tile_bucket_capacity = lease.set.bucket
                    cctx.frame_prezeroed = 1 if cctx.list_split == 0 else 0
                if getattr(ctx, "object_gate", None) is not None:
                    gate = N.DqoObjectGate(gaussian_object=N.ptr(ctx.object_gate[0]), pixel_object=N.ptr(ctx.object_gate[1]))
                    cctx.object_gate = ctypes.addressof(gate)
                grads = N.DqoRastGrads(dL_dmeans3D=g_means3D.data_ptr(), dL_dsh=N.ptr(g_sh), dL_dcolors=N.ptr(g_colors),
                                       dL_dopacity=g_opacity.data_ptr(), dL_dscales=g_scales.data_ptr(),
                                       dL_drotations=g_rot.data_ptr(), dL_dcov3D=N.ptr(g_cov3D), dL_dmeans2D=None,
                                       skip_culled_rows=1 if getattr(ctx, "sparse_grad_rows", False) else 0)
                N.check(lib.dqo_rast_backward(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(cctx),
                                              grad_out_color.data_ptr(), grad_out_depth.data_ptr(), hit_depth.data_ptr(),
                                              ctypes.byref(grads), ws.data_ptr(), ws.numel(), stream))
                if lease is not None and cctx.frame_prezeroed:
                    with _lock:
                        lease.set.clean = True
        # gradient slots of the reference (__init__.py:273-283); inputs handed over as empty tensors get None
        def slot(g, inp):
            return g if (g is not None and inp.numel() != 0) else None
        return (g_means3D, slot(g_sh, sh), slot(g_colors, colors_precomp), g_opacity.view_as(opacities) if opacities.numel() else None,
                slot(g_scales, scales), slot(g_rot, rotations), slot(g_cov3D, cov3Ds_precomp), None, None, None, None, None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    opaque_threshold: float
    normal_threshold: float
    depth_threshold: float
    prefiltered: bool
    debug: bool
    cx: float
    cy: float
    color_sigma: float = 3.0
    T_threshold: float = 0.0001


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # frustum test of rasterizer_impl.cu:54-66 for the camera in raster_settings
        with torch.no_grad():
            rs = self.raster_settings
            N.require_gpu(positions, rs.viewmatrix, rs.projmatrix)
            if not positions.is_cuda:
                raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
            positions = _f32(positions, "positions")
            P = positions.size(0)
            visible = torch.zeros((P,), dtype=torch.bool, device=positions.device)
            with torch.cuda.device(positions.device):
                N.check(N.lib().dqo_mark_visible(P, N.ptr(positions), N.ptr(_f32(rs.viewmatrix, "viewmatrix")),
                                                 N.ptr(_f32(rs.projmatrix, "projmatrix")), N.ptr(visible), N.current_stream()))
        return visible

    def forward(self, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                tile_mask=None, normal_w=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        # normal_w is accepted and ignored, as in the reference (__init__.py:335)
        return rasterize_gaussians(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, tile_mask,
                                   raster_settings)
