"""Per-object sharding of the mapping hot path across GPUs (new design, the reference is single-GPU).

Every Gaussian carries an integer object id (SLAM/gaussian_pointcloud.py:497 `_obj_id`) and quadric fits are independent per
object (SLAM/multiprocess/quadrics.py:2245-2295).  With the loss evaluated per object on that object's screen mask
(disjoint instance masks => L = sum_k L_k(G_k)) the path partitions by object id with NO data-path collective:
shard s owns the objects assigned to it, holds only their Gaussians + Adam state and renders only the tiles its objects'
masks touch.  The single exchange per iteration is one packed fp32 all-reduce (RCCL over xGMI on MI355X; gloo in the CPU
tests) of the quantities shared by all shards: the loss scalars for logging and, if a caller optimises shared parameters
(camera twist / exposure — not present in the reference, F2), their gradients.  Payload <= 1 KB => latency bound, one
collective per iteration, never one per tensor.
"""
import numpy as np


def assign_objects(obj_sizes, world):
    """Longest-processing-time greedy: objects sorted by size, each to the currently lightest shard.
    obj_sizes: {object id: #Gaussians or any other cost}.  Returns {object id: shard}."""
    load = [0] * world
    out = {}
    for k, n in sorted(obj_sizes.items(), key=lambda kv: (-kv[1], kv[0])):
        s = int(np.argmin(load))
        out[k] = s
        load[s] += n
    return out


def shard_indices(obj_id, assignment, rank):
    """Indices of the Gaussians owned by `rank`."""
    owner = np.vectorize(lambda k: assignment[int(k)])(obj_id) if len(obj_id) else np.zeros(0, int)
    return np.nonzero(owner == rank)[0]


def shard_scene(scene, rank, world, work=None):
    """(the rank's Gaussians, {object id: shard}).  Objects are balanced over the shards by their Gaussian counts, or — `work`, one
    non-negative number per Gaussian, e.g. the tiles it covers in the current view (view_work) — by the sum of `work` over the
    object's Gaussians: a shard's time follows the instances its objects put on screen, not how many Gaussians it stores."""
    ids, inv, counts = np.unique(scene["obj_id"], return_inverse=True, return_counts=True)
    if work is None:
        sizes = {int(k): int(c) for k, c in zip(ids, counts)}
    else:
        w = np.bincount(inv, weights=np.asarray(work, np.float64), minlength=len(ids))
        # (the Gaussian count only breaks ties, e.g. between objects that are out of view)
        sizes = {int(k): float(w[j]) + 1e-3 * int(c) for j, (k, c) in enumerate(zip(ids, counts))}
    assignment = assign_objects(sizes, world)
    keep = shard_indices(scene["obj_id"], assignment, rank)
    return {k: v[keep] for k, v in scene.items()}, assignment


def view_work(radii):
    """Per-Gaussian work estimate in one view from the op's `radii` output: the tiles inside the bounding square of radius r that
    the reference lists the Gaussian in (forward.cu:344-353) — 0 for culled Gaussians."""
    r = np.asarray(radii, np.float64)
    side = np.floor(2.0 * r / 16.0) + 2.0
    return np.where(r > 0, side * side, 0.0)


def tile_mask_from_pixel_mask(pixel_mask):
    """16-px tile mask (int32 [ceil(H/16), ceil(W/16)]) covering every True pixel: the shard renders only these tiles."""
    H, W = pixel_mask.shape
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = np.zeros((gy * 16, gx * 16), bool)
    pad[:H, :W] = pixel_mask
    return pad.reshape(gy, 16, gx, 16).any(axis=(1, 3)).astype(np.int32)


class PackedAllReduce:
    """One all-reduce per iteration over a packed fp32 buffer of named shared quantities."""

    def __init__(self, spec, device, group=None, force=False):
        """force: issue the collective on a ONE-rank group too (a sum over one rank: the values come back unchanged): the N-rank code
        path, with its staging ring and stream semantics, on the backend the group was created with."""
        import torch
        self.force = bool(force)
        self.spec = [(name, int(n)) for name, n in spec]
        self.offsets, off = {}, 0
        for name, n in self.spec:
            self.offsets[name] = (off, n)
            off += n
        self.buf = torch.zeros(off, dtype=torch.float32, device=device)
        self.group = group

    def put(self, name, value):
        off, n = self.offsets[name]
        self.buf[off:off + n] = value.reshape(-1) if hasattr(value, "reshape") else value

    def reduce(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force):
            dist.all_reduce(self.buf, group=self.group)
        return self

    # Nothing in the next iteration depends on the reduced values (loss scalars for the log), so the collective need not sit
    # on the compute stream's critical path: reduce_async() snapshots the buffer into one of RING staging buffers and starts
    # the all-reduce without making the caller's stream wait for it; the staging buffer is only waited for when its turn
    # comes again (RING iterations later, long finished) or in finish(), which leaves the newest result in self.buf.
    RING = 4

    def reduce_async(self, src=None):
        """Start the all-reduce of the current buffer contents (or of `src`, copied straight into the staging buffer: its
        first src.numel() entries, the others zero) without waiting for it."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.force)):
            return self
        if not hasattr(self, "_ring"):
            self._ring = [torch.zeros_like(self.buf) for _ in range(self.RING)]
            self._work = [None] * self.RING
            self._turn = 0
        i = self._turn % self.RING
        if self._work[i] is not None:
            self._work[i].wait()
        if src is None:
            self._ring[i].copy_(self.buf)
        else:
            self._ring[i][:src.numel()].copy_(src.reshape(-1))
        self._work[i] = dist.all_reduce(self._ring[i], group=self.group, async_op=True)
        self._last = i
        self._turn += 1
        return self

    def finish(self):
        """Wait for every outstanding reduce_async(); self.buf then holds the most recent reduced values."""
        if getattr(self, "_work", None) is None:
            return self
        for w in self._work:
            if w is not None:
                w.wait()
        self._work = [None] * self.RING
        if getattr(self, "_last", None) is not None:
            self.buf.copy_(self._ring[self._last])
            self._last = None
        return self

    def get(self, name):
        off, n = self.offsets[name]
        return self.buf[off:off + n]
