"""Fused mapping iteration: the same arithmetic as

    out = render(settings, params.activated()); loss = mapping_loss(out, gt, mask); loss.backward(); adam.step()

(dqo_harness/mapping.py, i.e. SLAM/multiprocess/mapper.py:531-605 + 836-875 + gaussian_pointcloud.py:331-378), but without
the autograd graph and the ~110 eager launches per iteration: activation kernel -> rasteriser forward -> fused masked-loss
forward+backward -> rasteriser backward -> fused (activation-Jacobian + Adam) kernel.  The rasteriser is called through the
same operator code (`_RasterizeGaussians.forward / .backward`), so it is the same C-ABI path the drop-in op uses.

`capture()` records one whole iteration (eight kernel launches since round 3: zero fill, preprocess, binning, two sort kernels, forward
blend with the loss tap, backward blend, and ONE per-Gaussian tail — record sums, the per-Gaussian chain and Adam; no memset node, no
host synchronisation, the Adam step count, the attach set and its size kept on the device) into a hipGraph over persistent buffers;
`replay()` re-issues it with a single launch call — at ~0.53 ms of GPU work per iteration on config 3 the per-launch host work of the
eager path is otherwise longer than the GPU work.  `reserve()` + `grow()` + `begin_mapping_call()` keep the captured graph valid across
map-growth steps and mapping calls (everything they change is rewritten in place).

Round 6 — the reference's optimise LOOP, not one frame of it (mapper.py:531-605, 1105-1228): a frame set (`capture_window`: one graph
per frame of the window / keyframe selection, each with its own context buffers and capacities, all sharing parameters, moments and the
device-side step count; `replay(frame)` = the per-iteration frame choice, `run_window` = the reference's schedule), the two clouds in
one map (`set_training_rows`: the rows the call trains and the rows it renders; frozen rows are rendered and back-propagated through
but are no parameters of the call), per-call learning rates in device memory (`set_lrs`), the per-iteration confidence counter
(`confidence`, mapper.py:908-910) inside the tail kernel, and `history_merge()` (mapper.py:607-650) as one kernel.

With a render mask (every live call site of the reference) the loss is the masked L1 pair and the reference skips SSIM (B14); without
one the SSIM term of mapper.py:839-845 is added by dqo_map_ssim_fwd_bwd (three launches; the loss tap is off then — the SSIM gradient
is an image).  GPU only.
"""
import ctypes
import os

import numpy as np
import torch

import _dqo_native as N
import diff_gaussian_rasterization_depth as dgr
from dqo_harness import mapping


def _normalised_settings(st, device):
    """The op's forward makes its settings tensors fp32 + contiguous on every call (diff_gaussian_rasterization_depth/__init__.py);
    the captured path takes raw pointers once, so it does the same once.  The reference's cameras hand over
    `world_view_transform = torch.tensor(...).transpose(0, 1).cuda()` (scene/cameras.py:137-139): a non-contiguous view whose
    data_ptr() is the UN-transposed matrix."""
    def fix(t, name, n):
        if not torch.is_tensor(t):
            raise RuntimeError(f"raster settings: {name} must be a tensor")
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected scalar type Float but found {t.dtype} ({name})")
        if not t.is_cuda:
            raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
        if t.numel() != n:
            raise RuntimeError(f"raster settings: {name} must have {n} elements")
        return t.to(device).contiguous()
    return st._replace(bg=fix(st.bg, "bg", 3), viewmatrix=fix(st.viewmatrix, "viewmatrix", 16), projmatrix=fix(st.projmatrix, "projmatrix", 16),
                       campos=fix(st.campos, "campos", 3))


def _checked_tile_mask(tm, device, H, W):
    if tm.dtype != torch.int32:
        raise RuntimeError(f"expected scalar type Int but found {tm.dtype} (tile_mask)")
    if not tm.is_cuda:
        raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path.")
    if tm.numel() != ((H + 15) // 16) * ((W + 15) // 16):
        raise RuntimeError("tile_mask must have ceil(H/16) x ceil(W/16) elements")
    return tm.to(device).contiguous()


def tile_object_sets(pixel_object):
    """DqoObjectGate.tile_objects of a pixel_object map (int32 [H, W], ids in [0, 64), < 0 = no owner): per 16x16 tile the 64-bit set of
    the owners among its pixels, as an int64 tensor [ceil(H/16) * ceil(W/16)] (bit patterns; row-major tiles)."""
    H, W = pixel_object.shape
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = torch.full((gy * 16, gx * 16), -1, dtype=torch.int64, device=pixel_object.device)
    pad[:H, :W] = pixel_object
    t = pad.reshape(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256)
    bits = torch.where(t >= 0, torch.ones_like(t) << t.clamp(min=0), torch.zeros_like(t))
    out = bits[:, 0].clone()
    for j in range(1, 256):  # (bitwise OR over the tile's pixels; once per mapping call)
        out |= bits[:, j]
    return out.contiguous()


class _Ctx:
    """Stand-in for the autograd ctx when the op's static forward/backward are driven directly."""

    def save_for_backward(self, *a):
        self.saved_tensors = a

    def mark_non_differentiable(self, *a):
        pass


class FusedMapper:
    def __init__(self, scene, settings, device, lrs=None, betas=(0.9, 0.999), eps=1e-15, color_weight=mapping.COLOR_WEIGHT,
                 depth_weight=mapping.DEPTH_WEIGHT, add_depth_thres=0.1, sparse_moments=True, attach=True, attach_count_reducer=None):
        t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=device)
        self.device = device
        self.settings = _normalised_settings(settings, device)
        self.xyz = t(scene["xyz"])
        self.shs = t(scene["shs"])  # [P, M, 3]; coefficient 0 is f_dc, the rest f_rest (no torch.cat per iteration)
        op = t(scene["opacity"]).clamp(1e-4, 1 - 1e-4)
        self.opacity_raw = torch.log(op / (1 - op))
        self.scaling_raw = torch.log(t(scene["scales"]))
        self.rotation_raw = t(scene["rotations"]).clone()
        self.P, self.M = self.xyz.shape[0], self.shs.shape[1]
        self.lrs = dict(mapping.LRS if lrs is None else lrs)
        self.betas, self.eps = betas, eps
        self.color_weight, self.depth_weight, self.add_depth_thres = color_weight, depth_weight, add_depth_thres
        self.state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in self._params().items()}
        # exact sparse Adam (DqoAdamStep.moment_live): 0 = the Gaussian's moments are still identically zero
        self.moment_live = torch.zeros((self.xyz.shape[0],), dtype=torch.uint8, device=device) if sparse_moments else None
        self.step_count = 0
        # Device-side step state SHARED by every captured graph of this mapper (one graph per frame of a window: capture_window):
        # DqoAdamStep.step_dev / block_ticket / bias_table.  _expected_step = what step_dev holds while host and device counts agree.
        i32 = dict(dtype=torch.int32, device=device)
        self._step_dev = torch.full((1,), 1, **i32)
        self._ticket = torch.zeros((16 + 16 * 64,), **i32)  # DQO_TICKET_WORDS
        self._bias = torch.zeros((8,), dtype=torch.float32, device=device)
        self._expected_step, self._unsettled = 1, False
        # DqoAdamStep.block_ticket: the Adam launch advances the device step count itself (False: a one-thread launch behind it does)
        self.use_block_ticket = True
        self._frames = []  # the captured graphs of a window (capture_window); self._g = the one replayed last / by default
        self._mixed = {}   # (k, m) -> graph of frame k's camera and target under frame m's masks (global_optimization's quirk)
        # The reference keeps TWO clouds and trains one of them per mapping call while it renders one or both (mapper.py:533, 578,
        # 1119, 1199-1204).  One map here: DqoAdamStep.row_flags / DqoRastInputs.row_flags (bit 0 = not trained, bit 1 = not rendered),
        # rewritten in place by set_training_rows — a captured graph follows.
        self.row_flags = torch.zeros((self.P,), dtype=torch.uint8, device=device)
        self._first_trained_row = 0
        # `_confidence` (SLAM/gaussian_pointcloud.py:42, 836): += 1 per iteration for every trained row with a non-zero f_dc gradient
        # (mapper.py:908-910), counted by the Adam launch itself (DqoAdamStep.confidence)
        self.confidence = torch.zeros((self.P,), dtype=torch.float32, device=device)
        self.count_confidence = True
        # the six groups' learning rates in device memory (DqoAdamStep.lr_table): set_lrs rewrites them between two mapping calls
        self.lr_table = torch.zeros((6,), dtype=torch.float32, device=device)
        self._write_lr_table()
        self.history_merge_weight = 0.5  # history_merge_max_weight, configs/base.yaml:54
        self.init_shs = self.init_confidence = None  # history_stat's other members (begin_mapping_call(history=True))
        self._act_valid = False  # opacity / scales / rotations hold the activations of the current raw parameters
        self.use_attach = bool(attach)
        # The attach loss is a mean over the attach set of the WHOLE map (mapper.py:812-829).  A mapper that holds a shard of the map
        # must divide by the whole map's count for the shards to add up to the unsharded job: attach_count_reducer(local count) -> global
        # count (e.g. one all-reduce of one integer per mapping call — not per iteration); None = this mapper holds the whole map.
        self.attach_count_reducer = attach_count_reducer
        self.gaussian_object = self.pixel_object = self.tile_objects = None  # set_object_gate()
        self.object_cell = None  # per-object growth decisions: cell size of dqo_mapgrowth.object_offsets (None = its 16 m default)
        self.per_object_loss = False
        self.alive = None  # reserve(): uint8 [P], 0 = a spare row (parked behind the camera, no Gaussian of the map)
        self._n_spare = 0
        # DqoAdamStep.attach_gains: the attach term's two factors in device memory, rewritten in place by begin_mapping_call — a captured
        # graph survives a new mapping call
        self.attach_gains = torch.zeros((2,), dtype=torch.float32, device=device)
        self.begin_mapping_call(reset_optimizer=False)
        P = self.P
        f = dict(dtype=torch.float32, device=device)
        self.opacity = torch.empty((P, 1), **f)
        self.scales = torch.empty((P, 3), **f)
        self.rotations = torch.empty((P, 4), **f)
        H, W = settings.image_height, settings.image_width
        self.dL_dcolor = torch.empty((3, H, W), **f)
        self.dL_ddepth = torch.empty((1, H, W), **f)
        self.loss = torch.zeros(8, **f)  # dqo_map_loss_fwd_bwd: total, colour, depth, 0, then the four unnormalised sums
        lib = N.lib()
        self.loss_ws = torch.empty((lib.dqo_map_loss_workspace_bytes(),), dtype=torch.uint8, device=device)
        # the unmasked branch of Mapping.loss_update adds 0.2 * (1 - ssim) (mapper.py:839-845): dqo_map_ssim_fwd_bwd, buffers made on first use
        self.ssim_weight = mapping.SSIM_WEIGHT
        self.ssim_out = self.ssim_ws = None
        self._empty = torch.Tensor([])
        self.tile_mask = torch.ones(((H + 15) // 16, (W + 15) // 16), dtype=torch.int32, device=device)

    def set_object_gate(self, gaussian_object, pixel_object, per_object_loss=True):
        """The per-object job of SURVEY.md §8(e) (not a reference feature): gaussian_object int32 [P] (every Gaussian's object id, >= 0),
        pixel_object int32 [H, W] (every pixel's owner, < 0 = none).  A list entry then acts on a pixel only if the ids agree
        (DqoObjectGate), and — per_object_loss — the loss is the sum of the objects' own masked losses (DqoLossTap.per_object;
        ids in [0, 64)), so that what an object learns does not depend on which other objects this mapper holds: shards of one map add up
        to the unsharded job.  None, None switches the gate off (the reference's semantics).  A captured graph must be captured again."""
        if gaussian_object is None:
            self.gaussian_object = self.pixel_object = self.tile_objects = None
            self.per_object_loss = False
        else:
            H, W = int(self.settings.image_height), int(self.settings.image_width)
            go = torch.as_tensor(gaussian_object).to(self.device, torch.int32).contiguous().reshape(-1)
            po = torch.as_tensor(pixel_object).to(self.device, torch.int32).contiguous().reshape(H, W)
            if go.numel() != self.P:
                raise RuntimeError("set_object_gate: gaussian_object must have one id per Gaussian")
            if int(go.min().item()) < 0 or int(go.max().item()) > 63 or int(po.max().item()) > 63:
                raise RuntimeError("set_object_gate: object ids must lie in [0, 64)")
            self.gaussian_object, self.pixel_object, self.per_object_loss = go, po, bool(per_object_loss)
            self.tile_objects = tile_object_sets(po)  # DqoObjectGate.tile_objects: which objects own a pixel of each 16x16 tile
        for g in self._graphs():
            g.stale = True
        return self

    # ------------------------------------------------------------------ the two clouds, per-call learning rates ---------
    def _graphs(self):
        gs = [g for g in getattr(self, "_frames", []) if g is not None] + list(getattr(self, "_mixed", {}).values())
        g = getattr(self, "_g", None)
        if g is not None and all(g is not f for f in gs):
            gs.append(g)
        return gs

    _LR_ORDER = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")

    def _write_lr_table(self):
        # (fill_ carries the python float as a kernel argument: no pageable host copy, no synchronisation)
        for i, k in enumerate(self._LR_ORDER):
            self.lr_table[i:i + 1].fill_(float(self.lrs[k]))

    def set_lrs(self, lrs=None, **scale):
        """The learning rates of the NEXT mapping call, rewritten in device memory (DqoAdamStep.lr_table): captured graphs stay valid.
        `lrs`: dict over xyz / f_dc / f_rest / opacity / scaling / rotation (missing keys keep their value); `scale`: factors on the
        current values, e.g. Mapping.global_optimization's  l[0]["lr"] = 0; l[i]["lr"] *= 0.1  (mapper.py:1120-1123) is
        set_lrs(dict(xyz=0.0), f_dc=0.1, f_rest=0.1, opacity=0.1, scaling=0.1, rotation=0.1).  The bias-correction table of the device
        step is dropped (it holds products with the old rates)."""
        if lrs:
            self.lrs.update({k: float(v) for k, v in lrs.items()})
        for k, f in scale.items():
            self.lrs[k] = self.lrs[k] * float(f)
        self._write_lr_table()
        self._bias.zero_()
        return self

    @torch.no_grad()
    def set_training_rows(self, trainable=None, rendered=None):
        """Which rows the next mapping call TRAINS and which it RENDERS (bool [P] GPU tensors or None = all).  The reference's two clouds:
          local_optimize        trains `pointcloud` (unstable), renders cat(unstable, stable)      (mapper.py:533, 578, 1810-1840)
                                -> set_training_rows(trainable=~stable_mask)
          global_optimization   trains and renders `stable_pointcloud` alone                        (mapper.py:1119, 1199-1204)
                                -> set_training_rows(trainable=stable_mask, rendered=stable_mask)
        A row that is not trained is rendered and back-propagated THROUGH (its list entries shape every pixel it touches) but is no
        parameter of the call: no gradient row is formed, parameters, moments and confidence stay bit for bit, it is in no attach set.
        A row that is not rendered is culled by the per-Gaussian forward (and not trained).  Written in place (DqoAdamStep.row_flags /
        DqoRastInputs.row_flags live in device memory): captured graphs stay valid.  Call before begin_mapping_call (the attach set and
        history_merge's "first row" follow the trained rows)."""
        dev = self.device
        fl = torch.zeros((self.P,), dtype=torch.uint8, device=dev)
        if trainable is not None:
            fl |= (~trainable.to(dev).bool().reshape(-1)).to(torch.uint8) * N.ROW_FROZEN
        if rendered is not None:
            fl |= (~rendered.to(dev).bool().reshape(-1)).to(torch.uint8) * (N.ROW_HIDDEN | N.ROW_FROZEN)
        if self.alive is not None:  # a spare row is no Gaussian of the map, whichever camera looks its way
            fl |= (self.alive == 0).to(torch.uint8) * (N.ROW_HIDDEN | N.ROW_FROZEN)
        self.row_flags.copy_(fl)
        # row 0 of the reference's trained cloud: its history weight serves every row's feature / scaling merge (mapper.py:620-637)
        self._first_trained_row = int(torch.argmax(((fl & N.ROW_FROZEN) == 0).to(torch.uint8)).item())
        return self

    def trained_rows(self):
        """bool [P]: the rows the current mapping call trains."""
        return (self.row_flags & N.ROW_FROZEN) == 0

    @torch.no_grad()
    def history_merge(self, max_weight=None):
        """Mapping.history_merge (mapper.py:607-650), the statement that closes every local_optimize call: the trained rows are pulled back
        towards their state at the start of the call (begin_mapping_call(history=True) took `history_stat`), weighted by the share of
        their confidence that is older than the call.  One launch (dqo_map_history_merge); the activations are recomputed."""
        w = self.history_merge_weight if max_weight is None else float(max_weight)
        if w <= 0:
            return self
        if self.init_shs is None or self.init_confidence is None or self.init_shs.shape[0] != self.P:
            raise RuntimeError("FusedMapper.history_merge: begin_mapping_call(history=True) must have taken history_stat for this map")
        rot0 = torch.nn.functional.normalize(self.init_rotation)  # history_stat["rotation"] = get_rotation (mapper.py:541)
        with torch.cuda.device(self.device):
            N.check(N.lib().dqo_map_history_merge(self.P, self.M, w, self._first_trained_row, N.ptr(self.row_flags), N.ptr(self.init_confidence),
                                                  N.ptr(self.confidence), N.ptr(self.init_xyz), N.ptr(self.init_shs), N.ptr(self.init_scaling),
                                                  N.ptr(rot0), N.ptr(self.xyz), N.ptr(self.shs), N.ptr(self.scaling_raw),
                                                  N.ptr(self.rotation_raw), N.current_stream()))
        self._act_valid = False
        self.activate()  # (a captured iteration starts from the activations of the current parameters)
        return self

    def begin_mapping_call(self, reset_optimizer=True, history=False):
        """Start of one `local_optimize` call (SLAM/multiprocess/mapper.py:531-548): snapshot `init_stat` (the raw parameters the
        attach loss pulls towards, :533-545, and the attach set `sigmoid(opacity) < 0.9`, :812-813) and — reset_optimizer — drop the
        Adam moments, because the reference builds a fresh torch.optim.Adam per call (:548, B13).  Everything is rewritten in place
        when the map has kept its size (buffers, attach set and its size live in device memory: a captured graph stays valid)."""
        P = self.xyz.shape[0]
        mask = (torch.sigmoid(self.opacity_raw) < 0.9).reshape(-1)
        if self.alive is not None:
            mask &= self.alive.bool()  # (a spare row is no Gaussian of the map)
        if getattr(self, "row_flags", None) is not None and self.row_flags.shape[0] == P:
            mask &= (self.row_flags & N.ROW_FROZEN) == 0  # init_stat is the TRAINED cloud's (mapper.py:533-545, 1132-1137)
        if history:  # history_stat's members that only history_merge reads (mapper.py:535-540)
            if self.init_shs is not None and self.init_shs.shape[0] == P:
                self.init_shs.copy_(self.shs), self.init_confidence.copy_(self.confidence)
            else:
                self.init_shs, self.init_confidence = self.shs.clone(), self.confidence.clone()
        same = getattr(self, "init_xyz", None) is not None and self.init_xyz.shape[0] == P and self.attach_mask.shape[0] == P
        if same:
            self.init_xyz.copy_(self.xyz), self.init_scaling.copy_(self.scaling_raw), self.init_rotation.copy_(self.rotation_raw)
            self.attach_mask.copy_(mask)
            self.attach_partial.zero_()
        else:
            self.init_xyz, self.init_scaling, self.init_rotation = self.xyz.clone(), self.scaling_raw.clone(), self.rotation_raw.clone()
            self.attach_mask = mask.to(torch.uint8).contiguous()
            # (one partial sum per block of 256 Gaussians from dqo_map_adam_step, per wave of 64 from dqo_rast_backward_adam)
            self.attach_partial = torch.zeros((4 * ((P + 255) // 256),), dtype=torch.float32, device=self.device)
        self._count_attach_set()
        self._attach_n = 0
        if reset_optimizer:
            for m, v in self.state.values():
                m.zero_(), v.zero_()
            if self.moment_live is not None:
                self.moment_live.zero_()
            self.step_count = 0
            if getattr(self, "_step_dev", None) is not None:
                self._step_dev.fill_(1)
                self._expected_step, self._unsettled = 1, False
        for g in self._graphs():
            if not (same and getattr(g, "attach_in_place", False)):
                g.stale = True  # the captured kernel arguments (attach set buffers) were fixed at capture time

    def _count_attach_set(self):
        self.attach_count = int(self.attach_mask.sum().item()) if self.use_attach else 0
        if self.use_attach and self.attach_count_reducer is not None:
            self.attach_count = int(self.attach_count_reducer(self.attach_count))
        n = self.attach_count
        # the two factors exactly as the library derives them from attach_count: double arithmetic, rounded to float once
        # (fill_ carries the python double as a kernel argument and rounds it to float once; a host tensor copied to the device would be
        # a synchronous pageable copy: ~1 ms per mapping call)
        self.attach_gains[0:1].fill_(2000.0 / (3.0 * n) if n > 0 else 0.0)
        self.attach_gains[1:2].fill_(2000.0 / (4.0 * n) if n > 0 else 0.0)

    def activate(self):
        """(opacity [P,1], scales [P,3], rotations [P,4]): the activations of the current raw parameters (SLAM/gaussian_pointcloud.py:
        732-733, 746-747), i.e. what the rasteriser sees in the next iteration — computed if they are not up to date.  The tensors are
        this mapper's own buffers: the next Adam step overwrites them."""
        if not self._act_valid:
            with torch.cuda.device(self.device):
                N.check(N.lib().dqo_map_activate(self.P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                                 N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), N.current_stream()))
            self._act_valid = True
        return self.opacity, self.scales, self.rotations

    # ------------------------------------------------------------------ map growth ---------------------------------------
    def radius(self):
        """GaussianPointCloud.get_radius (SLAM/gaussian_pointcloud.py:739-743): mean of the two larger scales."""
        sc = torch.exp(self.scaling_raw)
        return (sc.sum(dim=1) - sc.min(dim=1).values) / 2

    def normals(self, rows=None):
        """GaussianPointCloud.get_normal (SLAM/gaussian_pointcloud.py:780-791): the column of R(q / |q|) along the smallest scale,
        normalised with the reference's + 1e-8 (utils/general_utils.py:108-137 build_rotation)."""
        q = self.rotation_raw if rows is None else self.rotation_raw[rows]
        sc = self.scaling_raw if rows is None else self.scaling_raw[rows]
        q = q / torch.sqrt((q * q).sum(1, keepdim=True))
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        cols = (torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)], 1),
                torch.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)], 1),
                torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], 1))
        k = torch.argmin(sc, dim=1)  # (exp is monotone: the smallest raw scale is the smallest scale)
        nrm = torch.where((k == 0)[:, None], cols[0], torch.where((k == 1)[:, None], cols[1], cols[2]))
        return nrm / (torch.sqrt((nrm * nrm).sum(1, keepdim=True)) + 1e-8)

    def _park_position(self):
        """Where spare rows sit: 10^4 units behind the camera of this mapper's settings — culled by the frustum test (p_view.z <= 0.2,
        forward.cu:258-262) like any Gaussian behind the camera, and far from every search box of the growth step."""
        V = self.settings.viewmatrix  # world_view_transform as the reference passes it: p_view = p @ V[:3, :3] + V[3, :3]
        return (self.settings.campos.reshape(3) - 1.0e4 * V[:3, 2]).to(torch.float32)

    @torch.no_grad()
    def reserve(self, spare_rows):
        """Room for `spare_rows` more Gaussians in every per-Gaussian buffer, so that a growth step writes the new Gaussians into spare
        rows and turns deleted ones into spare rows IN PLACE: no re-allocation, and a captured graph (whose kernel arguments are these
        buffers and their length) stays valid across growth steps and mapping calls.  A spare row is parked behind the camera
        (_park_position) with the raw parameters of a tiny transparent Gaussian: the preprocess stage culls it, it gets no gradient, the
        sparse Adam never touches it, it is in no attach set — the map behaves as if the row did not exist, at the cost of the
        per-Gaussian kernels looking at it.  `alive` [P] uint8 tells the rows apart.  Call before capture()."""
        n = int(spare_rows)
        if n <= 0:
            return self
        dev, P0 = self.device, self.xyz.shape[0]

        def pad(a, fill):
            tail = torch.empty((n,) + tuple(a.shape[1:]), dtype=a.dtype, device=dev)
            tail[:] = fill if not torch.is_tensor(fill) else fill.to(a.dtype)
            return torch.cat([a, tail]).contiguous()

        park = self._park_position()
        import dqo_mapgrowth as mg
        unit_q = mg.const_tensor([1.0, 0.0, 0.0, 0.0], dev)
        self.alive = pad(self.alive if self.alive is not None else torch.ones((P0,), dtype=torch.uint8, device=dev), 0)
        self.init_xyz, self.init_scaling, self.init_rotation = pad(self.init_xyz, park), pad(self.init_scaling, -10.0), pad(self.init_rotation, unit_q)
        self.xyz, self.shs = pad(self.xyz, park), pad(self.shs, 0.0)
        self.opacity_raw, self.scaling_raw, self.rotation_raw = pad(self.opacity_raw, -10.0), pad(self.scaling_raw, -10.0), pad(self.rotation_raw, unit_q)
        self.state = {k: (pad(m, 0.0), pad(v, 0.0)) for k, (m, v) in self.state.items()}
        if self.moment_live is not None:
            self.moment_live = pad(self.moment_live, 0)
        if self.gaussian_object is not None:
            self.gaussian_object = pad(self.gaussian_object, 0)
        self.attach_mask = pad(self.attach_mask, 0)
        self.row_flags = pad(self.row_flags, N.ROW_HIDDEN | N.ROW_FROZEN)  # (a spare row: not rendered, not trained)
        self.confidence = pad(self.confidence, 0.0)
        self.init_shs = self.init_confidence = None
        self._spare_rows = n
        self._n_spare = int(self.alive.numel() - int(self.alive.sum().item()))  # (host copy of the spare-row count: grow() keeps it up to date)
        self.P = P = P0 + n
        f = dict(dtype=torch.float32, device=dev)
        self.opacity, self.scales, self.rotations = torch.empty((P, 1), **f), torch.empty((P, 3), **f), torch.empty((P, 4), **f)
        self.attach_partial = torch.zeros((4 * ((P + 255) // 256),), **f)
        self._attach_n = 0
        self._act_valid = False
        self._g, self._frames, self._mixed = None, [], {}
        return self

    @property
    def n_alive(self):
        return self.P if self.alive is None else int(self.alive.sum().item())

    @torch.no_grad()
    def grow(self, new, delete_mask=None, min_radius=0.001, max_radius=0.05, xyz_factor=(1.0, 1.0, 0.1), scale_factor=1.0,
             new_mapping_call=False, stable_mask=None, unstable_opacity_low=0.1, attach_async=True):
        """The map-growth step between two mapping calls — Mapping.gaussians_add (SLAM/multiprocess/mapper.py:249-254) and the
        deletion half of error_gaussians_remove (:1086-1096) — on this mapper's map:
          1. temp_points_filter (:1351-1380): new points that fall inside an existing Gaussian (one of their 3 nearest existing
             centres closer than 0.6 x its radius; dqo_knn3_query) are dropped;
          2. temp_to_optimize -> GaussianPointCloud.update_geometry (:1438-1442, gaussian_pointcloud.py:519-570): the survivors'
             scales come from the gaps to their 3 nearest neighbours among (survivors + existing points in their bounding box)
             (distCUDA2 = dqo_knn3), points whose neighbours' 3-sigma spheres already reach them are dropped;
          3. delete_mask [P] bool (optional): existing Gaussians to delete (the reference derives it from
             accumulate_gaussian_error's per-Gaussian depth error, cuda_utils._C);
          4. cat (:1466): the rest joins the map with zero Adam moments.
        stable_mask [P] bool GPU tensor (optional) is the reference's split into its two clouds — 1 = a Gaussian of `stable_pointcloud`, 0 =
        of the unstable `pointcloud`; an in-place step clears it IN PLACE on the rows the new Gaussians take — and switches on the two
        steps that need it:
          1'. the filter of step 1 looks at the UNSTABLE Gaussians only (:1356-1357, unstable_params);
          1b. temp_points_attach (:1384-1436): the survivors of step 1 that project onto a pixel whose strongest contributor in a render
             of the STABLE Gaussians alone exists and whose plane they lie within 0.5 x add_depth_thres of get opacity
             `unstable_opacity_low` — which makes them members of the next mapping call's attach set (opacity < 0.9).  The stable-only
             render is this mapper's map with the other Gaussians parked behind the camera (the index map is the stable cloud's, in
             map rows).
        With reserve()d spare rows and new_mapping_call=True the step is IN PLACE: deleted Gaussians become spare rows, new ones take
        spare rows (row order then differs from the reference's cat; nothing depends on it), no buffer moves and a captured graph
        stays valid; it falls back to re-allocation (with the same number of spare rows again) when the spare rows run out.  `new`: dict(xyz [Q,3], scales [Q,3], rotations [Q,4], opacity [Q,1], shs [Q,M,3]) of numpy arrays or GPU
        tensors.  Per-Gaussian buffers are re-allocated: a captured graph is dropped (capture() again), and the next mapping
        call starts with begin_mapping_call() — new_mapping_call=True does that here (fresh Adam, fresh init_stat, as the reference
        does after every growth step, mapper.py:533-548) and then neither gathers nor concatenates the old moments and snapshots,
        which the new call would overwrite (a third of the step's copies).  Returns the counts of each stage."""
        import dqo_mapgrowth as mg
        dev = self.device
        t = lambda a: a.to(dev).float().contiguous() if torch.is_tensor(a) else torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
        nx, nsc, nrot, nop, nsh = t(new["xyz"]), t(new["scales"]), t(new["rotations"]), t(new["opacity"]).reshape(-1, 1), t(new["shs"])
        Q = nx.shape[0]
        nobj = None
        if self.gaussian_object is not None:  # the object gate needs every new Gaussian's object id (`_obj_id`, gaussian_pointcloud.py:497)
            if new.get("obj_id") is None:
                raise RuntimeError("FusedMapper.grow: with an object gate the new points need 'obj_id'")
            nobj = torch.as_tensor(new["obj_id"]).to(dev, torch.int32).reshape(-1)
        stats = dict(candidates=int(Q), inside_existing=0, invalid_scale=0, added=0, deleted=0)
        exist_xyz, exist_radius = self.xyz, self.radius()  # (spare rows sit 10^4 units away: outside every search box)
        keep = torch.ones((Q,), dtype=torch.bool, device=dev)
        # The per-object job (set_object_gate): every decision of the step judges a candidate against the Gaussians of its OWN object
        # only (dqo_mapgrowth.*_per_object), so a shard — which holds whole objects — takes exactly the decisions the unsharded map
        # takes for its objects: the N-rank map grows like the N = 1 map.  Without a gate: the reference's decisions.
        per_obj = nobj is not None
        live_rows = None if self.alive is None else self.alive.bool()
        attach_job = None
        if stable_mask is not None and Q > 0:
            # temp_points_attach only changes opacities and judges every candidate by itself; the filter and update_geometry only read
            # positions and radii: the attach runs beside BOTH, on ALL candidates (the ones the filter drops are dropped from its
            # answer afterwards) — on a stream and a host thread of its own (both sides wait on the host for small results in between
            # their kernels).  Round 4 started it behind the filter: the step's critical path was filter + attach.
            import threading
            if not self._act_valid:
                N.check(N.lib().dqo_map_activate(self.P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                                 N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), N.current_stream()))
                self._act_valid = True
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=dev)  # (made once: creating a stream costs a growth step ~1 ms)
            side, box = self._side_stream, {}
            side.wait_stream(torch.cuda.current_stream())

            def attach_work(tx=nx, to=nop, tobj=nobj):
                try:
                    with torch.cuda.device(dev), torch.cuda.stream(side), torch.no_grad():
                        box["att"] = self._temp_points_attach(tx, to, stable_mask, unstable_opacity_low, temp_obj=tobj)
                except BaseException as e:  # (re-raised by the caller's thread)
                    box["err"] = e

            if attach_async:
                attach_job = threading.Thread(target=attach_work)
                attach_job.start()
            else:  # (A/B of the overlap: the same work in line)
                attach_work()
                attach_job = threading.Thread(target=lambda: None)
                attach_job.start()
        if Q > 0:
            if stable_mask is None and not per_obj:
                inside = mg.temp_points_filter_mask(nx, exist_xyz, exist_radius)
            else:  # the reference filters against its unstable cloud
                un = torch.ones((self.P,), dtype=torch.bool, device=dev) if stable_mask is None else ~stable_mask.to(dev).bool().reshape(-1)
                if live_rows is not None:
                    un = un & live_rows  # (spare rows are no Gaussians of the map)
                un = un.nonzero().reshape(-1)  # (the unstable cloud is small: a few thousand rows of a 2 M map)
                if per_obj:
                    inside = mg.temp_points_filter_mask_per_object(nx, nobj, exist_xyz[un], exist_radius[un], self.gaussian_object[un],
                                                                   cell=self.object_cell)
                else:
                    inside = mg.temp_points_filter_mask(nx, exist_xyz[un], exist_radius[un])
            if inside is not None:
                keep &= ~inside
                stats["inside_existing"] = int(inside.sum().item())
        idx = keep.nonzero().reshape(-1)
        nx, nsc, nrot, nop, nsh = nx[idx], nsc[idx], nrot[idx], nop[idx], nsh[idx]
        nobj = None if nobj is None else nobj[idx]
        log_scales = None
        if nx.shape[0] > 0:
            nrad = (nsc.sum(dim=1) - nsc.min(dim=1).values) / 2
            if per_obj:
                gobj_live = self.gaussian_object if live_rows is None else torch.where(live_rows, self.gaussian_object, -1)
                scales, invalid = mg.update_geometry_scales_per_object(nx, nobj, nrad, exist_xyz, exist_radius, gobj_live, min_radius, max_radius,
                                                                       cell=self.object_cell)
            else:
                scales, invalid = mg.update_geometry_scales(nx, nrad, exist_xyz, exist_radius, min_radius, max_radius)
        if attach_job is not None:
            attach_job.join()
            torch.cuda.current_stream().wait_stream(side)
            if "err" in box:
                raise box["err"]
            # (the attach judged all Q candidates: keep its answer for the ones the filter kept, as positions among them)
            att_all = torch.zeros((Q,), dtype=torch.bool, device=dev)
            att_all.index_fill_(0, box["att"], True)  # (x[rows] = scalar stages the scalar through a host tensor: a blocking copy)
            att = att_all[idx].nonzero().reshape(-1)
            stats["attached"] = int(att.numel())
            nop = nop.clone()
            nop.index_fill_(0, att, unstable_opacity_low)
        if nx.shape[0] > 0:
            stats["invalid_scale"] = int(invalid.sum().item())
            ok = (~invalid).nonzero().reshape(-1)
            if ok.numel() > 0:  # gaussian_pointcloud.py:558-568
                fac = scale_factor * scales[:, None].repeat(1, 3) * mg.const_tensor(xyz_factor, dev)
                log_scales = torch.log(fac)[ok]
            nx, nrot, nop, nsh = nx[ok], nrot[ok], nop[ok], nsh[ok]
            nobj = None if nobj is None else nobj[ok]
        if log_scales is None:
            nx, nrot, nop, nsh = nx[:0], nrot[:0], nop[:0], nsh[:0]
            nobj = None if nobj is None else nobj[:0]
            log_scales = torch.empty((0, 3), dtype=torch.float32, device=dev)
        stats["added"] = n_add = int(nx.shape[0])
        opc = nop.clamp(1e-4, 1 - 1e-4)
        spare = 0 if self.alive is None else self._n_spare
        if self.alive is not None:
            # (the deleted rows as indices, found once: four boolean-mask writes were four passes over the map + four host round trips)
            if delete_mask is None:
                del_rows = torch.empty((0,), dtype=torch.long, device=dev)
            else:
                del_rows = (delete_mask.to(dev).bool().reshape(-1) & live_rows).nonzero().reshape(-1)
            stats["deleted"] = n_del = int(del_rows.numel())
            if new_mapping_call and n_add <= spare + n_del:
                # ---- in place: deleted Gaussians become spare rows, the new ones take spare rows ----
                if n_del:
                    self.alive.index_fill_(0, del_rows, 0)
                    self.xyz[del_rows] = self._park_position()
                    self.opacity_raw.index_fill_(0, del_rows, -10.0), self.scaling_raw.index_fill_(0, del_rows, -10.0)
                    self.row_flags.index_fill_(0, del_rows, N.ROW_HIDDEN | N.ROW_FROZEN)  # a spare row again
                    self.confidence.index_fill_(0, del_rows, 0.0)
                self._n_spare += n_del - n_add
                if n_add:
                    rows = (self.alive == 0).nonzero().reshape(-1)[:n_add]
                    self.xyz[rows], self.shs[rows], self.rotation_raw[rows] = nx, nsh, nrot
                    self.opacity_raw[rows], self.scaling_raw[rows] = torch.log(opc / (1 - opc)), log_scales
                    if self.gaussian_object is not None:
                        self.gaussian_object[rows] = nobj
                    self.alive.index_fill_(0, rows, 1)
                    # what growth adds is rendered and trained by the next call (the unstable cloud, mapper.py:1438-1466), confidence 0
                    self.row_flags.index_fill_(0, rows, 0)
                    self.confidence.index_fill_(0, rows, 0.0)
                    stats["rows"] = rows
                    if stable_mask is not None:
                        # what growth adds belongs to the UNSTABLE cloud (mapper.py:1438-1466) — also when it lands in a row a deleted
                        # stable Gaussian just freed (spare rows are handed out lowest index first): the caller's mask is updated in
                        # place, so the next step's filter / stable-only render see the row as unstable
                        stable_mask.index_fill_(0, rows, False)
                # (a captured iteration starts from the activations its previous Adam launch left: bring them up to date for the new rows)
                N.check(N.lib().dqo_map_activate(self.P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                                 N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), N.current_stream()))
                self._act_valid = True
                self.begin_mapping_call(reset_optimizer=True)  # in place too: fresh moments, fresh init_stat, the new attach set
                stats["in_place"] = True
                return stats
            # spare rows exhausted (or the mapping call goes on): compact — spare rows go with the deleted ones — and reserve again
            delete_mask = ~live_rows
            delete_mask.index_fill_(0, del_rows, True)
            self.alive = None
            self._n_spare = 0
            stats["in_place"] = False
        keep_old = None
        if delete_mask is not None:
            keep_old = (~delete_mask.to(dev).bool().reshape(-1)).nonzero().reshape(-1)
            if "deleted" not in stats or stats.get("in_place") is None:
                stats["deleted"] = int(self.P - keep_old.numel())
        sel = (lambda a: a) if keep_old is None else (lambda a: a[keep_old])
        stats["kept_rows"] = keep_old  # (None = all of them, in place) the old rows that now lead the map, for per-row data of the caller's
        self.xyz = torch.cat([sel(self.xyz), nx]).contiguous()
        self.shs = torch.cat([sel(self.shs), nsh]).contiguous()
        self.opacity_raw = torch.cat([sel(self.opacity_raw), torch.log(opc / (1 - opc))]).contiguous()
        self.scaling_raw = torch.cat([sel(self.scaling_raw), log_scales]).contiguous()
        self.rotation_raw = torch.cat([sel(self.rotation_raw), nrot]).contiguous()
        if self.gaussian_object is not None:
            self.gaussian_object = torch.cat([sel(self.gaussian_object), nobj]).contiguous()
        self.row_flags = torch.cat([sel(self.row_flags), torch.zeros((nx.shape[0],), dtype=torch.uint8, device=dev)]).contiguous()
        self.confidence = torch.cat([sel(self.confidence), torch.zeros((nx.shape[0],), dtype=torch.float32, device=dev)]).contiguous()
        self.init_shs = self.init_confidence = None
        params = self._params()
        n_new = nx.shape[0]
        if new_mapping_call:
            self.state = {k: (torch.zeros_like(pv), torch.zeros_like(pv)) for k, pv in params.items()}
            if self.moment_live is not None:
                self.moment_live = torch.zeros((self.xyz.shape[0],), dtype=torch.uint8, device=dev)
        else:
            self.state = {k: (torch.cat([sel(m), torch.zeros_like(params[k][m.shape[0] if keep_old is None else keep_old.numel():])]),
                              torch.cat([sel(v), torch.zeros_like(params[k][v.shape[0] if keep_old is None else keep_old.numel():])]))
                          for k, (m, v) in self.state.items()}
            if self.moment_live is not None:
                self.moment_live = torch.cat([sel(self.moment_live), torch.zeros((n_new,), dtype=torch.uint8, device=dev)])
        self.P = P = self.xyz.shape[0]
        f = dict(dtype=torch.float32, device=dev)
        self.opacity, self.scales, self.rotations = torch.empty((P, 1), **f), torch.empty((P, 3), **f), torch.empty((P, 4), **f)
        self._act_valid = False
        self._g, self._frames, self._mixed = None, [], {}
        refill = stats.get("in_place") is False  # the map had spare rows and ran out of them: the same number again
        if new_mapping_call:
            self.init_xyz = None  # (sizes changed: begin_mapping_call takes fresh snapshots)
            self.begin_mapping_call(reset_optimizer=True)
            if refill:
                self.reserve(self._spare_rows)
            return stats
        # init_stat / attach set: kept for the old Gaussians, the new ones start at their own values (they have not moved);
        # a new mapping call re-snapshots everything (begin_mapping_call)
        self.init_xyz = torch.cat([sel(self.init_xyz), nx])
        self.init_scaling = torch.cat([sel(self.init_scaling), log_scales])
        self.init_rotation = torch.cat([sel(self.init_rotation), nrot])
        self.attach_mask = torch.cat([sel(self.attach_mask), (nop.reshape(-1) < 0.9).to(torch.uint8)]).contiguous()
        self._count_attach_set()
        self.attach_partial = torch.zeros((4 * ((P + 255) // 256),), **f)
        self._attach_n = 0
        if refill:
            self.reserve(self._spare_rows)
        return stats

    def _temp_points_attach(self, temp_xyz, temp_opacity, stable_mask, unstable_opacity_low, temp_obj=None):
        """mapper.py:1384-1436 on this mapper's map: indices (into the temp points) that fall onto the stable cloud's surfaces.
        temp_obj (the per-object job): the stable cloud is rendered through the object gate — every pixel shows its owner object's
        stable Gaussians, exactly what a shard that owns the object renders there — and a candidate only attaches to a stable Gaussian of
        its own object; the zero fill of never-rendered tiles counts as no hit (the reference's alias of such a pixel to "Gaussian 0"
        would name a different Gaussian on every shard layout)."""
        import dqo_mapgrowth as mg
        from . import mapping
        st, dev = self.settings, self.device
        sm = stable_mask.to(dev).bool().reshape(-1)
        if self.alive is not None:
            sm = sm & self.alive.bool()
        if not self._act_valid:
            N.check(N.lib().dqo_map_activate(self.P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                             N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), N.current_stream()))
            self._act_valid = True
        # the stable cloud alone = the map with every other Gaussian parked behind the camera (culled before the binning, so the tile
        # lists are the stable cloud's), indices in map rows.  One quirk to carry over: a tile that renders nothing keeps the op's
        # zero fill, which the reference's `>= 0` test reads as a hit on Gaussian 0 OF THE STABLE CLOUD (F3 / rasterize_points.cu:79-89)
        # — here that is the first stable row, and such a pixel is told from a real hit on row 0 by its zero weight.
        gated = temp_obj is not None and self.gaussian_object is not None
        if not gated:
            if not bool(sm.any()):
                return torch.empty((0,), dtype=torch.long, device=dev)
            first = torch.argmax(sm.to(torch.uint8)).reshape(1)  # (the first stable row)
        data = dict(xyz=torch.where(sm[:, None], self.xyz, self._park_position()[None, :]), opacity=self.opacity, scales=self.scales,
                    rotations=self.rotations, shs=self.shs)
        H, W = int(st.image_height), int(st.image_width)
        K = mg.const_tensor([[W / (2.0 * st.tanfovx), 0.0, st.cx], [0.0, H / (2.0 * st.tanfovy), st.cy], [0.0, 0.0, 1.0]], dev)
        if gated:
            # Only the candidates' own pixels are ever looked at (one pixel in twenty of a frame): the render goes through the object
            # gate with every OTHER pixel ownerless — such a pixel starts finished, an entry that reaches no owned pixel of a quadrant is
            # dropped by the quadrant's owner set, a (Gaussian, tile) pair whose object owns no candidate pixel of the tile is dropped by
            # the binning, a quadrant without a candidate ends at once; the owned pixels blend exactly what the full-frame gated
            # render blends for them (pixels are independent).  Two launches around the render (csrc/map_attach.hip) instead of the
            # reference's chain of boolean-index ops: the growth step is bound by the host's op issue rate.
            import diff_gaussian_rasterization_depth as dgr
            lin, sparse, tile_sets = mg.attach_pixels(temp_xyz, st.viewmatrix, W / (2.0 * st.tanfovx), H / (2.0 * st.tanfovy), st.cx, st.cy,
                                                      W, H, self.pixel_object)
            dgr.gate_ids_checked(sparse)  # (values of pixel_object, which has been checked, or -1)
            out = mapping.render(st, data, object_gate=(self.gaussian_object, sparse, tile_sets))
            ok = mg.attach_decide(temp_xyz, temp_opacity, temp_obj, lin, out["color_index_map"], out["color_hit_weight"], self.xyz,
                                  self.scaling_raw, self.rotation_raw, self.gaussian_object, self.add_depth_thres, unstable_opacity_low)
            return ok.nonzero().reshape(-1)
        out = mapping.render(st, data, object_gate=None)
        cim = out["color_index_map"]
        zero_fill = (cim == 0) & (out["color_hit_weight"] == 0)
        cim = torch.where(zero_fill, first.to(cim.dtype).reshape(1, 1, 1), cim)
        return mg.temp_points_attach_indices(temp_xyz, temp_opacity, st.viewmatrix.T.contiguous(), K, W, H, cim, self.xyz,
                                             lambda rows: self.normals(rows), self.add_depth_thres, unstable_opacity_low)

    def attach_loss(self):
        """The reference's reported "scale_loss" of the most recent iteration (attach loss at its pre-update parameters)."""
        return self.attach_partial[:self._attach_n].sum()

    def _attach_fields(self):
        if not self.use_attach:
            return dict(attach_mask=None, init_xyz=None, init_scaling_raw=None, init_rotation_raw=None, attach_count=0, attach_partial=None)
        # (attach_gains: the set's size is read from device memory when the launch runs — an empty set costs the mask read)
        return dict(attach_mask=N.ptr(self.attach_mask), init_xyz=N.ptr(self.init_xyz), init_scaling_raw=N.ptr(self.init_scaling),
                    init_rotation_raw=N.ptr(self.init_rotation), attach_count=self.attach_count, attach_partial=N.ptr(self.attach_partial),
                    attach_gains=N.ptr(self.attach_gains))

    @staticmethod
    def pick_list_split_pair(list_split, tile_mask, st, longest=0):
        """(forward, backward) values of DqoRastCtx.list_split for a capture: an int for both kernels, a pair (f, b) as given (b must be
        0 or f: the backward walks the forward's queue), or "auto" by the number of tiles the frame renders and the longest list of the
        state the capture is taken on.  Four waves per tile fill the 1024 SIMDs six deep at 1500 tiles; below that the blend kernels'
        time is the time of their longest lists, and the fewer tiles there are, the shorter the lists worth sharing between eight waves
        (measured on the shards of configs 4 and 5).  The eight-wave blocks cost the short lists occupancy (config 3's
        shards, whose lists all stay below 600 entries, lose 10 %), so a split is only taken when some list is long enough to be a tail
        worth cutting; and the fuller the GPU, the less the BACKWARD gains — it has no early exit, so its tail is the whole list's
        work, not a handful of quadrants' — so frames of more than 1300 tiles share lists in the forward only (config 5: half the
        frame 1.12 -> 0.96 ms, the full frame 1.69 -> 1.66 ms)."""
        if isinstance(list_split, (tuple, list)):
            f, b = int(list_split[0]), int(list_split[1])
            if f < 0 or b not in (0, f):
                raise ValueError("list_split pair: (forward threshold, 0 or the same threshold)")
            return f, b
        if list_split != "auto":
            if int(list_split) < 0:
                raise ValueError("list_split is 0 (off), a list length, a pair or 'auto'")
            return int(list_split), int(list_split)
        tiles = int((tile_mask != 0).sum().item()) if tile_mask is not None else ((st.image_width + 15) // 16) * ((st.image_height + 15) // 16)
        if tiles <= 1300:
            t = 256 if tiles <= 500 else 512 if tiles <= 800 else 1024
            return (t, t) if longest >= 1024 else (0, 0)
        if tiles <= 2600:
            return (1024, 0) if longest >= 2048 else (0, 0)
        return (2048, 0) if longest >= 4096 else (0, 0)

    @staticmethod
    def pick_list_split(list_split, tile_mask, st, longest=0):
        """The threshold BOTH blend kernels share under pick_list_split_pair's rule (0 when only the forward splits)."""
        return FusedMapper.pick_list_split_pair(list_split, tile_mask, st, longest)[1]

    # ------------------------------------------------------------------ hipGraph path ------------------------------------
    def capture(self, gt_color, gt_depth, render_mask, tile_mask=None, capacity_margin=1.15, tile_buckets=True, keep_tile_order=True,
                loss_tap=True, reuse_probe=False, fused_tail=True, list_split=0, unroll=1, settings=None, pixel_object=None, frame=None,
                run_unroll=1,
                short_bucket_margin=1.3):
        """Allocate persistent buffers for every intermediate of an iteration, run it once eagerly, then capture it into a
        hipGraph.  The inputs (gt images, masks) are read from the tensors passed here at every replay().

        settings / pixel_object / frame (round 6, the frame set of a mapping call — capture_window drives them): the camera of THIS
        graph (default: the mapper's), its pixel -> owner map when the object gate is on (default: set_object_gate's), and the slot of
        the window it fills (None = a single-frame mapper: the graph replaces whatever was captured before).  Every graph has its own
        context buffers, capacities and outputs; parameters, moments, activations, the device-side step count, the row flags, the
        confidence counter and the learning-rate table are the mapper's and shared.

        The capture fixes two capacities from the state it is taken on: the instance capacity (candidates x capacity_margin)
        and, with tile_buckets, the per-tile list bucket (twice the longest list, power of two).  A replay that outgrows
        either leaves invalid outputs and raises the device-side overflow flag — check graph_overflowed() (one small D2H read,
        e.g. once per batch of replays) and call capture() again when it is set, or use run(), which does both; tile_buckets=False
        keeps the packed lists (any list length, two more kernels per iteration).

        keep_tile_order (bucket mode): the replays keep the tile launch order of the capture's eager iteration instead of
        recomputing it (DqoRastCtx.keep_tile_order: no scan kernel in a replay).  loss_tap: the masked loss is summed inside the
        forward's blend kernel and its gradient is formed inside the backward's (DqoRastCtx.loss_tap: no loss kernels; self.loss is
        written by the backward).  fused_tail: the per-Gaussian half of the backward and the Adam step run as ONE kernel
        (dqo_rast_backward_adam: record sum -> per-Gaussian chain -> Adam per block of 256 Gaussians, gradient rows in LDS) instead of
        three (gradient rows and summed records through HBM).  All three leave every parameter and moment bit for bit as it is without
        them.  reuse_probe: size the capacities from
        the previous capture's counts (scaled by the map's growth) instead of a probing forward — for a re-capture right after a small
        change of the map; falls back to probing if the eager iteration overflows.  unroll: iterations per graph — replay() then runs
        `unroll` iterations with one launch call (back-to-back graph launches leave the GPU idle for ~9 us each on MI355X; the
        iterations inside one graph follow each other without a gap); self._g.out / self.loss then show the last of them."""
        lib = N.lib()
        dev, P, M = self.device, self.P, self.M
        st = self.settings if settings is None else _normalised_settings(settings, dev)
        if settings is not None:  # (the graph's own copies: set_frame rewrites them in place)
            st = st._replace(bg=st.bg.clone(), viewmatrix=st.viewmatrix.clone(), projmatrix=st.projmatrix.clone(), campos=st.campos.clone())
        H, W = int(st.image_height), int(st.image_width)
        if (H, W) != (int(self.settings.image_height), int(self.settings.image_width)):
            raise RuntimeError("FusedMapper.capture: every frame of a mapper has the mapper's image size")
        f = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        tile_mask = self.tile_mask if tile_mask is None else _checked_tile_mask(tile_mask, dev, H, W)
        call_kw = dict(tile_mask=tile_mask, capacity_margin=capacity_margin, tile_buckets=tile_buckets, keep_tile_order=keep_tile_order,
                       loss_tap=loss_tap, fused_tail=fused_tail, list_split=list_split, unroll=unroll, run_unroll=run_unroll, settings=settings,
                       pixel_object=pixel_object, frame=frame, short_bucket_margin=short_bucket_margin)
        pix_obj, tile_obj = self.pixel_object, self.tile_objects
        if pixel_object is not None:
            if self.gaussian_object is None:
                raise RuntimeError("FusedMapper.capture: a per-frame pixel_object needs set_object_gate first")
            pix_obj = torch.as_tensor(pixel_object).to(dev, torch.int32).contiguous().reshape(H, W)
            tile_obj = tile_object_sets(pix_obj)
        with torch.cuda.device(dev):
            # re-capture (e.g. after an overflow): replays of invalid frames did not advance the device-side step count
            # (DqoAdamStep.frame_header), the host count assumed they did — _settle_replays takes exactly those back; eager step()
            # calls since then advanced the host count alone (they never touch the device count) and stay counted
            self._settle_replays()
            if frame is None:
                self._g, self._frames, self._mixed = None, [], {}
            if not self._act_valid:
                N.check(lib.dqo_map_activate(P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                             N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), N.current_stream()))
                self._act_valid = True
            # capacity: the reference's num_rendered of the current state (an upper bound of the instances kept) plus a margin
            # for the Gaussians that move while the graph is being replayed; the device header flags an overflow.  The probe runs
            # through the C ABI on buffers of its own (two header reads), so the operator's global state is not touched.
            # (reuse_probe: the counts of the previous capture, scaled by the growth of the map since — for a re-capture right after a
            # growth step, which changes the map by a fraction of a percent; if the eager iteration below overflows, the capture
            # is redone with a real probe)
            last = getattr(self, "_last_probe", None)
            if reuse_probe and last is not None and last[2] > 0:
                grown = max(1.0, P / last[2])
                cand, longest = int(last[0] * grown) + 1, int(last[1] * grown) + 1
            else:
                reuse_probe = False
                cand, longest = self._probe(tile_mask, st, pix_obj, tile_obj)
            self._last_probe = (cand, longest, P)
            cap = int(cand * capacity_margin) + 4096
            g = self._g = type("G", (), {})()
            g.cap = cap
            g.settings, g.pixel_object, g.tile_objects = st, pix_obj, tile_obj
            g.frame = frame
            if isinstance(frame, tuple):  # (camera / target of frame k, masks of frame m): global_optimization's second half
                self._mixed[frame] = g
            elif frame is not None:
                while len(self._frames) <= frame:
                    self._frames.append(None)
                self._frames[frame] = g
            # fixed per-tile list buckets (DqoRastCtx.tile_bucket_capacity): at least twice the longest list of the current state,
            # a power of two; a tile that outgrows it raises the same overflow flag as running out of instance capacity
            g.bucket = 0
            if tile_buckets:
                g.bucket = 256
                while g.bucket < 2 * longest:
                    g.bucket *= 2
                # 1024 entries are what the per-tile sort launch reaches: while no list can be longer, the long-list sort launch is
                # dropped (6 us of launch on a map whose lists are all short) — worth a smaller margin (short_bucket_margin, default
                # 1.3 x the longest list; pass 2.0 to never trade margin for that launch) to stay there.  A tile that outgrows its
                # bucket flags the frame (graph_overflowed(); run() / run_window() re-capture; the replays in between train nothing).
                if g.bucket == 2048 and float(short_bucket_margin) * longest <= 1024:
                    g.bucket = 1024
            for name, t_, shape in (("gt_color", gt_color, (3, H, W)), ("gt_depth", gt_depth, (1, H, W))):
                if t_.dtype != torch.float32 or not t_.is_cuda or not t_.is_contiguous() or tuple(t_.shape) != shape:
                    raise RuntimeError(f"FusedMapper.capture: {name} must be a contiguous float32 GPU tensor of shape {shape} "
                                       "(the graph reads it in place at every replay)")
            g.gt_color, g.gt_depth = gt_color, gt_depth
            g.mask = None if render_mask is None else render_mask.to(torch.uint8).contiguous()
            g.tile_mask = tile_mask
            g.stale = False
            g.attach_in_place = True  # attach set, init_stat and its size are read from device memory at every replay
            g.fused_tail = bool(fused_tail) and M <= 16
            g.out = (torch.empty((3, H, W), **f), torch.empty((1, H, W), **f), torch.empty((1, H, W), **i32), torch.empty((1, H, W), **i32),
                     torch.empty((1, H, W), **f), torch.empty((1, H, W), **f), torch.empty((1, H, W), **f), torch.empty((P,), **i32),
                     torch.empty((P,), **i32))
            g.geom = torch.empty((lib.dqo_rast_geom_bytes(P, W, H),), **u8)
            g.img = torch.empty((lib.dqo_rast_image_bytes(W, H),), **u8)
            g.binning = torch.empty((lib.dqo_rast_binning_bytes_bucketed(cap, W, H, g.bucket),), **u8)
            g.ws = torch.empty((lib.dqo_rast_backward_workspace_bytes(cap),), **u8)
            # dL_dcolors / dL_dcov3D / dL_dmeans2D have no consumer in the mapping step: NULL = the backward does not store them
            g.grads = dict(means3D=torch.empty((P, 3), **f), sh=torch.empty((P, M, 3), **f),
                           opacity=torch.empty((P, 1), **f), scales=torch.empty((P, 3), **f), rot=torch.empty((P, 4), **f))
            # the step state is the mapper's, shared by all its graphs (DqoAdamStep.step_dev: the Adam launch advances it itself through
            # block_ticket; bias_table: the bias corrections, computed once per step)
            g.step_dev, g.ticket, g.bias = self._step_dev, self._ticket, self._bias
            self._step_dev.fill_(self.step_count + 1)
            self._expected_step = self.step_count + 1
            g.params = dgr._params(st, P, M)
            g.inputs = dgr._inputs(st, self.xyz, self.shs, self._empty, self.opacity, self.scales, self.rotations, self._empty, g.tile_mask,
                                   row_flags=self.row_flags)
            o = g.out
            g.outputs = N.DqoRastOutputs(out_color=o[0].data_ptr(), out_depth=o[1].data_ptr(), out_hit_color=o[2].data_ptr(),
                                         out_hit_depth=o[3].data_ptr(), out_hit_color_weight=o[4].data_ptr(),
                                         out_hit_depth_weight=o[5].data_ptr(), out_T=o[6].data_ptr(), n_touched=o[7].data_ptr(),
                                         radii=o[8].data_ptr())
            g.cctx = N.DqoRastCtx(geom=g.geom.data_ptr(), geom_bytes=g.geom.numel(), binning=g.binning.data_ptr(),
                                  binning_bytes=g.binning.numel(), image=g.img.data_ptr(), image_bytes=g.img.numel(), inst_capacity=cap,
                                  tile_bucket_capacity=g.bucket)
            # (DqoRastCtx.list_split is read by the forward and by the backward call: _static_iteration sets it before each)
            g.ls_fwd, g.ls_bwd = self.pick_list_split_pair(list_split, g.tile_mask, st, longest)
            g.cctx.list_split = g.ls_fwd
            g.list_split = list_split
            # DqoLossTap: the masked loss is summed by the forward's blend kernel and its gradient images are formed inside the
            # backward's (bit for bit what dqo_map_loss_fwd_bwd computes): no loss kernels, no passes over the full image
            g.tap = None
            if g.mask is None and self.ssim_weight != 0:
                # the SSIM term's gradient is an image (an 11 x 11 window around every pixel): the tap, which forms sign(error) x weight
                # inside the backward's blend kernel, cannot carry it -> loss kernels + gradient images (three more launches + the SSIM's three)
                loss_tap = False
            if loss_tap:
                g.grad_scale = torch.zeros((2,), **f)
                g.tap = N.DqoLossTap(gt_color=N.ptr(gt_color), gt_depth=N.ptr(gt_depth), render_mask=N.ptr(g.mask), out_color=o[0].data_ptr(),
                                     out_depth=o[1].data_ptr(), color_weight=self.color_weight, depth_weight=self.depth_weight,
                                     add_depth_thres=self.add_depth_thres, loss_out=N.ptr(self.loss), grad_scale=g.grad_scale.data_ptr(),
                                     per_object=1 if (self.per_object_loss and self.gaussian_object is not None) else 0)
                g.cctx.loss_tap = ctypes.addressof(g.tap)
            g.gate = None
            if self.gaussian_object is not None:
                g.gate = N.DqoObjectGate(gaussian_object=N.ptr(self.gaussian_object), pixel_object=N.ptr(pix_obj), tile_objects=N.ptr(tile_obj))
                g.cctx.object_gate = ctypes.addressof(g.gate)
                if self.per_object_loss and not loss_tap:
                    raise RuntimeError("FusedMapper.capture: the per-object loss is computed by the loss tap (loss_tap=True)")
            gr = g.grads
            g.cgrads = N.DqoRastGrads(dL_dmeans3D=gr["means3D"].data_ptr(), dL_dsh=gr["sh"].data_ptr(), dL_dcolors=None,
                                      dL_dopacity=gr["opacity"].data_ptr(), dL_dscales=gr["scales"].data_ptr(),
                                      dL_drotations=gr["rot"].data_ptr(), dL_dcov3D=None, dL_dmeans2D=None, skip_culled_rows=1)
            stt = self.state
            g.adam = N.DqoAdamStep(P=P, M=M, step=0, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, lr_xyz=self.lrs["xyz"],
                                   lr_f_dc=self.lrs["f_dc"], lr_f_rest=self.lrs["f_rest"], lr_opacity=self.lrs["opacity"],
                                   lr_scaling=self.lrs["scaling"], lr_rotation=self.lrs["rotation"], xyz=N.ptr(self.xyz), shs=N.ptr(self.shs),
                                   opacity_raw=N.ptr(self.opacity_raw), scaling_raw=N.ptr(self.scaling_raw),
                                   rotation_raw=N.ptr(self.rotation_raw), g_means3D=gr["means3D"].data_ptr(), g_sh=gr["sh"].data_ptr(),
                                   g_opacity=gr["opacity"].data_ptr(), g_scales=gr["scales"].data_ptr(), g_rotations=gr["rot"].data_ptr(),
                                   m_xyz=N.ptr(stt["xyz"][0]), m_shs=N.ptr(stt["shs"][0]), m_opacity=N.ptr(stt["opacity"][0]),
                                   m_scaling=N.ptr(stt["scaling"][0]), m_rotation=N.ptr(stt["rotation"][0]), v_xyz=N.ptr(stt["xyz"][1]),
                                   v_shs=N.ptr(stt["shs"][1]), v_opacity=N.ptr(stt["opacity"][1]), v_scaling=N.ptr(stt["scaling"][1]),
                                   v_rotation=N.ptr(stt["rotation"][1]), act_opacity=N.ptr(self.opacity), act_scales=N.ptr(self.scales),
                                   act_rotations=N.ptr(self.rotations), radii=o[8].data_ptr(), step_dev=g.step_dev.data_ptr(),
                                   moment_live=N.ptr(self.moment_live), frame_header=g.geom.data_ptr(),
                                   block_ticket=g.ticket.data_ptr() if self.use_block_ticket else None,
                                   bias_table=g.bias.data_ptr() if self.use_block_ticket else None, row_flags=N.ptr(self.row_flags),
                                   confidence=N.ptr(self.confidence) if self.count_confidence else None, lr_table=N.ptr(self.lr_table),
                                   **self._attach_fields())
            # one eager iteration on a side stream (warms every kernel up), then the capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._static_iteration()
            torch.cuda.current_stream().wait_stream(side)
            if self.graph_overflowed() and reuse_probe:  # the reused counts were too small after all: measure and start over
                self._last_probe = None
                return self.capture(gt_color, gt_depth, render_mask, reuse_probe=False, **call_kw)
            if not self.graph_overflowed():  # (an invalid frame is a no-op for the optimiser and its step count)
                self.step_count += 1
            # the eager iteration left its tile launch order in g.img; the replays keep it (DqoRastCtx.keep_tile_order: the order is
            # a scheduling hint, and the lists of one camera change little between the iterations of a mapping call)
            g.cctx.keep_tile_order = 1 if (g.bucket > 0 and keep_tile_order) else 0
            # ... and start from the counters the previous iteration's per-Gaussian kernel cleared (DqoRastCtx.frame_prezeroed: no
            # zero-fill launch in a replay; only dqo_rast_backward_adam clears them, so only with the fused tail)
            g.cctx.frame_prezeroed = 1 if g.fused_tail else 0
            g.graph = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (e.g. a collective library's watchdog) may keep issuing runtime calls
            g.unroll = max(1, int(unroll))
            with torch.cuda.graph(g.graph, capture_error_mode="thread_local"):
                for _ in range(g.unroll):
                    self._static_iteration()
            # run_unroll (capture_window): a SECOND graph over the same context buffers with that many iterations, for the stretches of a
            # schedule that stay on one frame — the second half of local_optimize renders the newest frame only (mapper.py:574-576) —
            # where one launch per iteration leaves the GPU idle for ~9 us between graphs (replay_run)
            g.run_graph, g.run_unroll = None, max(1, int(run_unroll))
            if g.run_unroll > 1:
                g.run_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g.run_graph, capture_error_mode="thread_local"):
                    for _ in range(g.run_unroll):
                        self._static_iteration()
            self._expected_step = self.step_count + 1
        return self

    # ------------------------------------------------------------------ the frame set of a mapping call -----------------
    def _snapshot_state(self):
        return dict(params={k: v.clone() for k, v in self._params().items()},
                    state={k: (m.clone(), v.clone()) for k, (m, v) in self.state.items()},
                    live=None if self.moment_live is None else self.moment_live.clone(), step=self.step_count,
                    conf=self.confidence.clone())

    def _restore_state(self, snap):
        self._settle_replays()
        for k, v in self._params().items():
            v.copy_(snap["params"][k])
        for k, (m, v) in self.state.items():
            m.copy_(snap["state"][k][0]), v.copy_(snap["state"][k][1])
        if self.moment_live is not None:
            self.moment_live.copy_(snap["live"])
        self.confidence.copy_(snap["conf"])
        self.step_count = snap["step"]
        self._step_dev.fill_(self.step_count + 1)
        self._expected_step, self._unsettled = self.step_count + 1, False
        self._bias.zero_()
        self._act_valid = False
        self.activate()

    def capture_window(self, frames, **kw):
        """One captured graph per frame of a mapping call's frame set — the five `processed_frames` of local_optimize (mapper.py:549-555)
        or the selected keyframes of global_optimization (:1159-1180).  frames: list of dict(gt_color, gt_depth, render_mask=None,
        tile_mask=None, settings=None, pixel_object=None); kw: capture()'s other arguments.  Each graph owns its context buffers and
        capacities (a frame's tail clears ITS counters for ITS next replay: DqoRastCtx.frame_prezeroed holds per frame); they share the
        map, the optimiser state and the device step count, so replay(k) in any order is the reference's loop with its per-iteration
        frame choice = one graph launch.  The eager iterations the captures run are undone (parameters, moments, confidence, step
        count are put back): the mapping call starts from the state it was given."""
        snap = self._snapshot_state()
        self._settle_replays()
        self._g, self._frames, self._mixed = None, [], {}
        self._window_kw, self._window_frames = dict(kw), list(frames)
        for k, fr in enumerate(frames):
            if k:
                self._restore_state(snap)  # every frame is sized on the call's initial state
            self.capture(fr["gt_color"], fr["gt_depth"], fr.get("render_mask"), tile_mask=fr.get("tile_mask"), settings=fr.get("settings"),
                         pixel_object=fr.get("pixel_object"), frame=k, **kw)
            torch.cuda.synchronize()
        self._restore_state(snap)
        return self

    @staticmethod
    def window_schedule(n_iters, n_frames, rng, final=False, random_keyframes=False, global_opt=False):
        """The per-iteration frame choice of the reference's loops as a list of frame indices:
          local_optimize (mapper.py:570-576):   random_index = random.randint(0, len - 1);  if iter > n / 2: random_index = -1
          global_optimization (:1186-1199):     the same draw — but the camera and the target images are taken BEFORE the index is
                                                overwritten (`frame_input = select_frame[random_index]` at :1188-1190, the `= -1` at
                                                :1194-1195), only the tile mask and the render mask follow it: in the second half of a
                                                (non-final) call a RANDOM keyframe is rendered and compared under the LAST keyframe's
                                                masks.  global_opt=True reproduces that: those entries are pairs (k, n_frames - 1) —
                                                replay() / run_window() capture such a mixed graph on first use.
        rng: a random.Random (the reference draws from the global `random` module).  Index -1 = the last frame of the set (the newest
        processed frame / select_frame[-1]); final = the `is_final` pass (random throughout)."""
        out = []
        for it in range(int(n_iters)):
            k = rng.randint(0, n_frames - 1)
            if it > n_iters / 2 and not final and not random_keyframes:
                k = (k, n_frames - 1) if (global_opt and k != n_frames - 1) else n_frames - 1
            out.append(k)
        return out

    def _graph_of(self, frame):
        """The captured graph of a schedule entry: a frame index, or a pair (k, m) = frame k's camera and target under frame m's render
        mask and tile mask (captured on first use from the window's frames, state untouched)."""
        if not isinstance(frame, tuple):
            return self._frames[frame]
        if frame not in self._mixed:
            k, m = frame
            fk, fm_ = self._window_frames[k], self._window_frames[m]
            snap = self._snapshot_state()
            cur = self._g
            self.capture(fk["gt_color"], fk["gt_depth"], fm_.get("render_mask"), tile_mask=fm_.get("tile_mask"), settings=fk.get("settings"),
                         pixel_object=fk.get("pixel_object"), frame=frame, **self._window_kw)
            self._restore_state(snap)
            self._g = cur
        return self._mixed[frame]

    def run_window(self, schedule, check_every=64, capacity_margin=1.5):
        """The iterations of `schedule` (frame indices, see window_schedule) on the captured frame set: one graph launch each, one small
        D2H read per `check_every` launches (device step count).  A frame that outgrew its captured capacities made its iterations
        no-ops for the optimiser (parameters, moments, confidence, step count untouched): that frame is captured again on the current
        state with `capacity_margin` and the lost iterations are replayed on it at the end of the batch.  Returns the re-captures."""
        recaptures, i = 0, 0
        while i < len(schedule):
            batch = schedule[i:i + check_every]
            start = self.step_count
            self.replay_schedule(batch)
            lost = (start + len(batch)) - (int(self._step_dev.item()) - 1)
            if lost > 0:
                self._settle_replays()
                bad = [k for k in sorted(set(batch), key=str) if self.graph_overflowed(self._graph_of(k))]
                if not bad or recaptures > 8 * max(1, len(self._frames)):
                    raise RuntimeError("FusedMapper.run_window: the map keeps outgrowing the captured capacities")
                for k in bad:
                    g = self._graph_of(k)
                    snap = self._snapshot_state()
                    self.capture(g.gt_color, g.gt_depth, g.mask, tile_mask=g.tile_mask, capacity_margin=capacity_margin,
                                 tile_buckets=g.bucket > 0, keep_tile_order=bool(g.cctx.keep_tile_order) or g.bucket > 0,
                                 loss_tap=g.tap is not None, fused_tail=g.fused_tail, list_split=g.list_split, unroll=g.unroll,
                                 run_unroll=getattr(g, "run_unroll", 1),
                                 settings=g.settings, pixel_object=g.pixel_object if self.gaussian_object is not None else None, frame=k)
                    self._restore_state(snap)
                    recaptures += 1
                for j in range(lost):
                    self.replay(frame=bad[j % len(bad)])
                torch.cuda.synchronize()
            i += len(batch)
        return recaptures

    @torch.no_grad()
    def set_frame(self, frame, gt_color=None, gt_depth=None, render_mask=None, tile_mask=None, settings=None, pixel_object=None):
        """New inputs for a captured frame WITHOUT a new capture: the next mapping call's window keeps four of its five frames and every
        call re-evaluates the masks (mapper.py:549-555).  Everything a graph reads per frame lives in device buffers the capture fixed:
        the camera's matrices / position / background, the ground-truth images, the render mask, the tile mask, the owner map — they
        are overwritten in place (same shapes; a frame captured without a render mask cannot get one later, and the scalar intrinsics
        — image size, tan(fov), principal point, thresholds — are kernel arguments: they must not change).  The captured capacities
        were sized on the old view: check graph_overflowed(frame) / use run_window, which re-captures a frame that outgrew them."""
        cams = [self._frames[frame]] + [g for (k, m), g in self._mixed.items() if k == frame]    # graphs that render frame's camera / target
        masks = [self._frames[frame]] + [g for (k, m), g in self._mixed.items() if m == frame]   # graphs that use frame's masks
        done = set()

        def once(t):  # (graphs of one window share the tensors they were given: write each buffer once)
            if t.data_ptr() in done:
                return False
            done.add(t.data_ptr())
            return True

        for g in cams:
            if gt_color is not None and once(g.gt_color):
                g.gt_color.copy_(gt_color)
            if gt_depth is not None and once(g.gt_depth):
                g.gt_depth.copy_(gt_depth)
        for g in masks:
            if render_mask is not None:
                if g.mask is None:
                    raise RuntimeError("FusedMapper.set_frame: the frame was captured without a render mask")
                if once(g.mask):
                    g.mask.copy_(render_mask.to(torch.uint8))
            if tile_mask is not None and once(g.tile_mask):
                g.tile_mask.copy_(_checked_tile_mask(tile_mask, self.device, int(g.settings.image_height), int(g.settings.image_width)))
        if settings is not None:
            new = _normalised_settings(settings, self.device)
            for g in cams:
                for name in ("image_height", "image_width", "tanfovx", "tanfovy", "cx", "cy", "sh_degree", "scale_modifier", "opaque_threshold",
                             "depth_threshold", "normal_threshold", "color_sigma", "T_threshold"):
                    if getattr(new, name) != getattr(g.settings, name):
                        raise RuntimeError(f"FusedMapper.set_frame: {name} is a captured kernel argument; capture the frame again")
                for name in ("bg", "viewmatrix", "projmatrix", "campos"):
                    getattr(g.settings, name).copy_(getattr(new, name))
        if pixel_object is not None:
            for g in cams:
                if g.pixel_object is None:
                    raise RuntimeError("FusedMapper.set_frame: the frame was captured without an object gate")
                if once(g.pixel_object):
                    g.pixel_object.copy_(torch.as_tensor(pixel_object).to(self.device, torch.int32).reshape(g.pixel_object.shape))
                    g.tile_objects.copy_(tile_object_sets(g.pixel_object))
        return self

    def capture_placed(self, *args, trials=4, probe_replays=12, **kw):
        """capture() on the best of `trials` PLACEMENTS of the context buffers.  Where the allocator puts the geometry / binning / image
        buffers in physical memory moves the binning kernel by +-5 us and the per-Gaussian tail by +-3 us on cfg 3 — same code, same
        virtual layout, another set of pages (tools/bin_count_place.py: 52-62 us between the buffer sets of ONE process; the device
        atomics and partial-line writes of those kernels meet the memory channels differently).  So: capture `trials` times, every time
        on freshly allocated buffers while the earlier sets are still held (the allocator must place the next set elsewhere), time
        `probe_replays` replays of each, keep the fastest graph and hand the other buffers back.  The optimisation state is put back
        in between and at the end — parameters, moments, step count: the iterations of the trials never happened; what remains is
        exactly capture()'s own eager iteration, re-run on the kept buffers.  Results do not depend on the placement (bit for bit)."""
        if int(trials) <= 1:
            return self.capture(*args, **kw)
        self._settle_replays()  # (the snapshot's step count is the settled one)
        snap = self._snapshot_state()

        def restore():
            self._restore_state(snap)

        cands = []
        for t in range(int(trials)):
            if t:
                restore()
                self._g = None  # (the earlier sets stay referenced by `cands`: this capture's buffers land elsewhere)
            self.capture(*args, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                self.replay()
            e0.record()
            for _ in range(int(probe_replays)):
                self.replay()
            e1.record()
            torch.cuda.synchronize()
            self._settle_replays()
            if self.graph_overflowed():  # (an overflowed trial replays no-ops: fast, and worthless)
                continue
            cands.append((e0.elapsed_time(e1) / probe_replays / self._g.unroll, self._g))
        if not cands:
            raise RuntimeError("FusedMapper.capture_placed: every trial overflowed its captured capacities")
        best = min(range(len(cands)), key=lambda i: cands[i][0])
        self.placement_trials_ms = [round(c[0], 4) for c in cands]
        restore()
        self._g = cands[best][1]
        if self._g.frame is not None:
            self._frames[self._g.frame] = self._g
        del cands
        self.step_static()    # capture()'s eager iteration, on the kept buffers
        torch.cuda.synchronize()
        return self

    def _probe(self, tile_mask, st=None, pixel_object=None, tile_objects=None):
        """(num_candidates, longest tile list) of the current state: one forward on scratch buffers with packed lists."""
        lib, dev, P, M = N.lib(), self.device, self.P, self.M
        st = self.settings if st is None else st
        if pixel_object is None:
            pixel_object, tile_objects = self.pixel_object, self.tile_objects
        H, W = int(st.image_height), int(st.image_width)
        f, i32, u8 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev), dict(dtype=torch.uint8, device=dev)
        out = [torch.empty((3, H, W), **f), torch.empty((1, H, W), **f), torch.empty((1, H, W), **i32), torch.empty((1, H, W), **i32),
               torch.empty((1, H, W), **f), torch.empty((1, H, W), **f), torch.empty((1, H, W), **f), torch.empty((P,), **i32),
               torch.empty((P,), **i32)]
        geom = torch.empty((lib.dqo_rast_geom_bytes(P, W, H),), **u8)
        img = torch.empty((lib.dqo_rast_image_bytes(W, H),), **u8)
        params = dgr._params(st, P, M)
        inputs = dgr._inputs(st, self.xyz, self.shs, self._empty, self.opacity, self.scales, self.rotations, self._empty, tile_mask,
                             row_flags=self.row_flags)
        outputs = N.DqoRastOutputs(out_color=out[0].data_ptr(), out_depth=out[1].data_ptr(), out_hit_color=out[2].data_ptr(),
                                   out_hit_depth=out[3].data_ptr(), out_hit_color_weight=out[4].data_ptr(),
                                   out_hit_depth_weight=out[5].data_ptr(), out_T=out[6].data_ptr(), n_touched=out[7].data_ptr(),
                                   radii=out[8].data_ptr())
        cctx = N.DqoRastCtx(geom=geom.data_ptr(), geom_bytes=geom.numel(), binning=None, binning_bytes=0, image=img.data_ptr(),
                            image_bytes=img.numel(), inst_capacity=0)
        if self.gaussian_object is not None:
            # the lists of the gated job (the binning drops a pair whose object owns no pixel of the tile): its longest list, not the
            # ungated frame's, sizes the buckets
            gate = N.DqoObjectGate(gaussian_object=N.ptr(self.gaussian_object), pixel_object=N.ptr(pixel_object), tile_objects=N.ptr(tile_objects))
            cctx.object_gate = ctypes.addressof(gate)
        stream = N.current_stream()
        hdr = N.DqoRastHeader()
        N.check(lib.dqo_rast_forward_prepare(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(outputs), ctypes.byref(cctx), stream))
        N.check(lib.dqo_rast_read_header(ctypes.byref(cctx), ctypes.byref(hdr), stream))
        cand = int(hdr.num_candidates)
        binning = torch.empty((lib.dqo_rast_binning_bytes(max(cand, 1)),), **u8)
        cctx.binning, cctx.binning_bytes, cctx.inst_capacity = binning.data_ptr(), binning.numel(), max(cand, 1)
        N.check(lib.dqo_rast_forward_render(ctypes.byref(params), ctypes.byref(inputs), ctypes.byref(outputs), ctypes.byref(cctx), stream))
        N.check(lib.dqo_rast_read_header(ctypes.byref(cctx), ctypes.byref(hdr), stream))
        return cand, int(hdr.max_tile_count)

    def _static_iteration(self):
        """The five C-ABI calls of one iteration over the persistent buffers (no allocation, no host-side per-step state)."""
        lib, g = N.lib(), self._g
        st = self.settings
        H, W = int(st.image_height), int(st.image_width)
        stream = N.current_stream()
        g.cctx.list_split = g.ls_fwd
        N.check(lib.dqo_rast_forward(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.outputs), ctypes.byref(g.cctx), stream))
        g.cctx.list_split = g.ls_bwd  # (0 with a split forward: the single-wave backward walks every list, the queue is ignored)
        o = g.out
        if g.tap is None:
            N.check(lib.dqo_map_loss_fwd_bwd(W, H, o[0].data_ptr(), o[1].data_ptr(), o[3].data_ptr(), N.ptr(g.gt_color), N.ptr(g.gt_depth),
                                             N.ptr(g.mask), self.color_weight, self.depth_weight, self.add_depth_thres, N.ptr(self.loss),
                                             N.ptr(self.dL_dcolor), N.ptr(self.dL_ddepth), N.ptr(self.loss_ws), self.loss_ws.numel(), stream))
            if g.mask is None and self.ssim_weight != 0:
                self._ssim_term(o[0].data_ptr(), g.gt_color, stream)
        dLc, dLd = (self.dL_dcolor.data_ptr(), self.dL_ddepth.data_ptr()) if g.tap is None else (None, None)
        self._attach_n = ((self.P + 255) // 256) * (4 if g.fused_tail else 1)
        if g.fused_tail:
            N.check(lib.dqo_rast_backward_adam(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.cctx), dLc, dLd,
                                               ctypes.byref(g.adam), g.ws.data_ptr(), g.ws.numel(), stream))
            return
        N.check(lib.dqo_rast_backward(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.cctx), dLc, dLd, o[3].data_ptr(),
                                      ctypes.byref(g.cgrads), g.ws.data_ptr(), g.ws.numel(), stream))
        N.check(lib.dqo_map_adam_step(ctypes.byref(g.adam), stream))

    def _ssim_term(self, color_ptr, gt_color, stream):
        """Without a render mask the loss carries the SSIM term (mapper.py:839-845; with one the reference skips it, B14): its gradient
        is added onto the colour-gradient image dqo_map_loss_fwd_bwd just wrote, its value onto self.loss[0] (slot 3 = 1 - ssim)."""
        lib = N.lib()
        H, W = int(self.settings.image_height), int(self.settings.image_width)
        if self.ssim_ws is None:
            self.ssim_ws = torch.empty((lib.dqo_map_ssim_workspace_bytes(W, H),), dtype=torch.uint8, device=self.device)
            self.ssim_out = torch.zeros((2,), dtype=torch.float32, device=self.device)
        N.check(lib.dqo_map_ssim_fwd_bwd(W, H, color_ptr, N.ptr(gt_color), self.ssim_weight, N.ptr(self.ssim_out), N.ptr(self.dL_dcolor), 1,
                                         N.ptr(self.loss), N.ptr(self.ssim_ws), self.ssim_ws.numel(), stream))

    def replay(self, frame=None):
        """One mapping iteration (capture(unroll=k): k of them) by replaying a captured graph — `frame`: which one of capture_window's
        (None: the one replayed last / the single-frame mapper's); outputs are the persistent tensors in self._g.out."""
        if frame is not None:
            self._g = self._graph_of(frame)
        g = self._g
        if g.stale:
            raise RuntimeError("FusedMapper: the attach set / object gate changed since capture(); capture again")
        if self._expected_step != self.step_count + 1:  # eager step() calls in between: resynchronise the device-side step count
            self._resync_step_count()
        g.graph.replay()
        self._unsettled = True
        self._attach_n = ((self.P + 255) // 256) * (4 if g.fused_tail else 1)  # (a new mapping call in between had reset it)
        self.step_count += g.unroll  # (assumes valid frames; _settle_replays re-reads the device-side count)
        self._expected_step = self.step_count + 1
        return g.out

    def replay_run(self, frame, n):
        """n consecutive iterations on ONE frame of the window: launches of the frame's run graph (capture_window(run_unroll=k): k
        iterations each) while at least k remain, single launches for the rest — the same kernels in the same order as n replay(frame)
        calls, bit for bit.  Returns the outputs like replay()."""
        g = self._graph_of(frame)
        k = getattr(g, "run_unroll", 1)
        out = None
        while n >= k and k > 1:
            self._g = g
            if g.stale:
                raise RuntimeError("FusedMapper: the attach set / object gate changed since capture(); capture again")
            if self._expected_step != self.step_count + 1:
                self._resync_step_count()
            g.run_graph.replay()
            self._unsettled = True
            self._attach_n = ((self.P + 255) // 256) * (4 if g.fused_tail else 1)
            self.step_count += k
            self._expected_step = self.step_count + 1
            out = g.out
            n -= k
        for _ in range(n):
            out = self.replay(frame=frame)
        return out

    def replay_schedule(self, schedule):
        """The iterations of `schedule` (window_schedule) with every stretch on one frame handed to replay_run."""
        i = 0
        while i < len(schedule):
            j = i
            while j < len(schedule) and schedule[j] == schedule[i]:
                j += 1
            self.replay_run(schedule[i], j - i)
            i = j

    def _settle_replays(self):
        """Make the host step count agree with what the device counted.  replay() assumes a valid frame; the device count only advances
        on valid ones (DqoAdamStep.frame_header: an overflowed frame is a no-op for the optimiser), so what it falls short of the count
        expected after the last replay is the number of invalid replays — taken back here, BEFORE anything else (an eager step(), a
        re-capture, a resynchronisation) builds on the host count or overwrites the device count.  One 4-byte read, and only when
        replays happened since the last time."""
        if not getattr(self, "_unsettled", False):
            return
        dev_step = int(self._step_dev.item())
        invalid = int(self._expected_step) - dev_step
        if invalid > 0:
            self.step_count -= invalid
        self._expected_step = dev_step
        self._unsettled = False

    def _resync_step_count(self):
        """Eager step() calls between two replays advanced the host count alone: bring the device count up to it."""
        self._settle_replays()
        self._step_dev.fill_(self.step_count + 1)
        self._expected_step = self.step_count + 1

    def run(self, n_iters, check_every=64, capacity_margin=1.5):
        """`n_iters` VALID mapping iterations on the captured graph: replays in batches of `check_every`, one small D2H read per batch
        (device-side step count + overflow flag); if the map outgrew the captured capacities in a batch, the invalid replays were no-ops for the
        optimiser (DqoAdamStep.frame_header: parameters, moments and the device step count are untouched), so the graph is captured
        again on the current state with `capacity_margin` and the missing iterations are replayed.  Returns the number of
        re-captures.  The inputs of the last capture() (ground-truth images, masks) are reused."""
        g = self._g
        if g.unroll != 1:
            raise RuntimeError("FusedMapper.run counts single iterations: capture with unroll=1")
        target = self.step_count + n_iters
        recaptures = 0
        while self.step_count < target:
            k = min(check_every, target - self.step_count)
            for _ in range(k):
                self.replay()
            # the device-side step count only advances on valid frames: one 4-byte read tells whether ANY replay of the batch was invalid
            # (the header's flag alone would only tell about the last one)
            if int(self._step_dev.item()) - 1 != self.step_count or self.graph_overflowed():
                if recaptures > 8:
                    raise RuntimeError("FusedMapper.run: the map keeps outgrowing the captured capacities")
                g = self._g
                # (capture() re-reads the device-side step count: the valid replays of this batch stay counted, the others do not)
                self.capture(g.gt_color, g.gt_depth, g.mask, tile_mask=g.tile_mask, capacity_margin=capacity_margin,
                             tile_buckets=g.bucket > 0, keep_tile_order=bool(g.cctx.keep_tile_order) or g.bucket > 0, loss_tap=g.tap is not None,
                             fused_tail=g.fused_tail, list_split=g.list_split, run_unroll=getattr(g, "run_unroll", 1),
                             settings=g.settings if g.settings is not self.settings else None,
                             pixel_object=g.pixel_object if (self.gaussian_object is not None and g.pixel_object is not self.pixel_object) else None,
                             frame=g.frame)
                recaptures += 1
        return recaptures

    def step_static(self):
        """One iteration over the persistent buffers issued eagerly — exactly the calls the captured graph holds (for per-kernel
        profiling: events cannot be recorded inside a replay)."""
        g = self._g
        if self._expected_step != self.step_count + 1:
            self._resync_step_count()
        with torch.cuda.device(self.device):
            self._static_iteration()
        self._unsettled = True
        self.step_count += 1
        self._expected_step = self.step_count + 1
        return g.out

    def header(self, g=None):
        """Device header of the captured iteration's last forward (one small D2H read, synchronises)."""
        h = (self._g if g is None else g).geom[:32].view(torch.int32).cpu().tolist()
        return dict(num_rendered=h[0], num_tiles=h[1], overflow=h[2], max_tile_count=h[3], num_visible=h[4], num_candidates=h[5])

    def graph_overflowed(self, g=None):
        """True if the last replayed iteration (of graph g: a frame of the window; default the current one) produced more instances than
        the captured capacity (one small D2H read)."""
        return bool((self._g if g is None else g).geom[:12].view(torch.int32)[2].item())

    def _params(self):
        return dict(xyz=self.xyz, shs=self.shs, opacity=self.opacity_raw, scaling=self.scaling_raw, rotation=self.rotation_raw)

    @torch.no_grad()
    def step(self, gt_color, gt_depth, render_mask, tile_mask=None):
        """One mapping iteration; returns the op's 9-tuple (views of this iteration's outputs) — losses are in self.loss."""
        lib = N.lib()
        P, M = self.P, self.M
        self._settle_replays()  # (replays of invalid frames since the last check do not count: this step's bias corrections depend on it)
        with torch.cuda.device(self.device):
            stream = N.current_stream()
            if not self._act_valid:  # later iterations get the activations from the previous Adam step
                N.check(lib.dqo_map_activate(P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                             N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), stream))
            ctx = _Ctx()
            out = dgr._RasterizeGaussians.forward(ctx, self.xyz, self.shs, self._empty, self.opacity, self.scales, self.rotations,
                                                  self._empty, self.tile_mask if tile_mask is None else tile_mask, self.settings,
                                                  self.gaussian_object, self.pixel_object)
            # NOTE: an eager step on an overflowed frame (lazy mode) is a no-op for the optimiser (DqoAdamStep.frame_header), but the
            # host-side step_count below still advances; the operator raises at its next synchronisation point in that case.
            color, depth, hit_depth = out[0], out[1], out[3]
            H, W = color.shape[1], color.shape[2]
            mask_u8 = None if render_mask is None else render_mask
            if mask_u8 is not None and mask_u8.dtype != torch.uint8:
                mask_u8 = mask_u8.to(torch.uint8)
            if self.per_object_loss and self.gaussian_object is not None:
                # the per-object loss in eager torch (the captured path computes it in the blend kernels: DqoLossTap.per_object)
                with torch.enable_grad():
                    c_, d_ = color.detach().requires_grad_(True), depth.detach().requires_grad_(True)
                    tot, parts = mapping.per_object_loss(dict(render=c_, depth=d_, depth_index_map=hit_depth), gt_color, gt_depth,
                                                         self.pixel_object, render_mask=mask_u8, add_depth_thres=self.add_depth_thres)
                    gc_, gd_ = torch.autograd.grad(tot, [c_, d_])
                self.dL_dcolor.copy_(gc_), self.dL_ddepth.copy_(gd_)
                self.loss[:3] = torch.stack([parts["total_loss"], parts["color_loss"], parts["depth_loss"]])
            else:
                N.check(lib.dqo_map_loss_fwd_bwd(W, H, N.ptr(color), N.ptr(depth), N.ptr(hit_depth), N.ptr(gt_color), N.ptr(gt_depth),
                                                 N.ptr(mask_u8), self.color_weight, self.depth_weight, self.add_depth_thres,
                                                 N.ptr(self.loss), N.ptr(self.dL_dcolor), N.ptr(self.dL_ddepth), N.ptr(self.loss_ws),
                                                 self.loss_ws.numel(), stream))
                if mask_u8 is None and self.ssim_weight != 0:
                    self._ssim_term(N.ptr(color), gt_color, stream)
            ctx.sparse_grad_rows = True  # gradient rows of culled Gaussians stay unwritten; the Adam kernel gets radii instead
            grads = dgr._RasterizeGaussians.backward(ctx, self.dL_dcolor, self.dL_ddepth, None, None, None, None, None, None, None)
            g_means3D, g_sh, _, g_opacity, g_scales, g_rot = grads[0], grads[1], grads[2], grads[3], grads[4], grads[5]
            self.step_count += 1
            st = N.DqoAdamStep(P=P, M=M, step=self.step_count, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                               lr_xyz=self.lrs["xyz"], lr_f_dc=self.lrs["f_dc"], lr_f_rest=self.lrs["f_rest"],
                               lr_opacity=self.lrs["opacity"], lr_scaling=self.lrs["scaling"], lr_rotation=self.lrs["rotation"],
                               xyz=N.ptr(self.xyz), shs=N.ptr(self.shs), opacity_raw=N.ptr(self.opacity_raw),
                               scaling_raw=N.ptr(self.scaling_raw), rotation_raw=N.ptr(self.rotation_raw), g_means3D=N.ptr(g_means3D),
                               g_sh=N.ptr(g_sh), g_opacity=N.ptr(g_opacity), g_scales=N.ptr(g_scales), g_rotations=N.ptr(g_rot),
                               m_xyz=N.ptr(self.state["xyz"][0]), m_shs=N.ptr(self.state["shs"][0]),
                               m_opacity=N.ptr(self.state["opacity"][0]), m_scaling=N.ptr(self.state["scaling"][0]),
                               m_rotation=N.ptr(self.state["rotation"][0]), v_xyz=N.ptr(self.state["xyz"][1]),
                               v_shs=N.ptr(self.state["shs"][1]), v_opacity=N.ptr(self.state["opacity"][1]),
                               v_scaling=N.ptr(self.state["scaling"][1]), v_rotation=N.ptr(self.state["rotation"][1]),
                               act_opacity=N.ptr(self.opacity), act_scales=N.ptr(self.scales), act_rotations=N.ptr(self.rotations),
                               radii=N.ptr(out[8]), moment_live=N.ptr(self.moment_live), frame_header=N.ptr(ctx.saved_tensors[8]),
                               row_flags=N.ptr(self.row_flags), confidence=N.ptr(self.confidence) if self.count_confidence else None,
                               **self._attach_fields())
            N.check(lib.dqo_map_adam_step(ctypes.byref(st), stream))
            self._attach_n = (P + 255) // 256
            self._act_valid = True
        return out
