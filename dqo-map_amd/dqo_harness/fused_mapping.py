"""Fused mapping iteration: the same arithmetic as

    out = render(settings, params.activated()); loss = mapping_loss(out, gt, mask); loss.backward(); adam.step()

(dqo_harness/mapping.py, i.e. SLAM/multiprocess/mapper.py:531-605 + 836-875 + gaussian_pointcloud.py:331-378), but without
the autograd graph and the ~110 eager launches per iteration: activation kernel -> rasteriser forward -> fused masked-loss
forward+backward -> rasteriser backward -> fused (activation-Jacobian + Adam) kernel.  The rasteriser is called through the
same operator code (`_RasterizeGaussians.forward / .backward`), so it is the same C-ABI path the drop-in op uses.

`capture()` records one whole iteration (13 kernel launches + one 16 KB memset, no host synchronisation, the Adam step count
kept on the device) into a hipGraph over persistent buffers; `replay()` re-issues it with a single launch call — at ~0.78 ms
of GPU work per iteration on config 3 the per-launch host work of the eager path is otherwise as long as the GPU work.

Only the masked-loss case is fused (SSIM needs an 11x11 convolution and is skipped by the reference when a render mask is
given, B14); without a mask use the autograd path of dqo_harness/mapping.py.  GPU only.
"""
import ctypes

import numpy as np
import torch

import _dqo_native as N
import diff_gaussian_rasterization_depth as dgr
from dqo_harness import mapping


class _Ctx:
    """Stand-in for the autograd ctx when the op's static forward/backward are driven directly."""

    def save_for_backward(self, *a):
        self.saved_tensors = a

    def mark_non_differentiable(self, *a):
        pass


class FusedMapper:
    def __init__(self, scene, settings, device, lrs=None, betas=(0.9, 0.999), eps=1e-15, color_weight=mapping.COLOR_WEIGHT,
                 depth_weight=mapping.DEPTH_WEIGHT, add_depth_thres=0.1, sparse_moments=True):
        t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=device)
        self.device = device
        self.settings = settings
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
        self._act_valid = False  # opacity / scales / rotations hold the activations of the current raw parameters
        P = self.P
        f = dict(dtype=torch.float32, device=device)
        self.opacity = torch.empty((P, 1), **f)
        self.scales = torch.empty((P, 3), **f)
        self.rotations = torch.empty((P, 4), **f)
        H, W = settings.image_height, settings.image_width
        self.dL_dcolor = torch.empty((3, H, W), **f)
        self.dL_ddepth = torch.empty((1, H, W), **f)
        self.loss = torch.zeros(4, **f)
        lib = N.lib()
        self.loss_ws = torch.empty((lib.dqo_map_loss_workspace_bytes(),), dtype=torch.uint8, device=device)
        self._empty = torch.Tensor([])
        self.tile_mask = torch.ones(((H + 15) // 16, (W + 15) // 16), dtype=torch.int32, device=device)

    # ------------------------------------------------------------------ hipGraph path ------------------------------------
    def capture(self, gt_color, gt_depth, render_mask, tile_mask=None, capacity_margin=1.15, tile_buckets=True):
        """Allocate persistent buffers for every intermediate of an iteration, run it once eagerly, then capture it into a
        hipGraph.  The inputs (gt images, masks) are read from the tensors passed here at every replay().

        The capture fixes two capacities from the state it is taken on: the instance capacity (candidates x capacity_margin)
        and, with tile_buckets, the per-tile list bucket (twice the longest list, power of two).  A replay that outgrows
        either leaves invalid outputs and raises the device-side overflow flag — check graph_overflowed() (one small D2H read,
        e.g. once per batch of replays) and call capture() again when it is set; tile_buckets=False keeps the packed lists
        (any list length, one more kernel per iteration)."""
        lib = N.lib()
        dev, P, M = self.device, self.P, self.M
        st = self.settings
        H, W = int(st.image_height), int(st.image_width)
        f = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            # capacity: the reference's num_rendered of the current state (an upper bound of the instances kept) plus a margin
            # for the Gaussians that move while the graph is being replayed; the device header flags an overflow
            dgr_mode = dgr._sync_mode
            dgr.set_sync_mode("exact")
            with torch.no_grad():
                probe = dgr._RasterizeGaussians.forward(_Ctx(), self.xyz, self.shs, self._empty, *self._activated_now(), self._empty,
                                                        self.tile_mask if tile_mask is None else tile_mask, st)
            cand = dgr.last_header()["num_candidates"]
            longest = dgr.last_header()["max_tile_count"]
            dgr.set_sync_mode(dgr_mode)
            del probe
            cap = int(cand * capacity_margin) + 4096
            g = self._g = type("G", (), {})()
            g.cap = cap
            # fixed per-tile list buckets (DqoRastCtx.tile_bucket_capacity): at least twice the longest list of the current state,
            # a power of two; a tile that outgrows it raises the same overflow flag as running out of instance capacity
            g.bucket = 0
            if tile_buckets:
                g.bucket = 256
                while g.bucket < 2 * longest:
                    g.bucket *= 2
            g.gt_color, g.gt_depth = gt_color, gt_depth
            g.mask = None if render_mask is None else render_mask.to(torch.uint8).contiguous()
            g.tile_mask = self.tile_mask if tile_mask is None else tile_mask
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
            g.step_dev = torch.full((1,), self.step_count + 1, **i32)
            g.params = dgr._params(st, P, M)
            g.inputs = dgr._inputs(st, self.xyz, self.shs, self._empty, self.opacity, self.scales, self.rotations, self._empty, g.tile_mask)
            o = g.out
            g.outputs = N.DqoRastOutputs(out_color=o[0].data_ptr(), out_depth=o[1].data_ptr(), out_hit_color=o[2].data_ptr(),
                                         out_hit_depth=o[3].data_ptr(), out_hit_color_weight=o[4].data_ptr(),
                                         out_hit_depth_weight=o[5].data_ptr(), out_T=o[6].data_ptr(), n_touched=o[7].data_ptr(),
                                         radii=o[8].data_ptr())
            g.cctx = N.DqoRastCtx(geom=g.geom.data_ptr(), geom_bytes=g.geom.numel(), binning=g.binning.data_ptr(),
                                  binning_bytes=g.binning.numel(), image=g.img.data_ptr(), image_bytes=g.img.numel(), inst_capacity=cap,
                                  tile_bucket_capacity=g.bucket)
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
                                   moment_live=N.ptr(self.moment_live))
            if not self._act_valid:
                stream = N.current_stream()
                N.check(lib.dqo_map_activate(P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                             N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), stream))
                self._act_valid = True
            # one eager iteration on a side stream (warms every kernel up), then the capture
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._static_iteration()
            torch.cuda.current_stream().wait_stream(side)
            self.step_count += 1
            g.graph = torch.cuda.CUDAGraph()
            # thread_local: other threads of the process (e.g. a collective library's watchdog) may keep issuing runtime calls
            with torch.cuda.graph(g.graph, capture_error_mode="thread_local"):
                self._static_iteration()
            g.expected_step = self.step_count + 1
        return self

    def _activated_now(self):
        """(opacity, scales, rotations) of the current raw parameters, computed eagerly (capacity probe only)."""
        op = torch.sigmoid(self.opacity_raw)
        sc = torch.exp(self.scaling_raw)
        rot = torch.nn.functional.normalize(self.rotation_raw)
        return op, sc, rot

    def _static_iteration(self):
        """The five C-ABI calls of one iteration over the persistent buffers (no allocation, no host-side per-step state)."""
        lib, g = N.lib(), self._g
        st = self.settings
        H, W = int(st.image_height), int(st.image_width)
        stream = N.current_stream()
        N.check(lib.dqo_rast_forward(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.outputs), ctypes.byref(g.cctx), stream))
        o = g.out
        N.check(lib.dqo_map_loss_fwd_bwd(W, H, o[0].data_ptr(), o[1].data_ptr(), o[3].data_ptr(), N.ptr(g.gt_color), N.ptr(g.gt_depth),
                                         N.ptr(g.mask), self.color_weight, self.depth_weight, self.add_depth_thres, N.ptr(self.loss),
                                         N.ptr(self.dL_dcolor), N.ptr(self.dL_ddepth), N.ptr(self.loss_ws), self.loss_ws.numel(), stream))
        N.check(lib.dqo_rast_backward(ctypes.byref(g.params), ctypes.byref(g.inputs), ctypes.byref(g.cctx), self.dL_dcolor.data_ptr(),
                                      self.dL_ddepth.data_ptr(), o[3].data_ptr(), ctypes.byref(g.cgrads), g.ws.data_ptr(), g.ws.numel(),
                                      stream))
        N.check(lib.dqo_map_adam_step(ctypes.byref(g.adam), stream))

    def replay(self):
        """One mapping iteration by replaying the captured graph; outputs are the persistent tensors in self._g.out."""
        g = self._g
        if g.expected_step != self.step_count + 1:  # eager step() calls in between: resynchronise the device-side step count
            g.step_dev.fill_(self.step_count + 1)
        g.graph.replay()
        self.step_count += 1
        g.expected_step = self.step_count + 1
        return g.out

    def step_static(self):
        """One iteration over the persistent buffers issued eagerly — exactly the calls the captured graph holds (for per-kernel
        profiling: events cannot be recorded inside a replay)."""
        g = self._g
        if g.expected_step != self.step_count + 1:
            g.step_dev.fill_(self.step_count + 1)
        with torch.cuda.device(self.device):
            self._static_iteration()
        self.step_count += 1
        g.expected_step = self.step_count + 1
        return g.out

    def header(self):
        """Device header of the captured iteration's last forward (one small D2H read, synchronises)."""
        h = self._g.geom[:32].view(torch.int32).cpu().tolist()
        return dict(num_rendered=h[0], num_tiles=h[1], overflow=h[2], max_tile_count=h[3], num_visible=h[4], num_candidates=h[5])

    def graph_overflowed(self):
        """True if the last replayed iteration produced more instances than the captured capacity (one small D2H read)."""
        return bool(self._g.geom[:12].view(torch.int32)[2].item())

    def _params(self):
        return dict(xyz=self.xyz, shs=self.shs, opacity=self.opacity_raw, scaling=self.scaling_raw, rotation=self.rotation_raw)

    @torch.no_grad()
    def step(self, gt_color, gt_depth, render_mask, tile_mask=None):
        """One mapping iteration; returns the op's 9-tuple (views of this iteration's outputs) — losses are in self.loss."""
        lib = N.lib()
        P, M = self.P, self.M
        with torch.cuda.device(self.device):
            stream = N.current_stream()
            if not self._act_valid:  # later iterations get the activations from the previous Adam step
                N.check(lib.dqo_map_activate(P, N.ptr(self.opacity_raw), N.ptr(self.scaling_raw), N.ptr(self.rotation_raw),
                                             N.ptr(self.opacity), N.ptr(self.scales), N.ptr(self.rotations), stream))
            ctx = _Ctx()
            out = dgr._RasterizeGaussians.forward(ctx, self.xyz, self.shs, self._empty, self.opacity, self.scales, self.rotations,
                                                  self._empty, self.tile_mask if tile_mask is None else tile_mask, self.settings)
            color, depth, hit_depth = out[0], out[1], out[3]
            H, W = color.shape[1], color.shape[2]
            mask_u8 = None if render_mask is None else render_mask
            if mask_u8 is not None and mask_u8.dtype != torch.uint8:
                mask_u8 = mask_u8.to(torch.uint8)
            N.check(lib.dqo_map_loss_fwd_bwd(W, H, N.ptr(color), N.ptr(depth), N.ptr(hit_depth), N.ptr(gt_color), N.ptr(gt_depth),
                                             N.ptr(mask_u8), self.color_weight, self.depth_weight, self.add_depth_thres,
                                             N.ptr(self.loss), N.ptr(self.dL_dcolor), N.ptr(self.dL_ddepth), N.ptr(self.loss_ws),
                                             self.loss_ws.numel(), stream))
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
                               radii=N.ptr(out[8]), moment_live=N.ptr(self.moment_live))
            N.check(lib.dqo_map_adam_step(ctypes.byref(st), stream))
            self._act_valid = True
        return out
