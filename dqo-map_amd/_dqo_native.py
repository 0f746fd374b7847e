"""ctypes binding of libdqoraster.so (include/dqo_raster.h).  Shared by the drop-in packages in this directory.

There is NO CPU fallback: if the HIP library is missing or a tensor is not on the GPU the call raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdqoraster.so")

c_f = ctypes.c_float
c_i32 = ctypes.c_int32
c_vp = ctypes.c_void_p


class DqoRastParams(ctypes.Structure):
    _fields_ = [("P", c_i32), ("D", c_i32), ("M", c_i32), ("W", c_i32), ("H", c_i32), ("prefiltered", c_i32), ("debug", c_i32),
                ("tanfovx", c_f), ("tanfovy", c_f), ("cx", c_f), ("cy", c_f), ("scale_modifier", c_f), ("color_sigma", c_f),
                ("opaque_threshold", c_f), ("depth_threshold", c_f), ("normal_threshold", c_f), ("T_threshold", c_f)]


class DqoRastInputs(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("bg", "means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp",
                                    "viewmatrix", "projmatrix", "campos", "tile_mask", "row_flags")]


class DqoRastOutputs(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("out_color", "out_depth", "out_hit_color", "out_hit_depth", "out_hit_color_weight",
                                    "out_hit_depth_weight", "out_T", "n_touched", "radii")]


class DqoLossTap(ctypes.Structure):
    _fields_ = [("gt_color", c_vp), ("gt_depth", c_vp), ("render_mask", c_vp), ("out_color", c_vp), ("out_depth", c_vp),
                ("color_weight", c_f), ("depth_weight", c_f), ("add_depth_thres", c_f), ("loss_out", c_vp), ("grad_scale", c_vp),
                ("per_object", c_i32)]


class DqoObjectGate(ctypes.Structure):
    _fields_ = [("gaussian_object", c_vp), ("pixel_object", c_vp), ("tile_objects", c_vp)]


class DqoRastCtx(ctypes.Structure):
    _fields_ = [("geom", c_vp), ("geom_bytes", ctypes.c_size_t), ("binning", c_vp), ("binning_bytes", ctypes.c_size_t),
                ("image", c_vp), ("image_bytes", ctypes.c_size_t), ("inst_capacity", ctypes.c_int64), ("tile_bucket_capacity", c_i32), ("keep_tile_order", c_i32), ("loss_tap", c_vp),
                ("object_gate", c_vp), ("list_split", c_i32), ("frame_prezeroed", c_i32)]


class DqoRastGrads(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("dL_dmeans3D", "dL_dsh", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcov3D",
                                    "dL_dmeans2D")] + [("skip_culled_rows", c_i32)]


class DqoRastHeader(ctypes.Structure):
    _fields_ = [("num_rendered", ctypes.c_uint32), ("num_tiles", ctypes.c_uint32), ("overflow", ctypes.c_uint32),
                ("max_tile_count", ctypes.c_uint32), ("num_visible", ctypes.c_uint32), ("num_candidates", ctypes.c_uint32),
                ("stage", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class DqoProfileEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("total_ms", ctypes.c_double), ("calls", ctypes.c_uint32)]


class DqoAdamStep(ctypes.Structure):
    _fields_ = ([("P", c_i32), ("M", c_i32), ("step", c_i32), ("beta1", c_f), ("beta2", c_f), ("eps", c_f), ("lr_xyz", c_f),
                 ("lr_f_dc", c_f), ("lr_f_rest", c_f), ("lr_opacity", c_f), ("lr_scaling", c_f), ("lr_rotation", c_f)] +
                [(n, c_vp) for n in ("xyz", "shs", "opacity_raw", "scaling_raw", "rotation_raw", "g_means3D", "g_sh", "g_opacity",
                                     "g_scales", "g_rotations", "m_xyz", "m_shs", "m_opacity", "m_scaling", "m_rotation", "v_xyz",
                                     "v_shs", "v_opacity", "v_scaling", "v_rotation", "act_opacity", "act_scales", "act_rotations", "radii", "step_dev", "moment_live",
                                     "attach_mask", "init_xyz", "init_scaling_raw", "init_rotation_raw")] +
                [("attach_count", c_i32), ("attach_partial", c_vp), ("frame_header", c_vp), ("block_ticket", c_vp), ("bias_table", c_vp), ("attach_gains", c_vp),
                 ("row_flags", c_vp), ("confidence", c_vp), ("lr_table", c_vp)])

ROW_FROZEN, ROW_HIDDEN = 1, 2  # DQO_ROW_FROZEN / DQO_ROW_HIDDEN (include/dqo_raster.h)


class DqoAdamTensor(ctypes.Structure):
    _fields_ = [("p", c_vp), ("g", c_vp), ("m", c_vp), ("v", c_vp), ("n", ctypes.c_int64), ("lr", ctypes.c_double)]


EXPORTS = ("dqo_abi_version", "dqo_abi_sizeof", "dqo_last_error", "dqo_profile_enable", "dqo_profile_collect", "dqo_map_activate",
           "dqo_map_loss_workspace_bytes", "dqo_map_loss_fwd_bwd", "dqo_map_ssim_workspace_bytes", "dqo_map_ssim_fwd_bwd", "dqo_map_adam_step", "dqo_adam_multi_dev", "dqo_map_attach_workspace_bytes",
           "dqo_map_attach_loss_fwd_bwd", "dqo_adam_multi", "dqo_accumulate_gaussian_error", "dqo_accumulate_gaussian_confidence", "dqo_rast_geom_bytes", "dqo_rast_image_bytes",
           "dqo_rast_binning_bytes", "dqo_rast_binning_bytes_bucketed",
           "dqo_rast_backward_workspace_bytes", "dqo_rast_forward_prepare", "dqo_rast_read_header", "dqo_rast_forward_render",
           "dqo_rast_forward", "dqo_rast_forward_async", "dqo_rast_backward", "dqo_rast_backward_adam", "dqo_mark_visible", "dqo_knn3_workspace_bytes", "dqo_knn3",
           "dqo_quadric_iou_fwd_bwd", "dqo_quadric_adam", "dqo_tile_count_mask", "dqo_transmission_mask", "dqo_tile_color_error", "dqo_knn3_query_workspace_bytes",
           "dqo_knn3_query", "dqo_knn3_query_within", "dqo_knn3_query_grouped", "dqo_icp_workspace_bytes", "dqo_icp_normal_equations",
           "dqo_attach_pixels", "dqo_attach_decide", "dqo_growth_scales", "dqo_growth_inside", "dqo_error_maps", "dqo_map_history_merge")

_lib = None


def lib():
    """Load libdqoraster.so (built by `make -C dqo-map_amd/csrc` / __graft_entry__.build()).  Fails loudly when absent."""
    global _lib
    if _lib is None:
        # torch first: it loads its bundled HIP runtime; libdqoraster.so then binds to that same copy (same SONAME).  The
        # other order puts two ROCr instances in one process and the second one sees "no ROCm-capable device".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()). "
                               "There is no CPU fallback for this operator.")
        L = ctypes.CDLL(LIB_PATH)
        L.dqo_last_error.restype = ctypes.c_char_p
        for n in ("dqo_rast_geom_bytes", "dqo_rast_image_bytes", "dqo_rast_binning_bytes", "dqo_rast_backward_workspace_bytes",
                  "dqo_knn3_workspace_bytes"):
            getattr(L, n).restype = ctypes.c_size_t
        L.dqo_rast_geom_bytes.argtypes = [c_i32, c_i32, c_i32]
        L.dqo_rast_image_bytes.argtypes = [c_i32, c_i32]
        L.dqo_rast_binning_bytes.argtypes = [ctypes.c_int64]
        L.dqo_rast_binning_bytes_bucketed.restype = ctypes.c_size_t
        L.dqo_rast_binning_bytes_bucketed.argtypes = [ctypes.c_int64, c_i32, c_i32, c_i32]
        L.dqo_rast_backward_workspace_bytes.argtypes = [ctypes.c_int64]
        L.dqo_knn3_workspace_bytes.argtypes = [c_i32]
        P = ctypes.POINTER
        L.dqo_rast_forward_prepare.argtypes = [P(DqoRastParams), P(DqoRastInputs), P(DqoRastOutputs), P(DqoRastCtx), c_vp]
        L.dqo_rast_forward_render.argtypes = L.dqo_rast_forward_prepare.argtypes
        L.dqo_rast_forward.argtypes = L.dqo_rast_forward_prepare.argtypes
        L.dqo_rast_forward_async.argtypes = L.dqo_rast_forward_prepare.argtypes[:4] + [c_vp, c_vp, c_vp]
        L.dqo_rast_read_header.argtypes = [P(DqoRastCtx), P(DqoRastHeader), c_vp]
        L.dqo_rast_backward.argtypes = [P(DqoRastParams), P(DqoRastInputs), P(DqoRastCtx), c_vp, c_vp, c_vp, P(DqoRastGrads), c_vp,
                                        ctypes.c_size_t, c_vp]
        L.dqo_rast_backward_adam.argtypes = [P(DqoRastParams), P(DqoRastInputs), P(DqoRastCtx), c_vp, c_vp, P(DqoAdamStep), c_vp,
                                             ctypes.c_size_t, c_vp]
        L.dqo_mark_visible.argtypes = [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.dqo_knn3.argtypes = [c_i32, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]
        L.dqo_quadric_iou_fwd_bwd.argtypes = [c_i32] + [c_vp] * 12
        L.dqo_quadric_adam.argtypes = [c_i32, c_i32] + [c_vp] * 9
        L.dqo_map_activate.argtypes = [c_i32] + [c_vp] * 7
        L.dqo_map_loss_workspace_bytes.restype = ctypes.c_size_t
        L.dqo_map_loss_workspace_bytes.argtypes = []
        L.dqo_map_ssim_workspace_bytes.restype = ctypes.c_size_t
        L.dqo_map_ssim_workspace_bytes.argtypes = [c_i32, c_i32]
        L.dqo_map_ssim_fwd_bwd.argtypes = [c_i32, c_i32, c_vp, c_vp, c_f, c_vp, c_vp, c_i32, c_vp, c_vp, ctypes.c_size_t, c_vp]
        L.dqo_map_loss_fwd_bwd.argtypes = [c_i32, c_i32] + [c_vp] * 6 + [c_f, c_f, c_f] + [c_vp] * 4 + [ctypes.c_size_t, c_vp]
        L.dqo_map_adam_step.argtypes = [P(DqoAdamStep), c_vp]
        L.dqo_map_attach_workspace_bytes.restype = ctypes.c_size_t
        L.dqo_map_attach_workspace_bytes.argtypes = [c_i32]
        L.dqo_adam_multi.argtypes = [c_vp, c_i32, c_i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_vp]
        L.dqo_adam_multi_dev.argtypes = [c_vp, c_i32, c_vp, c_i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_vp]
        L.dqo_map_attach_loss_fwd_bwd.argtypes = [c_i32] + [c_vp] * 7 + [c_i32] + [c_vp] * 5 + [ctypes.c_size_t, c_vp]
        L.dqo_accumulate_gaussian_error.argtypes = [c_i32] * 3 + [c_vp] * 5 + [c_f] * 3 + [c_i32] + [c_vp] * 6
        L.dqo_accumulate_gaussian_confidence.argtypes = [c_i32] * 3 + [c_vp] * 7
        L.dqo_knn3_query_workspace_bytes.restype = ctypes.c_size_t
        L.dqo_knn3_query_workspace_bytes.argtypes = [c_i32, c_i32]
        L.dqo_knn3_query.argtypes = [c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]
        L.dqo_knn3_query_within.argtypes = [c_i32, c_vp, c_i32, c_vp, c_f, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]
        L.dqo_knn3_query_grouped.argtypes = [c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_f, c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp]
        L.dqo_attach_pixels.argtypes = [c_i32, c_vp, c_vp, c_f, c_f, c_f, c_f, c_i32, c_i32] + [c_vp] * 5
        L.dqo_attach_decide.argtypes = [c_i32] + [c_vp] * 10 + [c_f, c_f, c_vp, c_vp]
        L.dqo_growth_scales.argtypes = [c_i32] + [c_vp] * 7 + [c_f, c_f, c_f, c_vp, c_vp, c_vp]
        L.dqo_growth_inside.argtypes = [c_i32] + [c_vp] * 5
        L.dqo_error_maps.argtypes = [c_i32, c_i32] + [c_vp] * 9
        L.dqo_icp_workspace_bytes.restype = ctypes.c_size_t
        L.dqo_icp_workspace_bytes.argtypes = []
        L.dqo_icp_normal_equations.argtypes = [c_i32, c_i32] + [c_vp] * 5 + [c_f] * 6 + [c_vp] * 4 + [ctypes.c_size_t, c_vp]
        L.dqo_tile_count_mask.argtypes = [c_i32, c_i32, c_vp, c_vp, c_vp]
        L.dqo_transmission_mask.argtypes = [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.dqo_tile_color_error.argtypes = [c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]
        L.dqo_map_history_merge.argtypes = [c_i32, c_i32, c_f, c_i32] + [c_vp] * 12
        L.dqo_profile_enable.argtypes = [ctypes.c_int]
        L.dqo_profile_collect.argtypes = [P(DqoProfileEntry), ctypes.c_int, ctypes.c_int]
        if L.dqo_abi_version() != 5:
            raise RuntimeError("libdqoraster.so ABI version mismatch")
        L.dqo_abi_sizeof.restype = ctypes.c_size_t
        L.dqo_abi_sizeof.argtypes = [c_i32]
        for k, st in enumerate((DqoRastParams, DqoRastInputs, DqoRastOutputs, DqoRastCtx, DqoRastGrads, DqoRastHeader, DqoProfileEntry,
                                DqoAdamStep, DqoLossTap, DqoObjectGate, DqoAdamTensor)):
            if L.dqo_abi_sizeof(k) != ctypes.sizeof(st):
                raise RuntimeError(f"libdqoraster.so: struct {st.__name__} is {L.dqo_abi_sizeof(k)} bytes in the library, "
                                   f"{ctypes.sizeof(st)} in the binding")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(lib().dqo_last_error().decode() or f"libdqoraster error {rc}")


def ptr(t):
    """Raw device pointer of a torch tensor (None / empty tensor -> NULL, like an empty tensor's data_ptr in the reference)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and t.numel() > 0 and not t.is_cuda:
            raise RuntimeError("libdqoraster operators need GPU (ROCm) tensors; there is no CPU path. Got a tensor on "
                               f"{t.device}.")


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def profile_enable(on):
    lib().dqo_profile_enable(1 if on else 0)


def profile_collect(reset=True):
    """{kernel name: (total_ms, calls)} of every launch bracketed since the last reset (synchronises)."""
    buf = (DqoProfileEntry * 64)()
    n = lib().dqo_profile_collect(buf, 64, 1 if reset else 0)
    return {buf[i].name.decode(): (buf[i].total_ms, buf[i].calls) for i in range(n)}
