"""Generates tests/golden/tilemask_golden.npz by importing the REFERENCE module /root/reference/SLAM/utils.py in the
authoring container (CPU, no GPU) and calling its own pixelmask2tilemask / transmission2tilemask / meanpool /
colorerror2tilemask (SLAM/utils.py:720-799) on seeded inputs.

Only runs where /root/reference exists; the produced .npz (inputs + expected outputs = data) is committed, the reference
source never is.  The module imports third-party packages that are absent here and that the four functions do not use
(cv2, open3d, plyfile, pytorch3d, skimage): empty stub modules stand in for them.  `utils.general_utils` (the reference's
own helper module) creates CUDA tensors at import time; a stand-in provides its three dtype casts devB / devF / devI
(general_utils.py:27-36: `tensor.type_as(<bool|float32|int32 dummy on cuda>)`) as the same casts on the CPU.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def import_reference_utils():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("cv2", COLORMAP_JET=2)  # only a default argument value of a plotting helper
    stub("open3d")
    stub("plyfile", PlyData=object, PlyElement=object)
    stub("pytorch3d")
    stub("pytorch3d.loss", chamfer_distance=None)
    stub("pytorch3d.ops", knn_points=None)
    stub("skimage", filters=None)
    stub("skimage.filters")
    stub("skimage.color", rgb2gray=None)
    stub("utils")
    stub("utils.general_utils", devB=lambda t: t.to(torch.bool), devF=lambda t: t.to(torch.float32), devI=lambda t: t.to(torch.int32),
         quaternion_from_axis_angle=None)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_slam_utils", os.path.join(REF, "SLAM", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    u = import_reference_utils()
    rng = np.random.default_rng(20250117)
    out = {}
    for ci, (h, w) in enumerate([(136, 240), (37, 50), (16, 16), (96, 128), (5, 100)]):
        # transmission map: 1 where nothing was rendered, < 1 elsewhere, in blobs so that tiles are full / partial / empty
        T = np.ones((h, w), np.float32)
        for _ in range(12):
            y0, x0 = rng.integers(0, h), rng.integers(0, w)
            hh, ww = rng.integers(1, max(2, h // 2)), rng.integers(1, max(2, w // 2))
            T[y0:y0 + hh, x0:x0 + ww] = rng.uniform(0, 0.999, (min(hh, h - y0), min(ww, w - x0))).astype(np.float32)
        mask = torch.from_numpy(T != 1)
        render = (rng.integers(0, 256, (3, h, w)).astype(np.float32) / 256.0).astype(np.float32)  # stored as uint8 (value * 256)
        render[:, T == 1] = 0  # unrendered pixels are black: exercises the filter of mapper.py:955-956
        gt = (rng.integers(0, 256, (3, h, w)).astype(np.float32) / 256.0).astype(np.float32)
        # color error exactly as evaluate_render_range builds it (mapper.py:949-956), with torch on the CPU
        ri, gi = torch.from_numpy(render).permute(1, 2, 0), torch.from_numpy(gt).permute(1, 2, 0)
        ce = torch.sum((ri - gi).abs(), dim=-1, keepdim=False)
        ce[ri.sum(dim=-1) == 0] = 0
        out[f"c{ci}_T"] = T
        out[f"c{ci}_render_u8"] = np.round(render * 256).astype(np.uint8)  # render = u8 / 256 exactly
        out[f"c{ci}_gt_u8"] = np.round(gt * 256).astype(np.uint8)
        out[f"c{ci}_color_error"] = ce.numpy()
        out[f"c{ci}_pixelmask2tilemask"] = u.pixelmask2tilemask(mask, 16).numpy()
        for r in (0.5, 0.25, 0.9):
            out[f"c{ci}_transmission2tilemask_{r}"] = u.transmission2tilemask(mask, 16, r).numpy()
        out[f"c{ci}_meanpool"] = u.meanpool(ce, 16).numpy()
        for r in (0.4, 0.1):
            out[f"c{ci}_colorerror2tilemask_{r}"] = u.colorerror2tilemask(ce, 16, r).numpy()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tilemask_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items() if k.startswith("c1_")})


if __name__ == "__main__":
    main()
