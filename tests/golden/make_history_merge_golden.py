"""Generates tests/golden/history_merge_golden.npz in the authoring container (CPU): the REFERENCE's own `slerp`
(/root/reference/SLAM/utils.py:650-709, imported exactly as tests/golden/make_tilemask_golden.py imports that module) inside the
statements of Mapping.history_merge (/root/reference/SLAM/multiprocess/mapper.py:607-650), which is a method of a class whose module
needs CUDA, open3d and the CUDA extensions at import time — its ten lerp statements are spelled out here with torch on the CPU (the
reference's own tensor library), including its `history_weight[0]` indexing.  Inputs and expected outputs = data; only the .npz is
committed."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_tilemask_golden import import_reference_utils  # noqa: E402


def main():
    u = import_reference_utils()
    rng = np.random.default_rng(20251005)
    out = {}
    for ci, (P, M) in enumerate([(257, 4), (64, 16), (1, 16), (300, 2)]):
        f = np.float32
        conf0 = rng.integers(0, 60, (P, 1)).astype(f)
        conf = conf0 + rng.integers(0, 40, (P, 1)).astype(f)
        if P > 3:
            conf[1], conf0[1] = 0.0, 0.0       # never counted: weight 0 / 1e-6 = 0
            conf[2], conf0[2] = 7.0, 7.0       # nothing gained in the call: weight max_weight (x 7 / (7 + 1e-6))
        hist = dict(xyz=rng.normal(size=(P, 3)).astype(f), dc=rng.normal(size=(P, 1, 3)).astype(f), rest=rng.normal(size=(P, M - 1, 3)).astype(f),
                    scaling=rng.normal(-4, 0.5, (P, 3)).astype(f))
        q0 = rng.normal(size=(P, 4)).astype(f)
        q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
        cur = dict(xyz=hist["xyz"] + rng.normal(0, 0.01, (P, 3)).astype(f), dc=hist["dc"] + rng.normal(0, 0.05, (P, 1, 3)).astype(f),
                   rest=hist["rest"] + rng.normal(0, 0.01, (P, M - 1, 3)).astype(f), scaling=hist["scaling"] + rng.normal(0, 0.05, (P, 3)).astype(f))
        # raw rotations now: mostly close to the history (the lerp branch: |dot| > 0.9995), some far (the slerp branch), one opposite
        q = q0 * rng.uniform(0.7, 1.3, (P, 1)).astype(f) + rng.normal(0, 0.004, (P, 4)).astype(f)
        far = rng.uniform(size=P) < 0.3
        q[far] += rng.normal(0, 0.4, (int(far.sum()), 4)).astype(f)
        if P > 5:
            q[4] = -q0[4]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        max_weight = 0.5
        # ---- mapper.py:610-644, on CPU tensors ----
        history_weight = max_weight * t(conf0) / (t(conf) + 1e-6)
        get_rotation = torch.nn.functional.normalize(t(q))  # GaussianPointCloud.get_rotation, gaussian_pointcloud.py:746-747
        xyz_merge = t(hist["xyz"]) * history_weight + (1 - history_weight) * t(cur["xyz"])
        dc_merge = t(hist["dc"]) * history_weight[0] + (1 - history_weight[0]) * t(cur["dc"])
        rest_merge = t(hist["rest"]) * history_weight[0] + (1 - history_weight[0]) * t(cur["rest"])
        scaling_merge = t(hist["scaling"]) * history_weight[0] + (1 - history_weight[0]) * t(cur["scaling"])
        rotation_merge = u.slerp(t(q0), get_rotation, 1 - history_weight)
        pre = f"c{ci}_"
        out.update({pre + "conf0": conf0, pre + "conf": conf, pre + "rot0": q0, pre + "rot_raw": q, pre + "max_weight": np.float32(max_weight)})
        for k in ("xyz", "dc", "rest", "scaling"):
            out[pre + "hist_" + k], out[pre + "cur_" + k] = hist[k], cur[k]
        out.update({pre + "out_xyz": xyz_merge.numpy(), pre + "out_dc": dc_merge.numpy(), pre + "out_rest": rest_merge.numpy(),
                    pre + "out_scaling": scaling_merge.numpy(), pre + "out_rotation": rotation_merge.numpy()})
    out["n_cases"] = np.int32(4)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "history_merge_golden.npz"), **out)
    print("wrote history_merge_golden.npz:", {k: v.shape for k, v in out.items() if k.startswith("c0_")})


if __name__ == "__main__":
    main()
