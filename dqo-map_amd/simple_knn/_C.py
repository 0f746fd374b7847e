"""`simple_knn._C` — same surface as the reference's pybind module (/root/reference/submodules/simple-knn/ext.cpp:15-17).

distCUDA2(points[P,3] float32, GPU) -> (meanDist2 float32[P], knn_idx int32[P,3])     (spatial.cu:15-28; this fork
returns the neighbour indices as well).  Backed by dqo_knn3 in libdqoraster.so; no CPU path.
"""
import torch

import _dqo_native as N


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("distCUDA2 needs a GPU (ROCm) tensor; there is no CPU path.")
    if points.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {points.dtype}")
    points = points.contiguous()
    P = points.size(0)
    dev = points.device
    means = torch.zeros((P,), dtype=torch.float32, device=dev)
    indices = torch.zeros((P, 3), dtype=torch.int32, device=dev)
    if P == 0:
        return means, indices
    lib = N.lib()
    with torch.cuda.device(dev):
        ws = torch.empty((lib.dqo_knn3_workspace_bytes(P),), dtype=torch.uint8, device=dev)
        N.check(lib.dqo_knn3(P, points.data_ptr(), means.data_ptr(), indices.data_ptr(), ws.data_ptr(), ws.numel(),
                             N.current_stream()))
    return means, indices
