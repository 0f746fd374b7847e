"""Debug helper: where do sparse-moment and dense Adam differ (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "dqo-map_amd"))
import numpy as np, torch
import test_gpu_fused_mapping as T
from dqo_harness.fused_mapping import FusedMapper
cam, scene, settings, gt_color, gt_depth, mask, dev = T._problem(torch, P=20000, cfg=3)
a = FusedMapper(scene, settings, dev, sparse_moments=True)
b = FusedMapper(scene, settings, dev, sparse_moments=False)
for it in range(3):
    oa = a.step(gt_color, gt_depth, mask)
    ob = b.step(gt_color, gt_depth, mask)
    torch.cuda.synchronize()
    radii = oa[8].cpu().numpy()
    live = a.moment_live.cpu().numpy()
    for k, pa in a._params().items():
        pb = b._params()[k]
        d = (pa != pb)
        if d.any():
            rows = d.reshape(d.shape[0], -1).any(1).cpu().numpy()
            cols = d.reshape(d.shape[0], -1).any(0).cpu().numpy()
            idx = np.nonzero(rows)[0]
            print("it", it, k, "rows differ:", rows.sum(), "first", idx[:10], "radii", radii[idx[:10]], "live", live[idx[:10]], "cols", np.nonzero(cols)[0][:20],
                  "maxdiff", (pa - pb).abs().max().item())
        for s in (0, 1):
            d = a.state[k][s] != b.state[k][s]
            if d.any():
                rows = d.reshape(d.shape[0], -1).any(1).cpu().numpy()
                print("it", it, k, "moment", s, "rows differ:", rows.sum(), np.nonzero(rows)[0][:10])
print("done")
