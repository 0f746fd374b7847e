import os, sys
R = "/root/repo" if os.path.isdir("/root/repo/dqo-map_amd") else os.getcwd()
sys.path[:0] = [R, R + "/dqo-map_amd"]
import numpy as np, torch
import _dqo_native as N
import dqo_mapgrowth as mg
from simple_knn._C import distCUDA2
rng = np.random.default_rng(0)
for Rn, Q in ((2_000_000, 40_800), (540_000, 0)):
    r = torch.tensor(rng.uniform(-3, 3, (Rn, 3)).astype(np.float32), device="cuda")
    q = torch.tensor(rng.uniform(-3, 3, (max(Q, 1), 3)).astype(np.float32), device="cuda")
    f = (lambda: mg.knn_points_k3(q, r)) if Q else (lambda: distCUDA2(r))
    for _ in range(2): f()
    torch.cuda.synchronize()
    N.profile_enable(True); N.profile_collect(reset=True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    prof = N.profile_collect(reset=True); N.profile_enable(False)
    tot = sum(v[0] for v in prof.values()) / 5
    print(("query %d vs %d" % (Q, Rn)) if Q else ("distCUDA2 on %d" % Rn), "total %.2f ms" % tot)
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print("   %-28s calls/run %5.1f  ms/run %7.3f" % (k, v[1] / 5, v[0] / 5))
