"""Soak / determinism check of the captured mapping iteration: the same map optimised twice for N iterations (with the automatic
re-capture of FusedMapper.run) must end in bit-identical parameters, moments and losses, all finite — the per-object job (object gate,
per-object loss tap, fused per-Gaussian tail), optionally with the long lists shared between eight waves (whose blocks draw their work
by ticket: which block blends which list changes from run to run, the results must not).   python tools/soak.py [cfg] [N] [list_split]"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
split = int(sys.argv[3]) if len(sys.argv) > 3 else 0
args = argparse.Namespace(cfg=cfg, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
from dqo_harness.fused_mapping import FusedMapper
mask = prob["render_mask"].to(torch.uint8).contiguous()
ends = []
for rep in range(2):
    fm = FusedMapper(prob["scene"], prob["settings"], dev)
    if prob.get("gate") is not None:
        fm.set_object_gate(prob["gate"][0], prob["gate"][1])
    fm.capture(prob["gt_color"], prob["gt_depth"], mask, tile_mask=prob["tile_mask"], list_split=split)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    recaps = fm.run(n, check_every=256)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = {k: v.clone() for k, v in fm._params().items()}
    st.update({k + "_m": m.clone() for k, (m, v) in fm.state.items()})
    st["loss"] = fm.loss.clone()
    ends.append(st)
    print(f"cfg {cfg} list_split {int(fm._g.ls_fwd)} run {rep}: {n} iterations in {dt:.2f} s ({n / dt:.0f} iter/s incl. {recaps} re-captures), loss {fm.loss[:3].tolist()}, "
          f"header {fm.header()}")
    assert all(torch.isfinite(v).all() for v in st.values()), "non-finite state"
same = all(torch.equal(ends[0][k], ends[1][k]) for k in ends[0])
print("bit-identical end states:", same)
sys.exit(0 if same else 1)
