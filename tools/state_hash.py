"""Fingerprint of what the library computes, for bit-for-bit A/B of two builds (tools/ab_bits.sh): sha1 of
  * the drop-in forward's nine outputs + the forward->backward live bytes on the bench problem (ungated, and through the object gate),
  * parameters, moments and loss after N fused iterations (per-object job, fused tail), and with the long lists split.
python tools/state_hash.py [cfg] [N]"""
import argparse, hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
args = argparse.Namespace(cfg=cfg, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)


def h(ts):
    m = hashlib.sha1()
    for t in ts:
        m.update(t.detach().contiguous().cpu().numpy().tobytes())
    return m.hexdigest()[:16]


# the drop-in operator (what unchanged DQO-MAP code calls)
import diff_gaussian_rasterization_depth as D
from dqo_harness import mapping
sc = prob["scene"]
t = lambda a: torch.tensor(a, device=dev)
means, opac, scales, rots, shs = t(sc["xyz"]), t(sc["opacity"]).reshape(-1, 1), t(sc["scales"]), t(sc["rotations"]), t(sc["shs"])
for name, tm in (("all tiles", None), ("tile mask", prob["tile_mask"])):
    r = D.GaussianRasterizer(prob["settings"])
    outs = r(means3D=means, opacities=opac, shs=shs, scales=scales, rotations=rots, tile_mask=tm)
    torch.cuda.synchronize()
    print(f"cfg {cfg} drop-in forward ({name}):", h(outs), flush=True)

from dqo_harness.fused_mapping import FusedMapper
mask = prob["render_mask"].to(torch.uint8).contiguous()
for split in (0, 2048):
    fm = FusedMapper(prob["scene"], prob["settings"], dev)
    if prob.get("gate") is not None:
        fm.set_object_gate(prob["gate"][0], prob["gate"][1])
    fm.capture(prob["gt_color"], prob["gt_depth"], mask, tile_mask=prob["tile_mask"], list_split=split)
    for _ in range(n):
        fm.replay()
    torch.cuda.synchronize()
    st = [v for k, v in sorted(fm._params().items())] + [m for k, (m, v) in sorted(fm.state.items())] + [v for k, (m, v) in sorted(fm.state.items())]
    print(f"cfg {cfg} fused, list_split {int(fm._g.ls_fwd)}: {n} iterations: state {h(st)} loss {h([fm.loss])} outputs {h(list(fm._g.out))}", flush=True)
    del fm
    torch.cuda.empty_cache()
