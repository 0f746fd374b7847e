"""Step statistics of the forward blend kernel on the bench problem.  Needs a library built with profiles/r06_fwd_step_counters.patch
applied and EXTRA=-DFWD_COUNT (the counters are not in the shipped sources): wave steps, steps with a valid pixel, valid lanes,
chunk-waves, entries that passed the quadrant test, list entries offered.      python tools/fwd_step_stats.py [cfg]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench
import diff_gaussian_rasterization_depth as D

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
args = argparse.Namespace(cfg=cfg, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
sc = prob["scene"]
t = lambda a: torch.tensor(a, device=dev)
means, opac, scales, rots, shs = t(sc["xyz"]), t(sc["opacity"]).reshape(-1, 1), t(sc["scales"]), t(sc["rotations"]), t(sc["shs"])
e = torch.empty(0, device=dev)
OFF = 512 + 63 * 256 + 32 * 4  # line 63, word 32 of the statistics lines (dqo_geom_layout)
for name in ("ungated", "gated"):
    if name == "gated":
        outs = D.rasterize_gaussians_gated(means, shs, e, opac, scales, rots, e, prob["tile_mask"], prob["settings"], prob["gate"][0], prob["gate"][1])
    else:
        outs = D.rasterize_gaussians(means, shs, e, opac, scales, rots, e, prob["tile_mask"], prob["settings"])
    torch.cuda.synchronize()
    steps, valid, lanes, chunks, reach, offered = D._last["header"][OFF:OFF + 48].view(torch.int64).cpu().tolist()
    print(f"cfg {cfg} {name}: wave steps {steps}, with a valid pixel {valid} ({valid / max(steps, 1):.3f}), valid lanes per valid step "
          f"{lanes / max(valid, 1):.1f} of 64, chunk-waves {chunks}, entries past the quadrant test {reach} of {offered} offered "
          f"({reach / max(offered, 1):.3f}); steps walked / entries past the test {steps / max(reach, 1):.3f}", flush=True)
