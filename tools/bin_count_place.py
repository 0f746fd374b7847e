"""Is bin_count_kernel's bimodal duration (50 vs 58 us on cfg 3, tools/bin_count_modes.py) a property of the PROCESS or of the buffers?
Several mappers in one process, each with freshly allocated context buffers (the allocator is pushed around in between):
    python tools/bin_count_place.py [n]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench
import _dqo_native as N
from dqo_harness.fused_mapping import FusedMapper

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
args = argparse.Namespace(cfg=3, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
keep = []
for i in range(n):
    keep.append(torch.empty(((3 + 5 * i) << 20,), dtype=torch.uint8, device=dev))  # held: the next buffers land elsewhere
    fm = FusedMapper(prob["scene"], prob["settings"], dev).set_object_gate(prob["gate"][0], prob["gate"][1])
    fm.capture(prob["gt_color"], prob["gt_depth"], prob["render_mask"].to(torch.uint8), tile_mask=prob["tile_mask"], unroll=1)
    for _ in range(20):
        fm.replay()
    torch.cuda.synchronize()
    N.profile_enable(True); N.profile_collect(reset=True)
    for _ in range(30):
        fm.step_static()
    torch.cuda.synchronize()
    prof = N.profile_collect(reset=True)
    N.profile_enable(False)
    g = fm._g
    us = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()}
    print(f"mapper {i}: bin_count {us['bin_count_kernel']:5.1f} us  sort_wave {us['tile_sort_wave_kernel']:5.1f}  tail {us['gaussian_tail_kernel']:6.1f}  "
          f"img 0x{g.img.data_ptr():x} geom 0x{g.geom.data_ptr():x} binning 0x{g.binning.data_ptr():x}", flush=True)
    keep.append((g.img, g.geom, g.binning))  # (held too: every mapper gets its own memory)
