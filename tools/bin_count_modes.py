"""bin_count_kernel's duration differs between processes of one box (50-51 us or 57-60 us on cfg 3): which buffer's placement is it?
Prints, per run of this script, the kernel's time and the addresses of the context buffers:   python tools/bin_count_modes.py [pad_bytes]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench
import _dqo_native as N
from dqo_harness.fused_mapping import FusedMapper

pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0
args = argparse.Namespace(cfg=3, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
junk = torch.empty((pad,), dtype=torch.uint8, device=dev) if pad else None  # shifts what the allocator hands out next
fm = FusedMapper(prob["scene"], prob["settings"], dev).set_object_gate(prob["gate"][0], prob["gate"][1])
fm.capture(prob["gt_color"], prob["gt_depth"], prob["render_mask"].to(torch.uint8), tile_mask=prob["tile_mask"], unroll=1)
for _ in range(20):
    fm.replay()
torch.cuda.synchronize()
N.profile_enable(True); N.profile_collect(reset=True)
for _ in range(20):
    fm.step_static()
torch.cuda.synchronize()
prof = N.profile_collect(reset=True)
g = fm._g
us = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items()}
print(f"pad {pad:9d}  bin_count {us['bin_count_kernel']:5.1f} us  sort_wave {us['tile_sort_wave_kernel']:5.1f}  img 0x{g.img.data_ptr():x}  geom 0x{g.geom.data_ptr():x}  "
      f"binning 0x{g.binning.data_ptr():x}  (img % 2 MiB = {g.img.data_ptr() % (2 << 20)}, % 64 KiB = {g.img.data_ptr() % 65536})")
