"""Does the tail's time depend on WHERE the three SH arrays (parameters, first and second moments: 96 MB each on cfg 3, read and written at
the same element index by every lane of the Adam pass) sit relative to each other?  One buffer, the three arrays carved out of it at chosen
relative byte offsets; the captured iteration timed per kernel (HIP events around eager passes).      python tools/sh_placement.py [cfg]"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import bench
import _dqo_native as N
from dqo_harness.fused_mapping import FusedMapper

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
args = argparse.Namespace(cfg=cfg, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=False, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
mask = prob["render_mask"].to(torch.uint8).contiguous()
KB, MB = 1024, 1024 * 1024
combos = [None, (0, 0), (4 * KB, 8 * KB), (64 * KB, 128 * KB), (MB + 4 * KB, 2 * MB + 8 * KB), (256, 512), (2 * MB, 4 * MB)]
for combo in combos:
    fm = FusedMapper(prob["scene"], prob["settings"], dev)
    if prob.get("gate") is not None:
        fm.set_object_gate(prob["gate"][0], prob["gate"][1])
    if combo is not None:
        n = fm.shs.numel()
        nbytes = n * 4
        step = ((nbytes + 2 * MB - 1) // (2 * MB)) * 2 * MB  # the arrays 2 MB-aligned apart, then the chosen offsets on top
        big = torch.zeros(3 * step + 16 * MB, dtype=torch.uint8, device=dev)
        def carve(off):
            return big[off:off + nbytes].view(torch.float32).view(fm.shs.shape)
        p, m, v = carve(0), carve(step + combo[0]), carve(2 * step + combo[1])
        p.copy_(fm.shs)
        fm.shs = p
        fm.state["shs"] = (m, v)
        fm._keep_big = big
    fm.capture(prob["gt_color"], prob["gt_depth"], mask, tile_mask=prob["tile_mask"])
    for _ in range(40):
        fm.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fm.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    N.profile_enable(True)
    N.profile_collect(reset=True)
    for _ in range(30):
        fm.step_static()
    torch.cuda.synchronize()
    prof = N.profile_collect(reset=True)
    N.profile_enable(False)
    tail = prof.get("gaussian_tail_kernel", (0, 1))
    ptrs = (fm.shs.data_ptr(), fm.state["shs"][0].data_ptr(), fm.state["shs"][1].data_ptr())
    print(f"offsets {combo}: {ms:.4f} ms / iteration, tail {tail[0] / max(tail[1], 1) * 1e3:.1f} us; m - p = {ptrs[1] - ptrs[0]:#x}, v - p = {ptrs[2] - ptrs[0]:#x}", flush=True)
    del fm
    torch.cuda.empty_cache()
