"""Per-kernel averages of the window loop (bench.py's window_benchmark problem): python tools/window_profile.py [cfg] [fractions...]
HIP events around every launch of eager passes over the frames' own calls (FusedMapper.step_static), all rows trained and a seeded share."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dqo-map_amd"))
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
fracs = [float(x) for x in sys.argv[2:]] or [1.0, 0.1]
sys.argv = ["bench.py", "--cfg", cfg]
import bench  # noqa: E402
import torch  # noqa: E402
import _dqo_native as N  # noqa: E402
from dqo_harness import mapping, sharding  # noqa: E402
from dqo_harness.fused_mapping import FusedMapper  # noqa: E402

args = bench.parse()
device = torch.device("cuda", 0)
prob = bench.build_problem(args, 0, 1, device)
K = 5
cams = bench.window_cameras(prob["cam"], K)
frames = []
for cam in cams:
    st = mapping.make_settings(cam, device)
    tgt = mapping.perturbed_target(prob["full"], st, device, prob["cfgd"]["seed"] + 7)
    mask = tgt["pix_obj"] >= 0
    frames.append(dict(settings=st, gt_color=tgt["gt_color"].contiguous(), gt_depth=tgt["gt_depth"].contiguous(),
                       render_mask=mask.to(torch.uint8).contiguous(),
                       tile_mask=torch.tensor(sharding.tile_mask_from_pixel_mask(mask.cpu().numpy()), device=device),
                       pixel_object=tgt["pix_obj"].to(torch.int32).contiguous()))
for frac in fracs:
    fm = FusedMapper(prob["scene"], frames[-1]["settings"], device)
    fm.set_object_gate(prob["gate"][0], frames[-1]["pixel_object"])
    fm.use_block_ticket = not os.environ.get("WP_NO_TICKET")
    if frac < 1.0:
        g_ = torch.Generator(device="cpu").manual_seed(11)
        fm.set_training_rows(trainable=(torch.rand(fm.P, generator=g_) < frac).to(device))
    fm.begin_mapping_call(reset_optimizer=True)
    fm.capture_window(frames, loss_tap=True, fused_tail=True, list_split="auto")
    sched = FusedMapper.window_schedule(200, K, random.Random(0))
    for k in sched[:60]:
        fm.replay(frame=k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in sched:
        fm.replay(frame=k)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / len(sched)
    N.profile_enable(True)
    N.profile_collect(reset=True)
    for k in sched[:40]:
        fm._g = fm._frames[k]
        fm.step_static()
    torch.cuda.synchronize()
    prof = N.profile_collect(reset=True)
    N.profile_enable(False)
    print(f"trained fraction {frac}: {ms:.4f} ms / iteration (replays, one launch each);",
          {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}, "header", fm.header(), flush=True)
    del fm
    torch.cuda.empty_cache()
