"""Worker of tests/test_gpu_window_gloo.py: rank r of a 2-rank gloo group on the box's one GPU.  Every rank builds the same map and the same
three frames, keeps the objects of its shard, and runs the reference's window schedule (mapper.py:570-576, seeded) on its shard with
the packed asynchronous all-reduce of the loss sums after every replay — the N > 1 form of FusedMapper's loop.  It also runs the
UNSHARDED job and checks that (1) the all-reduced losses are the unsharded job's at every iteration, (2) its shard's parameters and
confidence counters end bit for bit where the unsharded job's rows end.      python window_gloo_worker.py <rank> <world> <port> <out>"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dqo-map_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)

import numpy as np
import torch
import torch.distributed as dist

dist.init_process_group("gloo", rank=rank, world_size=world)
import util_rast as U  # noqa: E402
from dqo_harness import mapping, scenes, sharding  # noqa: E402
from dqo_harness.fused_mapping import FusedMapper  # noqa: E402

dev = torch.device("cuda", 0)
cam0, sc = scenes.make_config(3, P=16000)
go = np.asarray(sc["obj_id"], np.int32)
P = len(go)
rng = np.random.default_rng(3)
pert = dict(sc)
pert["xyz"] = (sc["xyz"] + rng.normal(0, 0.004, sc["xyz"].shape)).astype(np.float32)
pert["shs"] = sc["shs"].copy()
pert["shs"][:, 0, :] += rng.normal(0, 0.15, (P, 3)).astype(np.float32)
cams = [scenes.replica_camera(yaw=12.0 - 3.0 * k, pitch=4.0 + 0.7 * k, pos=(0.3 - 0.08 * k, 0.1, -1.85 + 0.05 * k)) for k in (2, 1, 0)]
go_t = torch.tensor(go, device=dev)
frames = []
for cam in cams:
    st = mapping.make_settings(cam, dev)
    with torch.no_grad():
        t0 = mapping.render(st, mapping.GaussianParams(pert, dev).activated())
        hit = t0["depth_index_map"][0]
        po = torch.where(hit >= 0, go_t[hit.long().clamp(min=0)], torch.full_like(hit, -1)).to(torch.int32).contiguous()
        tgt = mapping.render(st, mapping.GaussianParams(pert, dev).activated(), object_gate=(go_t, po))
    frames.append(dict(settings=st, gt_color=tgt["render"].clone(), gt_depth=tgt["depth"].clone(), pixel_object=po))
all_ids = sorted(set(np.unique(go).tolist()))
# object -> rank: the harness's own assignment (the same on every rank)
_, assignment = sharding.shard_scene(sc, rank, world)
mine_ids = [k for k in all_ids if assignment[k] == rank]
in_mine = np.isin(go, mine_ids)
trainable = rng.uniform(size=P) < 0.67
sub = lambda m: {k: (v[m] if hasattr(v, "shape") and v.shape[:1] == (P,) else v) for k, v in sc.items()}


def build(rows, ids, n_attach=None):
    fm = FusedMapper(sub(rows), frames[0]["settings"], dev, attach_count_reducer=(None if n_attach is None else (lambda n: n_attach)))
    fm.set_object_gate(go[rows], frames[0]["pixel_object"])
    fm.set_training_rows(trainable=torch.tensor(trainable[rows], device=dev))
    fm.begin_mapping_call(reset_optimizer=True)
    fr = [dict(f, render_mask=torch.isin(f["pixel_object"], torch.tensor(ids, device=dev, dtype=torch.int32)).to(torch.uint8).contiguous())
          for f in frames]
    return fm, fr


whole, fw = build(np.ones(P, bool), all_ids)
shard, fs = build(in_mine, mine_ids, whole.attach_count)
whole.capture_window(fw, loss_tap=True, fused_tail=True)
shard.capture_window(fs, loss_tap=True, fused_tail=True)
red = sharding.PackedAllReduce([("loss", 3)], "cpu")
sched = FusedMapper.window_schedule(8, 3, random.Random(2))
worst = 0.0
for it, k in enumerate(sched):
    whole.replay(frame=k), shard.replay(frame=k)
    torch.cuda.synchronize()
    assert not (whole.graph_overflowed() or shard.graph_overflowed())
    red.reduce_async(src=shard.loss[:3].double().cpu().float())  # (fp32 payload, like bench.py's)
    red.finish()
    got, want = red.get("loss").double().numpy(), whole.loss[:3].double().cpu().numpy()
    worst = max(worst, float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)))
    np.testing.assert_allclose(got, want, rtol=5e-6)
rows = torch.tensor(np.nonzero(in_mine)[0], device=dev)
for name, pw in whole._params().items():
    assert torch.equal(pw[rows], shard._params()[name]), name
assert torch.equal(whole.confidence[rows], shard.confidence)
json.dump(dict(rank=rank, objects=mine_ids, P_shard=int(in_mine.sum()), iterations=len(sched), worst_rel_loss_diff=worst,
               confidence_max=float(shard.confidence.max())), open(out, "w"))
dist.barrier()
dist.destroy_process_group()
