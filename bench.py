#!/usr/bin/env python3
"""Mapping-iteration benchmark of the MI355X rasteriser path (BASELINE.json metric).

One step = one mapping iteration of DQO-MAP's local_optimize loop (SLAM/multiprocess/mapper.py:531-605) on synthetic
data of BASELINE config 3: activations -> rasteriser forward -> loss (0.8 L1 colour + 1.0 depth L1 on the object masks,
mapper.py:836-875) -> rasteriser backward -> Adam step over the six parameter groups (gaussian_pointcloud.py:331-378).
500k Gaussians, 1200x680, 8 object ids; inputs resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--cfg 3] [--P 500000]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL): every rank owns its own object
shard (weak scaling: fixed Gaussians per GPU, per-object losses on disjoint masks), the only exchange is one packed
all-reduce of the per-iteration loss scalars.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dqo-map_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--P", type=int, default=None)
    ap.add_argument("--sync-mode", default="lazy", choices=("lazy", "exact"))
    ap.add_argument("--path", default="fused", choices=("fused", "dropin"),
                    help="fused: dqo_harness.FusedMapper (activation / loss / Adam kernels of row f2 around the op); "
                         "dropin: autograd through the drop-in op + torch.optim.Adam, exactly what unchanged DQO-MAP code runs")
    ap.add_argument("--no-graph", action="store_true", help="fused path: issue the kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample-P", type=int, default=500_000)
    return ap.parse_args()


def object_masks(cam, hit_ids, obj_id, n_objects, device):
    """Per-object screen masks (16-px tile granularity): a pixel belongs to the object of the Gaussian that
    fixes its depth in the initial render; the shard's render mask is the union of its objects' masks."""
    ids = hit_ids[0].long().clamp(min=0)
    pix_obj = obj_id[ids]
    pix_obj[hit_ids[0] < 0] = -1
    return pix_obj


def build_problem(args, rank, world, device):
    from dqo_harness import scenes, mapping
    cfgd = dict(scenes.CONFIGS[args.cfg])
    P = args.P or cfgd["P"]
    # weak scaling: every rank gets its own shard of P Gaussians (distinct objects / seed), same camera and image size
    cfgd["seed"] = cfgd["seed"] + 1000 * rank
    if args.cfg == 1:
        cam, scene = scenes.make_config(1, P=P)
    else:
        cam = scenes.replica_camera(cfgd["W"], cfgd["H"], cfgd["fx"], cfgd["fx"], cfgd["cx"], cfgd["cy"])
        scene = scenes.surfel_room(cfgd["seed"], P, n_objects=cfgd["n_objects"], rest_sigma=cfgd["rest_sigma"])
    if os.environ.get("DQO_BENCH_MORTON"):  # experiment: storage order of the Gaussians = Morton order of their centres
        q = scene["xyz"]
        lo, hi = q.min(0), q.max(0)
        u = np.clip(((q - lo) / (hi - lo + 1e-9) * 1023).astype(np.uint64), 0, 1023)
        def spread(x):
            x = (x | (x << 16)) & 0x030000FF
            x = (x | (x << 8)) & 0x0300F00F
            x = (x | (x << 4)) & 0x030C30C3
            x = (x | (x << 2)) & 0x09249249
            return x
        code = spread(u[:, 0]) | (spread(u[:, 1]) << 1) | (spread(u[:, 2]) << 2)
        perm = np.argsort(code, kind="stable")
        scene = {k: (v[perm] if hasattr(v, "shape") and v.shape[:1] == (len(perm),) else v) for k, v in scene.items()}
    params = mapping.GaussianParams(scene, device)
    settings = mapping.make_settings(cam, device)
    # target = render of a perturbed copy, so the gradients are non-trivial
    rng = np.random.default_rng(cfgd["seed"] + 7)
    pert = dict(scene)
    pert["xyz"] = (scene["xyz"] + rng.normal(0, 0.004, scene["xyz"].shape)).astype(np.float32)
    pert["shs"] = scene["shs"].copy()
    pert["shs"][:, 0, :] += rng.normal(0, 0.15, (P, 3)).astype(np.float32)
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(pert, device).activated())
        gt_color, gt_depth = tgt["render"].clone(), tgt["depth"].clone()
        pix_obj = object_masks(cam, tgt["depth_index_map"], params.obj_id, cfgd["n_objects"], device)
        render_mask = pix_obj >= 0  # union of the object masks of this shard
    return cam, scene, params, settings, gt_color, gt_depth, render_mask, cfgd, P


def make_step(params, settings, gt_color, gt_depth, render_mask, opt, loss_buf, world):
    from dqo_harness import mapping

    def step():
        out = mapping.render(settings, params.activated())
        loss, parts = mapping.mapping_loss(out, gt_color, gt_depth, render_mask=render_mask)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        loss_buf.put("total", parts["total_loss"])
        loss_buf.put("color", parts["color_loss"])
        loss_buf.put("depth", parts["depth_loss"])
        # the path's only exchange: ONE packed fp32 all-reduce of the quantities shared across object shards
        loss_buf.reduce()
        return out

    return step


def make_fused_step(scene, settings, device, gt_color, gt_depth, render_mask, loss_buf, use_graph=True, world=1):
    from dqo_harness.fused_mapping import FusedMapper
    fm = FusedMapper(scene, settings, device)
    mask_u8 = render_mask.to(torch.uint8).contiguous()

    def step_eager():
        out = fm.step(gt_color, gt_depth, mask_u8)
        loss_buf.buf[:3].copy_(fm.loss[:3])
        loss_buf.reduce()  # ONE packed all-reduce per iteration (no-op at world size 1)
        return {"radii": out[8]}

    step_eager.mapper = fm
    if not use_graph:
        return step_eager, step_eager
    try:
        fm.capture(gt_color, gt_depth, mask_u8)  # the whole iteration as one hipGraph over persistent buffers
    except Exception as e:  # e.g. a runtime that refuses the capture: the eager path is the same arithmetic
        print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running the fused path eagerly", file=sys.stderr)
        torch.cuda.synchronize()
        return step_eager, step_eager

    def step_graph():
        out = fm.replay()
        if world > 1:  # the per-iteration collective of the sharded path stays outside the graph and off its critical path
            loss_buf.reduce_async(src=fm.loss[:3])
        return {"radii": out[8]}

    def finish():
        if fm.graph_overflowed():
            raise RuntimeError("captured graph: instance capacity exceeded, outputs invalid")
        if world == 1:
            loss_buf.buf[:3].copy_(fm.loss[:3])

    def step_static():  # the graph's own calls issued eagerly: what the per-kernel profile pass times
        out = fm.step_static()
        loss_buf.buf[:3].copy_(fm.loss[:3])
        loss_buf.reduce()
        return {"radii": out[8]}

    step_graph.finish = finish
    step_graph.static = step_static
    step_graph.mapper = step_eager.mapper = fm
    return step_graph, step_eager


def cpu_baseline(args, cam, scene, P_sample):
    """Single-thread CPU oracle (kind 'port': the reference has no CPU renderer) on a bounded sample of the
    same workload: ONE forward + backward at the same image size with the first P_sample Gaussians of the scene."""
    from oracle import oracle_lib as ol
    ol.build()
    sub = {k: v[:P_sample] for k, v in scene.items()}
    st = ol.RastSettings(cam.W, cam.H, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, normal_threshold=float(np.cos(np.deg2rad(60.0))))
    o = ol.OracleRasterizer(np.float32)
    t0 = time.time()
    r = o.forward(st, sub["xyz"], sub["opacity"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center,
                  shs=sub["shs"], scales=sub["scales"], rotations=sub["rotations"])
    t1 = time.time()
    o.backward(np.ones((3, cam.H, cam.W), np.float32), np.ones((1, cam.H, cam.W), np.float32))
    t2 = time.time()
    return dict(value=1.0 / (t2 - t0), unit="iter/s", cores=1, kind="port",
                sample=f"1 iteration (raster fwd {t1 - t0:.2f}s + bwd {t2 - t1:.2f}s, no loss/Adam) of the single-thread C++ oracle on "
                       f"the first {P_sample} Gaussians of the workload at {cam.W}x{cam.H} (N={r.num_rendered} instances)")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(os.environ.get("DQO_BENCH_BACKEND", "nccl"))  # "nccl" is RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP rasteriser has no CPU path")
    if os.environ.get("DQO_BENCH_BACKEND") == "gloo":  # code-path rehearsal of the N-rank run on fewer GPUs (not a measurement)
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    import _dqo_native as N
    import diff_gaussian_rasterization_depth as dgr
    from dqo_harness import mapping
    N.lib()
    dgr.set_sync_mode(args.sync_mode)

    cam, scene, params, settings, gt_color, gt_depth, render_mask, cfgd, P = build_problem(args, rank, world, device)
    opt = mapping.make_optimizer(params)
    from dqo_harness.sharding import PackedAllReduce
    loss_buf = PackedAllReduce([("total", 1), ("color", 1), ("depth", 1)], device)
    step_dropin = make_step(params, settings, gt_color, gt_depth, render_mask, opt, loss_buf, world)
    step_fused, step_fused_eager = make_fused_step(scene, settings, device, gt_color, gt_depth, render_mask, loss_buf,
                                                   use_graph=not args.no_graph, world=world)
    step = step_fused if args.path == "fused" else step_dropin

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    loss_buf.finish()  # outstanding asynchronous all-reduces of the sharded path (inside the timed region)
    sync_all()
    dt = time.perf_counter() - t0
    if hasattr(step, "finish"):
        step.finish()  # graph path: overflow check + loss read-back, outside the timed region
    if args.sync_mode == "lazy":
        dgr._verify_pending(block=True)  # raises if any timed iteration overflowed its instance capacity
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    loss_now = [round(float(x) / world, 6) for x in loss_buf.buf.tolist()[:3]]
    # the other path, timed the same way (single GPU only), so both numbers come from one run
    alt = None
    if world == 1:
        other = step_dropin if args.path == "fused" else step_fused
        for _ in range(max(2, args.warmup // 2)):
            other()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            other()
        torch.cuda.synchronize()
        dt_alt = time.perf_counter() - t1
        if args.sync_mode == "lazy":
            dgr._verify_pending(block=True)
        alt = {"path": "dropin" if args.path == "fused" else "fused", "value": round(args.steps / dt_alt, 3), "unit": "iter/s",
               "ms_per_step": round(dt_alt / args.steps * 1e3, 4)}

    # workload statistics of the last iteration (reported next to every timing)
    n_vis = int((out["radii"] > 0).sum().item())
    stats = dict(P=P, P_visible=n_vis)
    fm_ = getattr(step_fused, "mapper", None)
    if args.path == "fused" and fm_ is not None and fm_.moment_live is not None:
        # exact sparse Adam (DqoAdamStep.moment_live): Gaussians with all-zero moments and no gradient are fixed points of the
        # update and are skipped; bitwise equal to the dense update (tests/test_gpu_fused_mapping.py)
        stats["adam"] = "exact-sparse"
        stats["adam_rows_touched"] = int(fm_.moment_live.sum().item())

    roofline = None
    kernels = None
    if not args.no_roofline:  # on every rank: the step function contains the per-iteration collective
        # per-kernel durations with HIP events on the launch stream, over the same step function
        N.profile_enable(True)
        N.profile_collect(reset=True)
        torch.cuda.synchronize()
        ksteps = min(args.steps, 20)
        # HIP events cannot be recorded inside a graph replay: the profile pass issues the graph's own calls eagerly
        step_prof = getattr(step_fused, "static", step_fused_eager) if args.path == "fused" else step_dropin
        for _ in range(ksteps):
            step_prof()
        torch.cuda.synchronize()
        prof = N.profile_collect(reset=True)
        N.profile_enable(False)
        kernels = {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in prof.items()}  # average microseconds per launch
        # read N / active tiles of the current state
        cap = dgr._cap_hint.get((device.index, P, cam.W, cam.H))
        stats["kernel_us"] = kernels
        dom = max(prof.items(), key=lambda kv: kv[1][0])
        dom_name, dom_ms = dom[0], dom[1][0] / max(dom[1][1], 1)
        # workload counts of the current state from the device header of the last forward
        hdr = fm_.header() if (args.path == "fused" and fm_ is not None and getattr(fm_, "_g", None) is not None) else dgr.last_header()
        n_inst, n_cand = hdr["num_rendered"], hdr["num_candidates"]
        HWa = cam.W * cam.H
        stats.update(N_instances=n_inst, N_candidates=n_cand, max_tile_list=hdr["max_tile_count"])
        # algorithmic bytes per launch (DESIGN.md "Kernels"): what the kernel must move at minimum, per unit
        # (instance = (Gaussian, tile) list entry; Gaussian; pixel) x the units of this launch
        alg = {
            "preprocess_kernel": 236 * n_vis + 12 * (P - n_vis) + 8 * P + 88 * n_vis,   # params of visible, xyz of culled; radii, rect; 5 tables + rect
            "bin_count_kernel": 40 * n_vis + 12 * n_inst,                                # rect + 2 tables per visible; (tile, rank, id) per instance
            "bin_place_kernel": 24 * n_inst,                                             # info + id in, key + slot out
            "tile_sort_wave_kernel": 20 * n_inst, "tile_sort_kernel": 20 * n_inst,       # key + slot in, id + slot out
            "blend_forward_kernel": 40 * n_inst + 36 * HWa,                              # id + 2 records (+ rgb for survivors), live bytes; 9 output planes
            "blend_backward_kernel": 120 * n_inst + 32 * HWa,                            # id, slot, 3 records in, one 64-byte gradient record out; 8 pixel planes
            "record_sum_kernel": 68 * n_inst + 64 * n_vis,                               # gradient records in, one summed record per visible Gaussian out
            # summed record, params, tables in; gradient rows out (fused path: only the rows of visible Gaussians are written)
            "gaussian_backward_kernel": (64 + 236 + 96) * n_vis + 284 * (n_vis if args.path == "fused" else P),
            "loss_reduce_kernel": 36 * HWa, "loss_grad_kernel": 52 * HWa,
            # 59 floats x (param, m, v in and out) of the Gaussians Adam touches (all of them in dense mode) + the gradient rows
            "adam_kernel": 236 * 6 * stats.get("adam_rows_touched", P) + 236 * n_vis,
        }
        bytes_dom = alg.get(dom_name, 0)
        achieved = bytes_dom / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # memory-side bytes per launch of the dominant kernel, from the separate rocprofv3 --pmc passes committed under profiles/
        # (tools/pmc_hbm.sh; counters cannot be collected from inside this process)
        traffic, valu_frac, valu_busy = None, None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_traffic.json")) as fh:
                tk = json.load(fh)["kernels"].get(dom_name)
            if tk is not None and args.cfg == 3 and args.P is None:
                traffic = int(tk["read_bytes"] + tk["write_bytes"])
                if "valu_insts" in tk:  # VALU-issue occupancy of the dominant kernel: wave instructions / (CUs x clock x duration)
                    valu_frac = tk["valu_insts"] / (256 * 2.4e9 * dom_ms * 1e-3)
                if "valu_active_quadcycles" in tk:  # SQ_ACTIVE_INST_VALU (4-cycle units, summed over the chip) / all SIMD cycles
                    valu_busy = tk["valu_active_quadcycles"] * 4 / (1024 * 2.4e9 * dom_ms * 1e-3)
        except OSError:
            pass
        roofline = dict(bound="hbm", kernel=dom_name, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, avg_launch_us=round(dom_ms * 1e3, 2),
                        algorithmic_bytes=int(bytes_dom), valu_issue_frac=None if valu_frac is None else round(valu_frac, 3),
                        valu_busy_frac=None if valu_busy is None else round(valu_busy, 3),
                        note="the blend kernels are VALU bound, not HBM bound: valu_busy_frac = SQ_ACTIVE_INST_VALU over all SIMD cycles "
                             "of the launch, valu_issue_frac = SQ_INSTS_VALU x 4 cycles over the same (profiles/README.md); per-kernel "
                             "GB/s of every kernel: config.kernel_gbs",
                        kernel_gbs={k: round(alg[k] / (us * 1e-6) / 1e9, 1) for k, us in kernels.items() if k in alg and us > 0})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, cam, scene, min(args.cpu_sample_P, P))

    if rank == 0:
        value = world * args.steps / dt
        line = {
            "metric": "mapping iters/sec (fwd+bwd raster) @ 500k Gaussians 1200x680",
            "value": round(value, 3), "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg{args.cfg}: surfel room, {P} Gaussians/GPU, {cam.W}x{cam.H}, {cfgd['n_objects']} object ids, "
                                   "SH degree 3, per-object masked loss (0.8 L1 colour + 1.0 depth L1), raster fwd+bwd + Adam (6 groups); path=" + args.path
                                   + ("" if args.path != "fused" or args.no_graph else " (one hipGraph replay per iteration)"),
                       "shards": world, "sync_mode": args.sync_mode, **stats},
            "loss": loss_now, "path": args.path,
        }
        if alt is not None:
            line["other_path"] = alt
        if roofline is not None:
            line["roofline"] = roofline
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
