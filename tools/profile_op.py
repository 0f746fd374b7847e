"""Host-side profile of the drop-in op alone (forward + backward with a fixed incoming gradient) on cfg 3: where does the launch
path's time go?      python tools/profile_op.py [exact|lazy|deferred ...]
Per sync mode: wall time per forward + backward pair in steady state, the host's issue time with the stream idle at the start of every
pair (python microseconds of the forward call and of the backward call alone: the GPU is drained before each, so nothing waits on it
except the mode's own synchronisation), and a cProfile table of the lazy mode."""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import argparse
import torch
import bench
import diff_gaussian_rasterization_depth as dgr
from dqo_harness import mapping

modes = sys.argv[1:] or ["exact", "lazy", "deferred"]
args = argparse.Namespace(cfg=3, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=True, as_shard=None)
dev = torch.device("cuda")
prob = bench.build_problem(args, 0, 1, dev)
params = mapping.GaussianParams(prob["scene"], dev)
st, tm = prob["settings"], prob["tile_mask"]
rast = dgr.GaussianRasterizer(raster_settings=st)
gC, gD = torch.randn(3, st.image_height, st.image_width, device=dev), torch.randn(1, st.image_height, st.image_width, device=dev)
act = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in params.activated().items()}
leaves = [act[k].clone().requires_grad_(True) for k in ("xyz", "opacity", "shs", "scales", "rotations")]


def step():
    a = params.activated()
    out = rast(means3D=a["xyz"], opacities=a["opacity"], shs=a["shs"], colors_precomp=None, scales=a["scales"], rotations=a["rotations"],
               cov3D_precomp=None, normal_w=None, tile_mask=tm)
    torch.autograd.backward([out[0], out[1]], [gC, gD])
    for grp in params.param_groups():
        for p in grp["params"]:
            p.grad = None


def op_alone():
    """The operator without torch's activation ops around it: its inputs are leaves.  Returns (forward us, backward us) of host time."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = rast(means3D=leaves[0], opacities=leaves[1], shs=leaves[2], colors_precomp=None, scales=leaves[3], rotations=leaves[4],
               cov3D_precomp=None, normal_w=None, tile_mask=tm)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    torch.autograd.backward([out[0], out[1]], [gC, gD])
    t3 = time.perf_counter()
    for l in leaves:
        l.grad = None
    return (t1 - t0) * 1e6, (t3 - t2) * 1e6


for mode in modes:
    dgr.set_sync_mode(mode)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs = [op_alone() for _ in range(50)][10:]
    f = sorted(h[0] for h in hs)[len(hs) // 2]
    b = sorted(h[1] for h in hs)[len(hs) // 2]
    print(f"[{mode:8s}] steady state {(t2 - t0) / n * 1e3:.3f} ms per forward + backward pair (host issue {(t1 - t0) / n * 1e3:.3f} ms, drained "
          f"{(t2 - t1) * 1e3:.3f} ms after the last call); op alone, stream idle: forward {f:.0f} us, backward {b:.0f} us of host time")
dgr.set_sync_mode("lazy")
torch.autograd.set_multithreading_enabled(False)  # the engine then runs the backward nodes on this thread: cProfile sees them
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
