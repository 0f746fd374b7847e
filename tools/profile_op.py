"""Host-side profile of the drop-in op alone (forward + backward with a fixed incoming gradient) on cfg 3: where does the launch
path's time go?      python tools/profile_op.py"""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import argparse
import torch
import bench
import diff_gaussian_rasterization_depth as dgr
from dqo_harness import mapping

args = argparse.Namespace(cfg=3, P=None, view="room", scaling="strong", shard_by="work", no_object_gate=True, as_shard=None)
dev = torch.device("cuda")
dgr.set_sync_mode("lazy")
prob = bench.build_problem(args, 0, 1, dev)
params = mapping.GaussianParams(prob["scene"], dev)
st, tm = prob["settings"], prob["tile_mask"]
rast = dgr.GaussianRasterizer(raster_settings=st)
gC, gD = torch.randn(3, st.image_height, st.image_width, device=dev), torch.randn(1, st.image_height, st.image_width, device=dev)


def step():
    a = params.activated()
    out = rast(means3D=a["xyz"], opacities=a["opacity"], shs=a["shs"], colors_precomp=None, scales=a["scales"], rotations=a["rotations"],
               cov3D_precomp=None, normal_w=None, tile_mask=tm)
    torch.autograd.backward([out[0], out[1]], [gC, gD])
    for grp in params.param_groups():
        for p in grp["params"]:
            p.grad = None


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue time per call pair: %.3f ms, drained after %.3f ms" % ((t1 - t0) * 10, (time.perf_counter() - t1) * 1e3))
torch.autograd.set_multithreading_enabled(False)  # the engine then runs the backward nodes on this thread: cProfile sees them
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
