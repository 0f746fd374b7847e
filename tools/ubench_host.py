"""Host-side cost of the primitives the drop-in op's launch path is made of (microseconds per call, stream idle):
    python tools/ubench_host.py"""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + "/dqo-map_amd"]
import torch
import _dqo_native as N

dev = torch.device("cuda")
H, W, P = 680, 1200, 500000


def t(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


f32 = dict(dtype=torch.float32, device=dev)
u8 = dict(dtype=torch.uint8, device=dev)
x = torch.empty((P, 3), **f32)
slab = torch.empty((7, H, W), **f32)
big = torch.empty((1 << 26,), **u8)
ev = torch.cuda.Event()
ev.record()
pin = torch.empty((8,), dtype=torch.int32).pin_memory()
rows = [
    ("torch.empty((3,H,W)) float32 cuda", lambda: torch.empty((3, H, W), **f32)),
    ("torch.empty(80 MB uint8)", lambda: torch.empty((80 << 20,), **u8)),
    ("slab[0:3] (dim-0 slice)", lambda: slab[0:3]),
    ("big[a:b].view(float32).view(3,H,W)", lambda: big[1024:1024 + 12 * H * W].view(torch.float32).view(3, H, W)),
    ("big.split_with_sizes(12 pieces)", lambda: big.split_with_sizes([1 << 20] * 12 + [(1 << 26) - 12 * (1 << 20)])),
    ("x.contiguous() (already contiguous)", lambda: x.contiguous()),
    ("x.is_contiguous()", lambda: x.is_contiguous()),
    ("x.data_ptr()", lambda: x.data_ptr()),
    ("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream),
    ("with torch.cuda.device(dev): pass", lambda: torch.cuda.device(dev).__enter__() or torch.cuda.device(dev).__exit__(None, None, None)),
    ("N.DqoRastInputs(12 kwargs)", lambda: N.DqoRastInputs(bg=1, means3D=2, shs=3, colors_precomp=None, opacities=4, scales=5, rotations=6,
                                                           cov3D_precomp=None, viewmatrix=7, projmatrix=8, campos=9, tile_mask=10)),
    ("event.record()", lambda: ev.record()),
    ("event.query()", lambda: ev.query()),
    ("pin.copy_(x_slice_view, non_blocking)", lambda: pin.copy_(big[:32].view(torch.int32), non_blocking=True)),
    ("torch.sigmoid(x)", lambda: torch.sigmoid(x)),
]
st = N.DqoRastInputs()


def refill():
    st.bg, st.means3D, st.shs, st.colors_precomp, st.opacities, st.scales = 1, 2, 3, None, 4, 5
    st.rotations, st.cov3D_precomp, st.viewmatrix, st.projmatrix, st.campos, st.tile_mask = 6, None, 7, 8, 9, 10


rows.append(("refill a persistent DqoRastInputs (12 fields)", refill))
lib = N.lib()
rows.append(("ctypes call dqo_abi_version()", lambda: lib.dqo_abi_version()))
rows.append(("ctypes call dqo_rast_geom_bytes(P, W, H)", lambda: lib.dqo_rast_geom_bytes(P, W, H)))
for name, fn in rows:
    print(f"{t(fn):8.2f} us  {name}")
    torch.cuda.synchronize()
