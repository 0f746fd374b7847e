"""Error of dqo_map_ssim_fwd_bwd and of the eager fp32 statement against float64 on the reference fixture (tests/golden/loss_golden.npz)."""
import os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "dqo-map_amd")]
from dqo_harness import mapping, fused_ops
d = np.load(os.path.join(R, "tests/golden/loss_golden.npz"))
F = torch.nn.functional
for k in range(3):
    a = torch.tensor(d[f"ssim{k}_img1"], device="cuda", requires_grad=True)
    b = torch.tensor(d[f"ssim{k}_img2"], device="cuda")
    s = fused_ops.fused_ssim(a, b)
    (g,) = torch.autograd.grad(1 - s, [a])
    a64 = a.detach().double().requires_grad_(True)
    w64 = mapping._gaussian_window(11, 1.5, 3, "cuda").double()
    c = lambda x: F.conv2d(x[None], w64, padding=5, groups=3)
    mu1, mu2 = c(a64), c(b.double())
    s1, s2, s12 = c(a64 * a64) - mu1 * mu1, c(b.double() ** 2) - mu2 * mu2, c(a64 * b.double()) - mu1 * mu2
    r = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    (g64,) = torch.autograd.grad(1 - r, [a64])
    gold_v, gold_g = float(d[f"ssim{k}_value"]), torch.tensor(d[f"ssim{k}_grad"], device="cuda").double()
    print(f"pair {k}: value fp64 {r.item():.9f} hip {s.item():.9f} (err {abs(s.item()-r.item()):.2e}) fixture {gold_v:.9f} (err {abs(gold_v-r.item()):.2e})")
    print(f"   grad max|fp64| {g64.abs().max().item():.3e}  hip err {(g.double()-g64).abs().max().item():.3e}  fixture err {(gold_g-g64).abs().max().item():.3e}"
          f"  hip vs fixture {(g.double()-gold_g).abs().max().item():.3e}")

# time of the SSIM term (value + gradient) at the mapping loop's image sizes: three launches against the eager autograd statement
for H, W in ((480, 640), (680, 1200)):
    g = torch.Generator(device="cuda").manual_seed(1)
    gt = torch.rand((3, H, W), device="cuda", generator=g)
    img = (gt + 0.1 * torch.randn((3, H, W), device="cuda", generator=g)).clamp(0, 1).requires_grad_(True)

    def fused():
        s = fused_ops.fused_ssim(img, gt)
        return torch.autograd.grad(1 - s, [img])

    def eager():
        s = mapping.ssim(img, gt)
        return torch.autograd.grad(1 - s, [img])

    for name, fn in (("fused", fused), ("eager", eager)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{W}x{H} {name}: {e0.elapsed_time(e1) / 50 * 1000:.1f} us per value+gradient (includes host time of the wrapper)")
