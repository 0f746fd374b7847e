"""Generates tests/golden/loss_golden.npz by IMPORTING the reference's own loss helpers (authoring container only):
/root/reference/utils/loss_utils.py (`ssim`, `l1_loss`, `l2_loss`; pure torch, importable as is).  The fixture holds inputs and
the reference's outputs only — never reference source.

  * SSIM branch of Mapping.loss_update (SLAM/multiprocess/mapper.py:843-845: `1 - ssim(image, gt)` when no render mask is given):
    value and autograd gradient w.r.t. the rendered image, on seeded 3x48x64 image pairs (random, smooth, identical);
  * attach loss (mapper.py:812-829) evaluated LITERALLY with the reference's l2_loss on boolean-indexed rows, value and autograd
    gradients w.r.t. the three raw parameter tensors.

Run:  python tests/golden/make_loss_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import utils.loss_utils as lu  # noqa: E402  (the reference module)


def main():
    rng = np.random.default_rng(20260401)
    out = {}
    H, W = 48, 64
    yy, xx = np.meshgrid(np.linspace(0, 1, H), np.linspace(0, 1, W), indexing="ij")
    smooth = np.stack([0.5 + 0.4 * np.sin(6 * xx + 2 * yy), 0.5 + 0.4 * np.cos(5 * yy), 0.3 + 0.5 * xx * yy]).astype(np.float32)
    pairs = [
        (rng.uniform(0, 1, (3, H, W)).astype(np.float32), rng.uniform(0, 1, (3, H, W)).astype(np.float32)),
        (smooth, np.clip(smooth + rng.normal(0, 0.05, smooth.shape), 0, 1).astype(np.float32)),
        (smooth, smooth.copy()),
    ]
    for k, (a, b) in enumerate(pairs):
        ta = torch.tensor(a, requires_grad=True)
        tb = torch.tensor(b)
        s = lu.ssim(ta, tb)
        loss = 1 - s
        loss.backward()
        out[f"ssim{k}_img1"], out[f"ssim{k}_img2"] = a, b
        out[f"ssim{k}_value"] = np.float32(s.item())
        out[f"ssim{k}_grad"] = ta.grad.numpy()
        out[f"ssim{k}_l1"] = np.float32(lu.l1_loss(ta.detach(), tb).item())
        out[f"ssim{k}_l2"] = np.float32(lu.l2_loss(ta.detach(), tb).item())
    # attach loss, literal mapper.py:812-829
    P = 500
    op0 = rng.normal(2.0, 2.5, (P, 1)).astype(np.float32)  # raw opacities: sigmoid crosses 0.9 at 2.197
    x0, s0, q0 = (rng.normal(size=(P, n)).astype(np.float32) for n in (3, 3, 4))
    x, sc, q = (torch.tensor(v + rng.normal(0, 0.01, v.shape).astype(np.float32), requires_grad=True) for v in (x0, s0, q0))
    opacity = torch.sigmoid(torch.tensor(op0))
    attach_mask = (opacity < 0.9).squeeze()
    assert 0 < int(attach_mask.sum()) < P
    attach = 1000 * (lu.l2_loss(sc[attach_mask], torch.tensor(s0)[attach_mask]) + lu.l2_loss(x[attach_mask], torch.tensor(x0)[attach_mask]) +
                     lu.l2_loss(q[attach_mask], torch.tensor(q0)[attach_mask]))
    attach.backward()
    out.update(att_opacity0=op0, att_xyz0=x0, att_scaling0=s0, att_rotation0=q0, att_xyz=x.detach().numpy(), att_scaling=sc.detach().numpy(),
               att_rotation=q.detach().numpy(), att_value=np.float32(attach.item()), att_g_xyz=x.grad.numpy(), att_g_scaling=sc.grad.numpy(),
               att_g_rotation=q.grad.numpy(), att_count=np.int32(int(attach_mask.sum())))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if "value" in k or "count" in k})


if __name__ == "__main__":
    main()
