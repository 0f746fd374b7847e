"""Generates tests/golden/quadric_golden.npz by importing the REFERENCE module
/root/reference/SLAM/multiprocess/quadrics.py in the authoring container (CPU, no GPU).

Only runs where /root/reference exists; the produced .npz (inputs + expected outputs = data) is committed,
the reference source never is.  Import needs two empty stub modules (plyfile, cv2 — used only by the reference's
plotting / PLY helpers) and a module-local redirect of device="cuda" -> "cpu" for the torch classes
(quadrics.py:2021,2039,2148-2157,2188-2192 hard-code "cuda").

Contents:
  single_*  : B random (object, view) pairs -> bbox, loss, grads of Ellipsoid_tensor.forward + bboxes_iou
              (quadrics.py:2178-2220, 2019-2091, 285-290)
  np_*      : numpy twin Ellipsoid(...).project(P).ComputeBbox() in fp64 (quadrics.py:388-408, 148-225)
  adam_*    : 20-step Adam trajectories of the Object_Optimize_only inner loop (quadrics.py:2251-2285) with the
              view schedule made deterministic (first 6 iterations use a recorded random view, then the latest)
  det_boxes : a handful of detection boxes copied from configs/Cube_Diorama/detect_obj/room.json (public config data)
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def import_reference_quadrics():
    for name in ("plyfile", "cv2"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "plyfile":
                m.PlyData = m.PlyElement = object
            sys.modules[name] = m
    sys.path.insert(0, REF)
    import importlib
    q = importlib.import_module("SLAM.multiprocess.quadrics")

    class TorchProxy:
        """module-local stand-in for `torch` that rewrites device="cuda" to "cpu"."""

        def __getattr__(self, k):
            v = getattr(torch, k)
            if k in ("tensor", "eye", "zeros", "ones"):
                def f(*a, **kw):
                    if kw.get("device") == "cuda":
                        kw["device"] = "cpu"
                    return v(*a, **kw)
                return f
            return v

    q.torch = TorchProxy()
    return q


def main():
    q = import_reference_quadrics()
    rng = np.random.default_rng(20241201)
    K = np.array([[600.0, 0, 599.5], [0, 600.0, 339.5], [0, 0, 1]])
    out = {}

    def rand_Rt():
        ang = rng.uniform(-0.3, 0.3, 3)
        cx, cy, cz = np.cos(ang)
        sx, sy, sz = np.sin(ang)
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R = Rz @ Ry @ Rx
        t = rng.uniform(-0.3, 0.3, 3)
        return np.concatenate([R, t[:, None]], 1)

    def ref_single(axes, R, center, P, obs):
        ell = q.Ellipsoid_tensor(axes, R, center, obs)
        Pt = torch.tensor(P, dtype=torch.float32)
        bbox = ell(Pt)
        iou = q.bboxes_iou(list(map(float, obs)), bbox)
        loss = 1.0 - iou
        if loss == 1:
            return bbox.detach().numpy(), 1.0, np.zeros(3), np.zeros((3, 3)), np.zeros(3), 0
        loss.backward()
        return (bbox.detach().numpy(), float(loss), ell.axes_.grad.numpy().copy(), ell.R_.grad.numpy().copy(),
                ell.center_.grad.numpy().copy(), 1)

    B = 24
    rec = {k: [] for k in ("axes", "R", "center", "P", "obs", "bbox", "loss", "g_axes", "g_R", "g_center", "valid", "np_bbox")}
    # known-answer case of SURVEY.md §8(c) first
    cases = [(np.array([0.3, 0.2, 0.1]), np.eye(3), np.array([0.1, -0.05, 2.0]),
              K @ np.concatenate([np.eye(3), np.zeros((3, 1))], 1), np.array([600.0, 280, 720, 400]))]
    while len(cases) < B:
        axes = rng.uniform(0.05, 0.5, 3)
        R = rand_Rt()[:, :3] + rng.normal(0, 0.02, (3, 3))  # not orthonormal on purpose (free 9-dof matrix)
        center = np.array([rng.uniform(-1, 1), rng.uniform(-0.6, 0.6), rng.uniform(1.5, 4.0)])
        P = K @ rand_Rt()
        nb = q.Ellipsoid(axes, R, center).project(P).ComputeBbox()
        jitter = rng.normal(0, 0.15, 4) * np.array([nb[2] - nb[0], nb[3] - nb[1]] * 2)
        obs = nb + jitter
        if len(cases) % 7 == 3:
            obs = nb + 5000.0  # disjoint boxes -> IoU 0 -> loss == 1 -> step skipped
        cases.append((axes, R, center, P, obs))
    for axes, R, center, P, obs in cases:
        bbox, loss, ga, gR, gc, valid = ref_single(axes, R, center, P, obs)
        nb = q.Ellipsoid(np.float64(axes), np.float64(R), np.float64(center)).project(np.float64(P)).ComputeBbox()
        for k, v in zip(rec.keys(), (axes, R, center, P, obs, bbox, loss, ga, gR, gc, valid, nb)):
            rec[k].append(np.asarray(v))
    for k, v in rec.items():
        out["single_" + k] = np.stack(v)

    # ---- Adam trajectories (quadrics.py:2251-2285) ----
    import torch.optim as optim
    n_obj, n_views, n_it = 4, 5, 20
    traj = {k: [] for k in ("axes0", "R0", "center0", "Pviews", "obsviews", "sched", "axes", "R", "center", "loss")}
    for o in range(n_obj):
        axes = rng.uniform(0.1, 0.4, 3)
        R = rand_Rt()[:, :3]
        center = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.3, 0.3), rng.uniform(2.0, 3.5)])
        Ps, obss = [], []
        for v in range(n_views):
            P = K @ rand_Rt()
            nb = q.Ellipsoid(axes, R, center).project(P).ComputeBbox()
            obss.append(nb + rng.normal(0, 0.08, 4) * np.array([nb[2] - nb[0], nb[3] - nb[1]] * 2))
            Ps.append(P)
        if o == 3:
            obss[2] = obss[2] + 9000.0  # one view with IoU 0 to exercise the skipped step
        sched = [int(rng.integers(0, n_views)) if it <= n_it / 4 else n_views - 1 for it in range(n_it)]
        if o == 3:
            sched[1] = 2
        ell = q.Ellipsoid_tensor(axes, R, center, obss)
        opt = optim.Adam([{"params": [ell.axes_], "lr": 0.01}, {"params": [ell.center_], "lr": 0.001},
                          {"params": [ell.R_], "lr": 0.01}], eps=1e-15)
        losses = []
        for it in range(n_it):
            opt.zero_grad()
            P = torch.tensor(Ps[sched[it]], dtype=torch.float32)
            bbox = ell(P)
            iou = q.bboxes_iou(list(map(float, obss[sched[it]])), bbox)
            loss = 1.0 - iou
            losses.append(float(loss))
            if loss == 1:
                continue
            loss.backward()
            opt.step()
        for k, v in zip(traj.keys(), (axes, R, center, np.stack(Ps), np.stack(obss), np.array(sched, np.int32),
                                      ell.axes_.detach().numpy().copy(), ell.R_.detach().numpy().copy(),
                                      ell.center_.detach().numpy().copy(), np.array(losses))):
            traj[k].append(np.asarray(v))
    for k, v in traj.items():
        out["adam_" + k] = np.stack(v)

    # a handful of real detection boxes (rows of public config data, not the file)
    det = json.load(open(os.path.join(REF, "configs/Cube_Diorama/detect_obj/room.json")))
    boxes = []
    for fr in det[:6]:
        for d in fr["detections"][:3]:
            boxes.append(d["bbox"])
    out["det_boxes"] = np.array(boxes, np.float64)
    out["K"] = K
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quadric_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
