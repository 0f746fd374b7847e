"""GPU parity sweep over the OPERATOR'S OWN PARAMETERS (VERDICT r4, weak #1c): every other GPU parity test runs util_rast's one setting
(scale_modifier 1, opaque_threshold 0.6, depth_threshold 1, normal_threshold cos 60 deg, color_sigma 3, T_threshold 1e-4, a principal
point < 1 px off centre), while the reference itself is run with others:
    renderer_opaque_threshold_eval 0.5 / 0.05      configs/base.yaml:122, configs/Cube_Diorama_base.yaml:38 (applied by metric.py:138)
    renderer_normal_threshold 80 deg               arguments/__init__.py:181
    principal points up to 16 px off centre        configs/orb_config/tum1.yaml:10-11 (TUM fr1: 517.3 / 516.5 / 318.6 / 255.3 at 640x480)
and quirk B6 (hit-depth test with smax * scale_modifier in the forward, forward.cu:73,798, but the raw smax in the backward,
backward.cu:1009,1018) and quirk B5 (pixel = ndc * S / 2 + c, ray = ((u - cx) / fx, ...), d(dx)/d(ndc) = W / 2 whatever cx is:
auxiliary.h:44-47, forward.cu:92-100, backward.cu:904-905) only separate from the default case when scale_modifier != 1 and when the
principal point is off centre.  Each case: the full protocol of util_rast.parity_case (forward 1e-4, flipped pixels within budget and
taken out of the incoming gradient on both sides, gradients 1e-3 with the fp64 oracle beside) on a cfg-1-sized cloud (the reference's
own CPU-runnable case) and, where the parameter only matters on planar surfels seen obliquely, on the surfel room."""
import json
import math
import os

import numpy as np
import pytest

from dqo_harness import scenes
import util_rast as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import _dqo_native
    _dqo_native.lib()
    return torch


def _dL(cam, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(3, cam.H, cam.W)).astype(np.float32), rng.normal(size=(1, cam.H, cam.W)).astype(np.float32)


def _cloud(P=8000, cam=None):
    """cfg 1's frustum cloud (optionally seen through another camera's intrinsics: the cloud fills THAT camera's frustum)."""
    if cam is None:
        return scenes.make_config(1, P=P)
    return cam, scenes.frustum_cloud(scenes.CONFIGS[1]["seed"], P, cam)


def _room(P=20000, W=640, H=480, fx=400.0, cx=None, cy=None):
    """The surfel room of cfg 2 .. 5 (planar patches seen obliquely: where the hit-depth branch and the normal test decide) at a size
    the serial fp64 oracle finishes in seconds."""
    cam = scenes.replica_camera(W, H, fx, fx, (W - 1) / 2 if cx is None else cx, (H - 1) / 2 if cy is None else cy)
    return cam, scenes.surfel_room(scenes.CONFIGS[2]["seed"], P, n_objects=4, rest_sigma=0.0)


def _report(name, fs, gs, extra=None):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_param_sweep.json")
    prev = json.load(open(path)) if os.path.exists(path) else {}
    row = dict(mismatch_px=fs.get("mismatch_px"), color=fs.get("color"), depth=fs.get("depth"), T_map=fs.get("T_map"),
               grads={k: dict(max_row_err=v["max_row_err"], q99=v["q99"], beyond_bar=v["beyond_bar"], explained=v["explained"],
                              unexplained=v["unexplained"]) for k, v in gs.items() if isinstance(v, dict) and "max_row_err" in v})
    if extra:
        row.update(extra)
    prev[name] = row
    json.dump(prev, open(path, "w"), indent=1)


CASES = [
    # name, scene, settings
    ("scale_modifier_0.5", "cloud", dict(scale_modifier=0.5)),
    ("scale_modifier_2.0", "cloud", dict(scale_modifier=2.0)),
    ("scale_modifier_0.5_room", "room", dict(scale_modifier=0.5)),   # quirk B6 needs hit pixels near the depth test's edge
    ("scale_modifier_2.0_room", "room", dict(scale_modifier=2.0)),
    ("opaque_threshold_0.05", "cloud", dict(opaque_threshold=0.05)),
    ("opaque_threshold_0.5", "cloud", dict(opaque_threshold=0.5)),
    ("opaque_threshold_0.05_room", "room", dict(opaque_threshold=0.05)),
    ("normal_threshold_cos80", "room", dict(normal_threshold=math.cos(math.radians(80.0)))),
    ("normal_threshold_cos80_cloud", "cloud", dict(normal_threshold=math.cos(math.radians(80.0)))),
    ("depth_threshold_0.3", "room", dict(depth_threshold=0.3)),
    ("depth_threshold_0.3_cloud", "cloud", dict(depth_threshold=0.3)),
    ("T_threshold_1e-2", "cloud", dict(T_threshold=1e-2)),
    ("T_threshold_1e-2_room", "room", dict(T_threshold=1e-2)),
    ("color_sigma_2.0", "cloud", dict(color_sigma=2.0)),
    ("everything_at_once", "room", dict(scale_modifier=1.5, opaque_threshold=0.5, normal_threshold=math.cos(math.radians(80.0)),
                                        depth_threshold=0.3, T_threshold=1e-3, color_sigma=2.5, bg=(0.1, 0.2, 0.3))),
]


@pytest.mark.parametrize("name,scene,kw", CASES, ids=[c[0] for c in CASES])
def test_operator_parameter(torch_cuda, oracle, name, scene, kw):
    cam, sc = _cloud() if scene == "cloud" else _room()
    fs, gs = U.parity_case(oracle, cam, sc, _dL(cam, 21), fp64=True, **kw)
    _report(name, fs, gs)
    # the parameter really changes the result: the default setting renders something else (guards against a setting that is dropped
    # on the way to the kernels on BOTH sides)
    ref, _ = U.run_hip(cam, sc)
    cur, _ = U.run_hip(cam, sc, **kw)
    changed = max(np.abs(ref["color"] - cur["color"]).max(), np.abs(ref["depth"] - cur["depth"]).max(),
                  np.abs(ref["T_map"] - cur["T_map"]).max(), float((ref["hit_depth"] != cur["hit_depth"]).mean()))
    assert changed > 1e-3, f"{name}: the setting changes nothing in this scene"


PP = [
    # name, W, H, fx, fy, cx, cy
    ("tum_fr1", 640, 480, 517.3, 516.5, 318.6, 255.3),          # configs/orb_config/tum1.yaml:8-11
    ("ours", 640, 480, 605.2, 604.9, 312.7, 246.9),              # 7 px off centre in both axes (configs/orb_config/ours.yaml:8-11 order of magnitude)
    ("shifted_40_25", 640, 480, 525.0, 525.0, 360.0, 215.0),     # (W/2 + 40, H/2 - 25)
]


@pytest.mark.parametrize("name,W,H,fx,fy,cx,cy", PP, ids=[c[0] for c in PP])
def test_off_centre_principal_point(torch_cuda, oracle, name, W, H, fx, fy, cx, cy):
    """Quirk B5: the projection matrix knows nothing of (cx, cy) (getProjectionMatrix is symmetric), the pixel mapping adds it, the
    +-1.3 NDC cull and the 1.3 tan(fov) clamp of the Jacobian stay centred — with the principal point 16-40 px off, Gaussians near one
    image border are culled or clamped that are visible, and the rays of the hit-depth test tilt."""
    cam = scenes.Camera(W, H, fx, fy, cx, cy, scenes.rot_yx(7.0, -3.0), np.array([0.05, -0.02, 0.1]))
    cam, sc = _cloud(8000, cam)
    fs, gs = U.parity_case(oracle, cam, sc, _dL(cam, 22), fp64=True)
    _report("principal_point_" + name, fs, gs, dict(cx=cx, cy=cy))
    centred = scenes.Camera(W, H, fx, fy, (W - 1) / 2, (H - 1) / 2, cam.Rw2c, cam.t)
    a, _ = U.run_hip(cam, sc)
    b, _ = U.run_hip(centred, sc)
    assert np.abs(a["color"] - b["color"]).max() > 1e-2
    if name != "tum_fr1":
        return
    # the same intrinsics on the surfel room (oblique planar surfels: the ray-plane branch of the hit depth with tilted rays)
    cam2, sc2 = _room(cx=cx, cy=cy, fx=fx)
    fs, gs = U.parity_case(oracle, cam2, sc2, _dL(cam2, 23), fp64=True)
    _report("principal_point_" + name + "_room", fs, gs, dict(cx=cx, cy=cy))


def test_prefiltered_flag_changes_nothing(torch_cuda):
    """`prefiltered=True` only arms a trap in the reference (auxiliary.h:155-161: a Gaussian behind the near plane then aborts the
    kernel); the library accepts and ignores it (include/dqo_raster.h, DqoRastParams.prefiltered): bit-identical outputs and gradients."""
    cam, sc = _cloud(5000)
    dL = _dL(cam, 24)
    a, ga = U.run_hip(cam, sc, dL=dL)
    b, gb = U.run_hip(cam, sc, dL=dL, prefiltered=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k


def _gate_from_render(cam, sc, **kw):
    res, _ = U.run_hip(cam, sc, **kw)
    hit = res["hit_depth"][0]
    go = np.asarray(sc["obj_id"], np.int32)
    return go, np.where(hit >= 0, go[np.clip(hit, 0, None)], -1).astype(np.int32)


def test_background_through_the_gated_op(torch_cuda, oracle):
    """Quirk B2 (forward.cu:812-852 adds T * bg with the RUNNING T, backward.cu:881,980 differentiates with end_T) through the GATED
    instantiation of the blend kernels — bg != 0 had only met the ungated kernel (test_gpu_rast.py)."""
    cam, sc = scenes.make_config(3, P=30000)
    bg = (0.1, 0.2, 0.3)
    go, po = _gate_from_render(cam, sc, bg=bg)
    fs, gs = U.parity_case(oracle, cam, sc, _dL(cam, 25), fp64=True, object_gate=(go, po), bg=bg)
    _report("bg_gated_op", fs, gs)
    hr = U.HipRun(cam, sc, grad=False, object_gate=(go, po), bg=bg)
    assert np.allclose(hr.res["color"][:, po < 0], np.asarray(bg, np.float32)[:, None])  # ownerless pixels of rendered tiles: T = 1 -> bg


def test_background_through_one_fused_iteration(torch_cuda, oracle):
    """... and through ONE iteration of the fused path (tile_objects binning, gated blend kernels, per-object loss tap, fused tail) at a
    size the serial oracle handles: loss to 1e-5, first Adam moment = the gradient the tail consumed to 1e-3 (util_rast.fused_iteration_case)."""
    fs, gs = U.fused_iteration_case(torch_cuda, oracle, 3, P=60000, bg=(0.1, 0.2, 0.3))
    _report("bg_fused_iteration", dict(mismatch_px=fs["flipped_px"]), gs, dict(loss_hip=fs["loss_hip"], loss_oracle=fs["loss_oracle"]))
