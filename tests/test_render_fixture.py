"""Row a17: the harness' `render()` (restated SLAM/render.py:134-272) against the committed 64x48 `(args -> 9-tuple)` fixtures
(tests/golden/render_golden.npz, made by tests/golden/make_render_golden.py with the fp32 oracle).
CPU: the oracle still reproduces the fixture (regression pin).  GPU: the HIP path through render() matches it."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_render_golden import NAMES, scene_64x48  # noqa: E402
import util_rast as U  # noqa: E402

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_golden.npz"))
BG = (0.05, 0.1, 0.15)


def test_fixture_inputs_are_the_seeded_scene():
    cam, sc = scene_64x48()
    for k in ("xyz", "scales", "rotations", "opacity", "shs", "normals"):
        assert np.array_equal(G[k], sc[k]), k


def test_oracle_reproduces_fixture(oracle):
    cam, sc = scene_64x48()
    dL = (G["dL_dcolor"], G["dL_ddepth"])
    for tag, tm in (("full", None), ("masked", G["tile_mask"])):
        _, r, g = U.run_oracle(oracle, cam, sc, tile_mask=tm, dL=dL, bg=BG)
        for k in NAMES:
            assert np.array_equal(r[k], G[f"{tag}_{k}"]), (tag, k)
        for k, v in g.items():
            np.testing.assert_allclose(v, G[f"{tag}_grad_{k}"], rtol=1e-6, atol=1e-9)
    # masked tiles keep the reference's initial fills (rasterize_points.cu:79-89): ids 0 (quirk B7), T 1, colour 0
    pm = np.repeat(np.repeat(G["tile_mask"], 16, 0), 16, 1)[:cam.H, :cam.W].astype(bool)
    assert (G["masked_hit_depth"][0][~pm] == 0).all() and (G["masked_T_map"][0][~pm] == 1).all() and (G["masked_color"][:, ~pm] == 0).all()


@pytest.mark.gpu
def test_render_dict_matches_fixture_on_gpu():
    import torch
    assert torch.cuda.is_available()
    import _dqo_native
    _dqo_native.lib()
    from dqo_harness import mapping
    cam, sc = scene_64x48()
    dev = torch.device("cuda")
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    settings = mapping.make_settings(cam, dev, bg=BG)
    for tag, tm in (("full", None), ("masked", G["tile_mask"])):
        data = dict(xyz=t(sc["xyz"]).requires_grad_(True), opacity=t(sc["opacity"]).requires_grad_(True), scales=t(sc["scales"]).requires_grad_(True),
                    rotations=t(sc["rotations"]).requires_grad_(True), shs=t(sc["shs"]).requires_grad_(True), normal=t(sc["normals"]),
                    semantics_color=None, instance=None)
        out = mapping.render(settings, data, tile_mask=None if tm is None else t(tm))
        # the reference's keys (SLAM/render.py:213-222, 269-270)
        assert {"render", "depth", "normal", "color_index_map", "depth_index_map", "color_hit_weight", "depth_hit_weight", "T_map",
                "semantic_seg", "instance", "n_touched"} <= set(out)
        assert out["semantic_seg"] is None and out["instance"] is None
        key = dict(color="render", depth="depth", hit_color="color_index_map", hit_depth="depth_index_map", hit_color_weight="color_hit_weight",
                   hit_depth_weight="depth_hit_weight", T_map="T_map", n_touched="n_touched", radii="radii")
        h = {k: out[v].detach().cpu().numpy() for k, v in key.items()}
        o = {k: G[f"{tag}_{k}"] for k in NAMES}
        bad = U.flipped_pixels(h, o)
        U.compare_forward(h, o)
        # normal gather (render.py:208-212), also over the aliased id-0 pixels of masked tiles
        rn = out["normal"].cpu().numpy()
        assert np.array_equal(rn[:, ~bad], G[f"{tag}_normal"][:, ~bad])
        keep = torch.tensor((~bad).astype(np.float32), device=dev)
        loss = (out["render"] * t(G["dL_dcolor"]) * keep).sum() + (out["depth"] * t(G["dL_ddepth"]) * keep).sum()
        loss.backward()
        if bad.sum() == 0:
            hg = dict(means3D=data["xyz"].grad, opacity=data["opacity"].grad, scales=data["scales"].grad, rotations=data["rotations"].grad,
                      sh=data["shs"].grad)
            U.compare_grads({k: v.cpu().numpy() for k, v in hg.items()}, {k: G[f"{tag}_grad_{k}"] for k in hg})
    # second pass with precomputed colours (semantics / instance renders, render.py:224-266)
    data["instance"] = t(np.random.default_rng(1).uniform(0, 1, (sc["xyz"].shape[0], 3)).astype(np.float32))
    out = mapping.render(settings, data)
    assert out["instance"] is not None and out["instance"].shape == out["render"].shape and out["semantic_seg"] is None
