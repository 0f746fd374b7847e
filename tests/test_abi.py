"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/dqo_raster.h declares
(no compute calls: there is no GPU here), size queries behave, and the Python surface mirrors the reference's."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dqo_raster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dqo_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    g.build_hip()
    import _dqo_native
    return _dqo_native


def test_exports_every_declared_symbol(native):
    lib = ctypes.CDLL(native.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dqo_raster.h but not exported"
    assert set(native.EXPORTS) == set(syms)


def test_size_queries_and_errors(native):
    lib = native.lib()
    assert lib.dqo_abi_version() == 5
    g1, g2 = lib.dqo_rast_geom_bytes(1000, 640, 480), lib.dqo_rast_geom_bytes(2000, 640, 480)
    assert 0 < g1 < g2 and g1 % 256 == 0
    assert lib.dqo_rast_image_bytes(1200, 680) >= 12 * 1200 * 680
    assert lib.dqo_rast_binning_bytes(1000) >= 20 * 1000
    assert lib.dqo_rast_backward_workspace_bytes(1000) >= 64 * 1000
    assert lib.dqo_knn3_workspace_bytes(5000) > 5000 * 24
    # argument validation happens before any launch: usable without a GPU
    p = native.DqoRastParams(P=-1, W=10, H=10)
    rc = lib.dqo_rast_forward_prepare(ctypes.byref(p), ctypes.byref(native.DqoRastInputs()), ctypes.byref(native.DqoRastOutputs()),
                                      ctypes.byref(native.DqoRastCtx()), None)
    assert rc == -1 and b"bad sizes" in lib.dqo_last_error()
    assert lib.dqo_knn3(-5, None, None, None, None, 0, None) == -1
    assert lib.dqo_knn3(10, 1, 1, 1, None, 0, None) == -2 and b"workspace" in lib.dqo_last_error()


def test_python_surface_matches_reference(native):
    import diff_gaussian_rasterization_depth as m
    assert m.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
        "campos", "opaque_threshold", "normal_threshold", "depth_threshold", "prefiltered", "debug", "cx", "cy", "color_sigma",
        "T_threshold")
    assert m.GaussianRasterizationSettings._field_defaults == {"color_sigma": 3.0, "T_threshold": 0.0001}
    sig = inspect.signature(m.GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp",
                                    "tile_mask", "normal_w"]
    assert list(inspect.signature(m.rasterize_gaussians).parameters) == [
        "means3D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp", "tile_mask", "raster_settings"]
    assert hasattr(m.GaussianRasterizer, "markVisible")
    from simple_knn._C import distCUDA2
    assert callable(distCUDA2)


def test_sync_modes_of_the_operator(native):
    """Host logic only: the four modes are accepted, anything else is refused, and the capacity hint of a shape only grows."""
    import diff_gaussian_rasterization_depth as m
    try:
        for mode in ("lazy", "deferred", "graph", "exact"):
            m.set_sync_mode(mode)
            assert m._sync_mode == mode
        with pytest.raises(ValueError):
            m.set_sync_mode("eager")
        m.set_capacity(1234, 64, 48, 1000, device_index=0)
        m.set_capacity(1234, 64, 48, 500, device_index=0)
        assert m._cap_hint[(0, 1234, 64, 48)] == 1000
    finally:
        m.set_sync_mode("exact")
        m._cap_hint.pop((0, 1234, 64, 48), None)


def test_no_cpu_fallback(native):
    import torch
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros(10, 3))
    import diff_gaussian_rasterization_depth as m
    rs = m.GaussianRasterizationSettings(48, 64, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3, torch.zeros(3), 0.6,
                                         0.5, 1.0, False, False, 31.5, 23.5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.GaussianRasterizer(rs)(means3D=torch.zeros(4, 3), opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3),
                                 scales=torch.ones(4, 3), rotations=torch.ones(4, 4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dqo-map_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_lib" not in txt and "libdqo_oracle" not in txt, f"{f} references the oracle"
    # the analysis scripts under tools/ run on the product only (diagnostics that need the oracle live under tests/)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", f)).read()
            assert "from oracle" not in txt and "import oracle" not in txt and "libdqo_oracle" not in txt, f"tools/{f} references the oracle"
