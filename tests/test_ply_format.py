"""CPU: the PLY wire format of the Gaussian maps (dqo_ply, row f4): header known-answer, channel-major SH layout, round trips,
ascii input, error paths.  (The reference writes through the third-party `plyfile` package, absent here: the format is pinned by
the public PLY specification + the attribute list / flattening order of gaussian_pointcloud.py:557-588, 641-684.)"""
import numpy as np
import pytest

import dqo_ply


def _map(P=7, M=16, seed=0):
    r = np.random.default_rng(seed)
    return dict(xyz=r.normal(size=(P, 3)).astype(np.float32), shs=r.normal(size=(P, M, 3)).astype(np.float32),
                opacity_raw=r.normal(size=(P, 1)).astype(np.float32), scaling_raw=r.normal(size=(P, 3)).astype(np.float32),
                rotation_raw=r.normal(size=(P, 4)).astype(np.float32), confidence=r.uniform(size=(P, 1)).astype(np.float32))


def test_header_and_record_layout(tmp_path):
    m = _map()
    p = tmp_path / "m.ply"
    dqo_ply.save_model_ply(p, **m)
    raw = p.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 7"]
    names = [l.split()[-1] for l in lines[3:] if l]
    assert names == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3", "confidence"]
    assert all(l.startswith("property float ") for l in lines[3:] if l)
    rec = np.frombuffer(body, "<f4").reshape(7, 63)
    np.testing.assert_array_equal(rec[:, :3], m["xyz"])
    assert (rec[:, 3:6] == 0).all()                                            # normals are written as zeros
    np.testing.assert_array_equal(rec[:, 6:9], m["shs"][:, 0, :])               # f_dc_c = channel c of coefficient 0
    np.testing.assert_array_equal(rec[:, 9 + 15 * 1 + 4], m["shs"][:, 1 + 4, 1])  # f_rest is CHANNEL-major: index = c * 15 + (k - 1)
    np.testing.assert_array_equal(rec[:, 54], m["opacity_raw"][:, 0])
    np.testing.assert_array_equal(rec[:, 62], m["confidence"][:, 0])


@pytest.mark.parametrize("deg,conf", [(3, True), (3, False), (1, True), (0, True)])
def test_round_trip(tmp_path, deg, conf):
    m = _map(P=100, M=(deg + 1) ** 2, seed=deg)
    p = tmp_path / "m.ply"
    dqo_ply.save_model_ply(p, include_confidence=conf, **m)
    back = dqo_ply.load_model_ply(p, max_sh_degree=deg)
    for k in ("xyz", "shs", "opacity_raw", "scaling_raw", "rotation_raw"):
        np.testing.assert_array_equal(back[k], m[k])
    np.testing.assert_array_equal(back["confidence"], m["confidence"] if conf else np.zeros((100, 1), np.float32))


def test_ascii_input_and_errors(tmp_path):
    m = _map(P=3, M=1)
    names = dqo_ply.attribute_names(0, True)
    p = tmp_path / "a.ply"
    rows = np.concatenate([m["xyz"], np.zeros((3, 3)), m["shs"][:, 0, :], m["opacity_raw"], m["scaling_raw"], m["rotation_raw"],
                           m["confidence"]], 1)
    p.write_text("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\n" + "".join(f"property float {n}\n" for n in names) +
                 "end_header\n" + "\n".join(" ".join(repr(float(x)) for x in r) for r in rows) + "\n")
    back = dqo_ply.load_model_ply(p, max_sh_degree=0)
    np.testing.assert_allclose(back["xyz"], m["xyz"], rtol=1e-7)
    np.testing.assert_allclose(back["rotation_raw"], m["rotation_raw"], rtol=1e-7)
    with pytest.raises(AssertionError):
        dqo_ply.load_model_ply(p, max_sh_degree=3)  # wrong number of f_rest columns for the degree (gaussian_pointcloud.py:162)
    q = tmp_path / "bad.ply"
    q.write_bytes(b"plx\n")
    with pytest.raises(ValueError):
        dqo_ply.load_model_ply(q)
    dqo_ply.save_model_ply(tmp_path / "empty.ply", **_map(P=0))
    assert not (tmp_path / "empty.ply").exists()  # an empty map writes nothing (gaussian_pointcloud.py:642-643)
