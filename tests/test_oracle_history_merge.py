"""CPU: the numpy restatement of Mapping.history_merge + slerp (oracle/map_oracle.py <- SLAM/multiprocess/mapper.py:607-650,
SLAM/utils.py:650-709) against fixtures produced with the REFERENCE's own slerp (tests/golden/make_history_merge_golden.py)."""
import os

import numpy as np

from oracle import map_oracle as mo

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "history_merge_golden.npz"))


def golden_case(ci):
    p = f"c{ci}_"
    hist = dict(confidence=G[p + "conf0"], xyz=G[p + "hist_xyz"], features_dc=G[p + "hist_dc"], features_rest=G[p + "hist_rest"],
                scaling=G[p + "hist_scaling"], rotation=G[p + "rot0"])
    cur = dict(confidence=G[p + "conf"], xyz=G[p + "cur_xyz"], features_dc=G[p + "cur_dc"], features_rest=G[p + "cur_rest"],
               scaling=G[p + "cur_scaling"], rotation_raw=G[p + "rot_raw"])
    want = dict(xyz=G[p + "out_xyz"], features_dc=G[p + "out_dc"], features_rest=G[p + "out_rest"], scaling=G[p + "out_scaling"],
                rotation=G[p + "out_rotation"])
    return hist, cur, float(G[p + "max_weight"]), want


def test_history_merge_restatement_matches_the_reference_fixtures():
    for ci in range(int(G["n_cases"])):
        hist, cur, mw, want = golden_case(ci)
        got = mo.history_merge(hist, cur, mw)
        for k in ("xyz", "features_dc", "features_rest", "scaling"):
            np.testing.assert_array_equal(got[k], want[k], err_msg=f"case {ci} {k}")  # IEEE lerps: bit for bit
        # slerp: arccos / sin of numpy vs torch differ in the last bits
        np.testing.assert_allclose(got["rotation"], want["rotation"], rtol=0, atol=2e-6, err_msg=f"case {ci} rotation")
        # both branches of the slerp are exercised
        n0, q = hist["rotation"], cur["rotation_raw"]
        dot = np.abs((n0 * (q / np.linalg.norm(q, axis=1, keepdims=True))).sum(1))
        if n0.shape[0] > 50:
            assert (dot > 0.9995).any() and (dot <= 0.9995).any()


def test_history_merge_quirk_first_rows_weight_serves_every_row():
    hist, cur, mw, want = golden_case(0)
    # changing the FIRST row's confidence changes every row's merged features; changing another row's does not
    cur2 = dict(cur, confidence=cur["confidence"].copy())
    cur2["confidence"][0] += 13.0
    got, got2 = mo.history_merge(hist, cur, mw), mo.history_merge(hist, cur2, mw)
    assert (got["features_dc"][5:] != got2["features_dc"][5:]).any() and (got["scaling"][5:] != got2["scaling"][5:]).any()
    np.testing.assert_array_equal(got["xyz"][1:], got2["xyz"][1:])
    cur3 = dict(cur, confidence=cur["confidence"].copy())
    cur3["confidence"][7] += 13.0
    got3 = mo.history_merge(hist, cur3, mw)
    np.testing.assert_array_equal(got["features_dc"], got3["features_dc"])
    assert mo.history_merge(hist, cur, 0.0)["xyz"] is not None and np.array_equal(mo.history_merge(hist, cur, 0.0)["xyz"], cur["xyz"])
