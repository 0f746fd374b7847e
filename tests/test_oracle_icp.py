"""CPU: the numpy oracle of the ICP normal equations (oracle/map_oracle.py, row f4) against goldens produced by the reference's own
ICP.compute_residuals_jacobian / compute_jtj / compute_jtr (tests/golden/make_icp_golden.py imports /root/reference/SLAM/icp.py)."""
import os

import numpy as np

from oracle import map_oracle as mo

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_golden.npz"))
CASES = sorted({k.split("_")[0] for k in G.files})
DIST_THR, NORMAL_THR = 0.2, float(np.cos(np.deg2rad(20)))


def case(c):
    return tuple(G[f"{c}_{k}_f16"].astype(np.float32) for k in ("v0", "v1", "n0", "n1")) + (G[f"{c}_pose"], G[f"{c}_K"])


def close_normal_equations(JtJ, JtR, c):
    scale = np.abs(G[f"{c}_JtJ"]).max()
    np.testing.assert_allclose(JtJ, G[f"{c}_JtJ"], rtol=0, atol=3e-4 * scale)
    np.testing.assert_allclose(np.asarray(JtR).reshape(-1), G[f"{c}_JtR"].reshape(-1), rtol=0, atol=3e-4 * np.abs(G[f"{c}_JtR"]).max() + 1e-6)


def test_normal_equations_match_reference_goldens():
    for c in CASES:
        v0, v1, n0, n1, pose, K = case(c)
        JtJ, JtR, valid = mo.icp_normal_equations(v0, v1, n0, n1, pose, K, DIST_THR, NORMAL_THR)
        # the valid mask is integer work: identical except pixels that sit on a threshold within fp32 rounding
        assert (valid != G[f"{c}_valid"]).mean() < 2e-3, c
        close_normal_equations(JtJ, JtR, c)
