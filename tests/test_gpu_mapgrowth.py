"""GPU: row f3, map-growth geometry — the 3-NN search with a query set different from the reference set (dqo_knn3_query through
dqo_mapgrowth) against the brute-force fp32 oracle and scipy's cKDTree, and the two reference decisions built on it."""
import numpy as np
import pytest

from oracle import map_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import _dqo_native
    _dqo_native.lib()
    import dqo_mapgrowth
    return torch, dqo_mapgrowth


def _check_knn(torch, M, q, r):
    d, i = M.knn_points_k3(torch.tensor(q, device="cuda"), torch.tensor(r, device="cuda"))
    d, i = d.cpu().numpy(), i.cpu().numpy()
    od, oi = mo.knn3_query(q, r)
    np.testing.assert_array_equal(d, od)  # the three smallest squared distances, bit-exact in fp32
    k = min(3, len(r))
    # indices: equal to the oracle's unless two references are equally far (then any of them is a correct answer)
    dd = q[:, None, :] - r[i[:, :k].clip(0)]
    dd = dd * dd
    np.testing.assert_array_equal(((dd[..., 0] + dd[..., 1]) + dd[..., 2]).astype(np.float32), od[:, :k])
    assert (i[:, k:] == -1).all()
    for row in i[:, :k]:
        assert len(set(row.tolist())) == k
    return d, i


@pytest.mark.parametrize("Q,R,seed", [(5000, 30000, 0), (1, 1, 1), (70, 2, 2), (3000, 5, 3), (257, 1025, 4), (20000, 100000, 5)])
def test_knn3_query_vs_oracle(mg, Q, R, seed):
    torch, M = mg
    rng = np.random.default_rng(seed)
    r = rng.uniform(-2, 2, (R, 3)).astype(np.float32)
    q = rng.uniform(-2.5, 2.5, (Q, 3)).astype(np.float32)
    if Q > 100 and R > 100:
        q[:50] = r[:50]              # queries that coincide with references (distance 0)
        r[100:110] = r[90:100]       # duplicated references (ties)
    d, i = _check_knn(torch, M, q, r)
    if R >= 3:
        from scipy.spatial import cKDTree
        dk, _ = cKDTree(r.astype(np.float64)).query(q.astype(np.float64), k=3)
        np.testing.assert_allclose(np.sqrt(d), dk, rtol=2e-5, atol=1e-6)


def test_clustered_points(mg):
    """Surfel-like data: points on a few planes, queries near them — the pruning path that real maps exercise."""
    torch, M = mg
    rng = np.random.default_rng(9)
    r = np.concatenate([np.c_[rng.uniform(-3, 3, (20000, 2)), np.full(20000, z)] for z in (-1.0, 0.0, 1.5)]).astype(np.float32)
    q = (r[rng.choice(len(r), 8000, replace=False)] + rng.normal(0, 0.01, (8000, 3))).astype(np.float32)
    _check_knn(torch, M, q, r)


def test_temp_points_filter_and_update_geometry(mg):
    torch, M = mg
    rng = np.random.default_rng(10)
    exist = rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
    radius = rng.uniform(0.005, 0.03, (20000, 1)).astype(np.float32)
    temp = rng.uniform(-0.5, 0.5, (6000, 3)).astype(np.float32)
    mask = M.temp_points_filter_mask(torch.tensor(temp, device="cuda"), torch.tensor(exist, device="cuda"),
                                     torch.tensor(radius, device="cuda")).cpu().numpy()
    # restatement of mapper.py:1351-1380 on the oracle's neighbours
    lo, hi = temp.min(0) - 0.05, temp.max(0) + 0.05
    inb = ((exist > lo).all(1)) & ((exist < hi).all(1))
    od, oi = mo.knn3_query(temp, exist[inb])
    want = (np.sqrt(od) < radius[inb].reshape(-1)[oi] * np.float32(0.6)).any(1)
    assert (mask != want).mean() < 1e-4  # sqrt / compare at the threshold may differ by an ulp
    assert 0.0 < want.mean() < 1.0
    # update_geometry: finite, clipped scales; invalid where a neighbour's 3-sigma sphere contains the point
    sc, inv = M.update_geometry_scales(torch.tensor(temp, device="cuda"), torch.full((6000, 1), 0.01, device="cuda"),
                                       torch.tensor(exist, device="cuda"), torch.tensor(radius, device="cuda"), 0.002, 0.05)
    sc = sc.cpu().numpy()
    assert np.isfinite(sc).all() and sc.min() >= 0.002 - 1e-9 and sc.max() <= 0.05 + 1e-9
    assert 0 < inv.float().mean().item() < 1
