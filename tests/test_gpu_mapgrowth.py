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


@pytest.mark.parametrize("Q,R,n_groups,reach,seed", [(4000, 60000, 8, 0.25, 0), (3000, 20000, 64, None, 1), (500, 300, 3, 0.5, 2), (2000, 50000, 1, 0.1, 3)])
def test_grouped_query_vs_brute_force(mg, Q, R, n_groups, reach, seed):
    """dqo_knn3_query_grouped (the per-object growth decisions' search): a reference counts for a query only if it carries the same
    group id, lies strictly inside the group's box and (reach) is closer than the bound — against a brute-force float32 statement of
    exactly that, distances bit-exact.  Points with an id outside [0, 64) belong to no group: never found, never finding."""
    torch, M = mg
    rng = np.random.default_rng(seed)
    r = rng.uniform(-2, 2, (R, 3)).astype(np.float32)
    q = rng.uniform(-2, 2, (Q, 3)).astype(np.float32)
    gr = rng.integers(0, n_groups, R).astype(np.int32)
    gq = rng.integers(0, n_groups, Q).astype(np.int32)
    gr[rng.choice(R, R // 50, replace=False)] = -1   # dead rows
    gq[rng.choice(Q, max(Q // 100, 1), replace=False)] = 77  # queries without a group
    box = np.full((64, 6), 0, np.float32)
    box[:, :3], box[:, 3:] = -np.inf, np.inf
    for g in range(0, n_groups, 2):  # every other group: a real box
        c = rng.uniform(-1, 1, 3)
        box[g, :3], box[g, 3:] = c - 1.2, c + 1.2
    t = lambda a: torch.tensor(a, device="cuda")
    d, i = M.knn_points_k3(t(q), t(r), max_dist=reach, groups=(t(gq), t(gr)), group_box=t(box))
    d, i = d.cpu().numpy(), i.cpu().numpy()
    FLT_MAX = np.float32(3.4028234663852886e38)
    bound2 = FLT_MAX if reach is None else np.float32(reach) * np.float32(reach)
    for k in rng.choice(Q, min(Q, 400), replace=False):
        g = gq[k]
        if not (0 <= g < 64):
            assert (i[k] == -1).all() and (d[k] == FLT_MAX).all()
            continue
        ok = (gr == g) & (r > box[g, :3]).all(1) & (r < box[g, 3:]).all(1)
        dd = (r - q[k]).astype(np.float32)
        dd = dd * dd
        d2 = ((dd[:, 0] + dd[:, 1]) + dd[:, 2]).astype(np.float32)
        d2 = np.where(ok & (d2 < bound2), d2, FLT_MAX)
        best = np.sort(d2)[:3]
        best = np.concatenate([best, np.full(3 - len(best), FLT_MAX, np.float32)])
        np.testing.assert_array_equal(d[k], best, err_msg=f"query {k}")
        for j in range(3):
            if best[j] < FLT_MAX:
                assert i[k, j] >= 0 and gr[i[k, j]] == g and d2[i[k, j]] == best[j]
            else:
                assert i[k, j] == -1


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


def test_fused_mapper_grow_and_recapture():
    """Row ★ (cfg 5's growth step): FusedMapper.grow = temp_points_filter + update_geometry + cat (+ delete) on the mapper's own map,
    then a NEW mapping call (fresh Adam, fresh init_stat) on re-allocated buffers and a re-captured graph — no process restart.
    The grown mapper must behave exactly like a mapper built directly on the grown map."""
    import torch
    from dqo_harness import mapping, scenes
    from dqo_harness.fused_mapping import FusedMapper
    dev = torch.device("cuda")
    cam, scene = scenes.make_config(3, P=12000)
    settings = mapping.make_settings(cam, dev)
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(scene, dev).activated())
    gt_color, gt_depth = tgt["render"].clone(), tgt["depth"].clone()
    mask = (tgt["depth_index_map"][0] >= 0)
    fm = FusedMapper(scene, settings, dev)
    fm.capture(gt_color, gt_depth, mask)
    for _ in range(3):
        fm.replay()
    torch.cuda.synchronize()
    old = {k: v.clone() for k, v in fm._params().items()}
    P0 = fm.P
    new = scenes.surfel_room(77, 3000, n_objects=8)  # other random patches: partly inside the existing map, partly new surface
    delete = torch.zeros(P0, dtype=torch.bool, device=dev)
    delete[::17] = True
    st = fm.grow(new, delete_mask=delete)
    assert st["candidates"] == 3000 and st["candidates"] == st["inside_existing"] + st["invalid_scale"] + st["added"]
    assert st["inside_existing"] > 0 and st["added"] > 0 and st["deleted"] == int(delete.sum().item())
    assert fm.P == P0 - st["deleted"] + st["added"]
    keep = (~delete).nonzero().reshape(-1)
    for k, v in fm._params().items():  # the surviving old Gaussians are untouched, in order
        assert torch.equal(v[: keep.numel()], old[k][keep]), k
    # the new scales follow gaussian_pointcloud.py:558-568: isotropic in-plane, 0.1 along the third axis, inside [min, max] radius
    ns = torch.exp(fm.scaling_raw[keep.numel():])
    assert (ns[:, 0] == ns[:, 1]).all() and torch.allclose(ns[:, 2], 0.1 * ns[:, 0]) and ns[:, 0].min() >= 0.001 - 1e-9 and ns[:, 0].max() <= 0.05 + 1e-9
    with pytest.raises(Exception):
        fm.replay()  # the captured graph died with the old buffers
    fm.begin_mapping_call(reset_optimizer=True)
    twin = FusedMapper(scenes.surfel_room(5, fm.P, n_objects=8), settings, dev)
    twin.xyz, twin.shs, twin.opacity_raw = fm.xyz.clone(), fm.shs.clone(), fm.opacity_raw.clone()
    twin.scaling_raw, twin.rotation_raw = fm.scaling_raw.clone(), fm.rotation_raw.clone()
    twin.begin_mapping_call(reset_optimizer=True)
    twin._act_valid = False
    fm.capture(gt_color, gt_depth, mask)
    twin.capture(gt_color, gt_depth, mask)
    for _ in range(3):
        fm.replay()
        twin.replay()
    torch.cuda.synchronize()
    assert not fm.graph_overflowed() and fm.step_count == twin.step_count == 4
    for k, v in fm._params().items():
        assert torch.equal(v, twin._params()[k]), k
    assert torch.equal(fm.loss, twin.loss) and torch.isfinite(fm.loss).all()


def test_scale_init_without_the_all_pairs_knn_equals_the_reference_call(mg):
    """update_geometry_scales: the default path (3 nearest among the other new points + 3 nearest among the existing ones, the 3
    nearest of those 6) gives what the reference's literal call gives (distCUDA2 over new + existing, first P rows)."""
    torch, m = mg
    rng = np.random.default_rng(8)
    new = torch.tensor(rng.uniform(-1, 1, (3000, 3)).astype(np.float32), device="cuda")
    old = torch.tensor(rng.uniform(-1.2, 1.2, (50000, 3)).astype(np.float32), device="cuda")
    r_new = torch.tensor(rng.uniform(0.001, 0.02, 3000).astype(np.float32), device="cuda")
    r_old = torch.tensor(rng.uniform(0.001, 0.02, 50000).astype(np.float32), device="cuda")
    s1, inv1 = m.update_geometry_scales(new, r_new, old, r_old, 0.001, 0.05, literal=True)
    s2, inv2 = m.update_geometry_scales(new, r_new, old, r_old, 0.001, 0.05)
    assert torch.equal(inv1, inv2)
    np.testing.assert_allclose(s2.cpu().numpy(), s1.cpu().numpy(), rtol=2e-6, atol=1e-9)
    # no existing points / a single new point
    s3, _ = m.update_geometry_scales(new, r_new, old[:0], r_old[:0], 0.001, 0.05)
    s4, _ = m.update_geometry_scales(new, r_new, old[:0], r_old[:0], 0.001, 0.05, literal=True)
    np.testing.assert_allclose(s3.cpu().numpy(), s4.cpu().numpy(), rtol=2e-6, atol=1e-9)


def test_grow_into_a_new_mapping_call_and_recapture_with_reused_counts():
    """grow(new_mapping_call=True) = grow() + begin_mapping_call(reset_optimizer=True) without carrying the old moments and snapshots
    along, and capture(reuse_probe=True) = capture() without the probing forward: same buffers' worth of results, bit for bit."""
    import torch
    from dqo_harness import mapping, scenes
    from dqo_harness.fused_mapping import FusedMapper
    dev = torch.device("cuda")
    cam, scene = scenes.make_config(3, P=12000)
    settings = mapping.make_settings(cam, dev)
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(scene, dev).activated())
    gt_color, gt_depth = tgt["render"].clone(), tgt["depth"].clone()
    mask = (tgt["depth_index_map"][0] >= 0)
    new = scenes.surfel_room(78, 3000, n_objects=8)
    ms = []
    for fast in (True, False):
        fm = FusedMapper(scene, settings, dev)
        fm.capture(gt_color, gt_depth, mask)
        for _ in range(3):
            fm.replay()
        torch.cuda.synchronize()
        delete = torch.zeros(fm.P, dtype=torch.bool, device=dev)
        delete[5::23] = True
        if fast:
            st = fm.grow(new, delete_mask=delete, new_mapping_call=True)
            fm.capture(gt_color, gt_depth, mask, reuse_probe=True)
            assert fm._last_probe[2] == fm.P
        else:
            st = fm.grow(new, delete_mask=delete)
            fm.begin_mapping_call(reset_optimizer=True)
            fm.capture(gt_color, gt_depth, mask)
        assert st["added"] > 0 and st["deleted"] == int(delete.sum().item())
        for _ in range(3):
            fm.replay()
        torch.cuda.synchronize()
        assert not fm.graph_overflowed() and fm.step_count == 4
        ms.append(fm)
    a, b = ms
    assert a.P == b.P and a.attach_count == b.attach_count
    for k, v in a._params().items():
        assert torch.equal(v, b._params()[k]), k
        assert torch.equal(a.state[k][0], b.state[k][0]) and torch.equal(a.state[k][1], b.state[k][1]), k
    assert torch.equal(a.loss, b.loss) and torch.equal(a.init_xyz, b.init_xyz) and torch.equal(a.attach_mask, b.attach_mask)
    # reused counts that turn out too small: the capture notices on its own eager iteration and measures after all
    c = FusedMapper(scene, settings, dev)
    c.capture(gt_color, gt_depth, mask)
    c._last_probe = (10, 4, c.P)  # absurdly small
    c.capture(gt_color, gt_depth, mask, reuse_probe=True)
    assert not c.graph_overflowed() and c._last_probe[0] > 1000


def _growth_problem(P=12000):
    import torch
    from dqo_harness import mapping, scenes
    dev = torch.device("cuda")
    cam, scene = scenes.make_config(3, P=P)
    settings = mapping.make_settings(cam, dev)
    with torch.no_grad():
        tgt = mapping.render(settings, mapping.GaussianParams(scene, dev).activated())
    return dev, cam, scene, settings, tgt["render"].clone(), tgt["depth"].clone(), (tgt["depth_index_map"][0] >= 0)


def _rows_sorted(fm):
    """The map's Gaussians in an order that does not depend on their rows (lexicographic by their centres at the start of the mapping
    call, which training does not move)."""
    import torch
    rows = torch.arange(fm.P, device=fm.xyz.device) if fm.alive is None else fm.alive.nonzero().reshape(-1)
    x = fm.init_xyz[rows].cpu().numpy()
    o = np.lexsort((x[:, 2], x[:, 1], x[:, 0]))
    return rows[torch.tensor(o, device=rows.device)]


def test_growth_in_place_keeps_the_captured_graph_and_equals_the_reallocating_step():
    """reserve() + grow(new_mapping_call=True): deleted Gaussians become spare rows, new ones take spare rows, the attach set / init_stat /
    Adam state are rewritten in place — the graph captured BEFORE the growth step is replayed after it, and the map trains like the map
    of the re-allocating step (same Gaussians in other rows).  Spare rows behave as if they did not exist."""
    import torch
    from dqo_harness import scenes
    from dqo_harness.fused_mapping import FusedMapper
    dev, cam, scene, settings, gt_color, gt_depth, mask = _growth_problem()
    new = scenes.surfel_room(79, 3000, n_objects=8)
    a = FusedMapper(scene, settings, dev)
    b = FusedMapper(scene, settings, dev).reserve(4000)
    assert b.P == a.P + 4000 and b.n_alive == a.P
    for fm in (a, b):
        fm.capture(gt_color, gt_depth, mask)
        for _ in range(3):
            fm.replay()
    torch.cuda.synchronize()
    # spare rows: invisible to the render, the loss and the optimiser
    assert torch.equal(a.loss, b.loss)
    for k, v in a._params().items():
        assert torch.equal(v, b._params()[k][: a.P]), k
    assert int(b._g.out[8][a.P:].abs().max().item()) == 0 and float(b.state["xyz"][0][a.P:].abs().max().item()) == 0.0
    delete = torch.zeros(a.P, dtype=torch.bool, device=dev)
    delete[3::19] = True
    sa = a.grow(new, delete_mask=delete, new_mapping_call=True)
    graph_before = b._g
    sb = b.grow(new, delete_mask=torch.cat([delete, torch.zeros(4000, dtype=torch.bool, device=dev)]), new_mapping_call=True)
    assert sb["in_place"] and b._g is graph_before and not b._g.stale
    assert (sa["added"], sa["deleted"], sa["inside_existing"], sa["invalid_scale"]) == (sb["added"], sb["deleted"], sb["inside_existing"], sb["invalid_scale"])
    assert b.n_alive == a.P and b.attach_count == a.attach_count and sa["added"] > 0
    a.capture(gt_color, gt_depth, mask)  # (one iteration of the new mapping call runs inside the capture)
    for _ in range(3):
        a.replay()
    for _ in range(4):
        b.replay()  # the graph captured before the growth step
    torch.cuda.synchronize()
    assert not a.graph_overflowed() and not b.graph_overflowed() and a.step_count == b.step_count
    ra, rb = _rows_sorted(a), _rows_sorted(b)
    # (Adam turns a gradient that changes sign near zero into a full step, and the two maps sum in different orders: compare in the
    # norm and by the share of elements that went another way, not element by element)
    for k, v in a._params().items():
        d = (b._params()[k][rb] - v[ra]).abs()
        assert float(d.mean()) < 2e-5 and float((d > 1e-4).float().mean()) < 0.02, (k, float(d.mean()), float((d > 1e-4).float().mean()))
    np.testing.assert_allclose(b.loss[:3].cpu().numpy(), a.loss[:3].cpu().numpy(), rtol=1e-4)
    np.testing.assert_allclose(float(b.attach_loss()), float(a.attach_loss()), rtol=1e-4)


def test_growth_runs_out_of_spare_rows_and_reserves_again():
    import torch
    from dqo_harness import scenes
    from dqo_harness.fused_mapping import FusedMapper
    dev, cam, scene, settings, gt_color, gt_depth, mask = _growth_problem(8000)
    fm = FusedMapper(scene, settings, dev).reserve(16)
    fm.capture(gt_color, gt_depth, mask)
    fm.replay()
    st = fm.grow(scenes.surfel_room(80, 3000, n_objects=8), new_mapping_call=True)
    assert st["added"] > 16 and st["in_place"] is False and fm._g is None
    assert fm.n_alive == 8000 + st["added"] and fm.P == fm.n_alive + 16  # compacted, the same number of spare rows again
    fm.capture(gt_color, gt_depth, mask)
    for _ in range(2):
        fm.replay()
    torch.cuda.synchronize()
    assert not fm.graph_overflowed() and torch.isfinite(fm.loss).all()
    st2 = fm.grow(dict(xyz=np.zeros((0, 3), np.float32), scales=np.zeros((0, 3), np.float32), rotations=np.zeros((0, 4), np.float32),
                       opacity=np.zeros((0, 1), np.float32), shs=np.zeros((0, 16, 3), np.float32)), new_mapping_call=True)
    assert st2["added"] == 0 and st2["in_place"] and fm._g is not None


def test_growth_attaches_new_points_that_fall_onto_the_stable_cloud(mg):
    """grow(stable_mask=...) = the reference's two clouds (mapper.py:1351-1466): the filter looks at the unstable Gaussians, and
    temp_points_attach gives the new points that lie on the stable cloud's surfaces opacity 0.1 — members of the next attach set.  The
    stable-only render (unstable opacities at zero) is checked against a render of the stable Gaussians as a map of their own."""
    torch, M = mg
    from dqo_harness import mapping, scenes
    from dqo_harness.fused_mapping import FusedMapper
    dev, cam, scene, settings, gt_color, gt_depth, mask = _growth_problem()
    fm = FusedMapper(scene, settings, dev).reserve(4000)
    stable = torch.zeros(fm.P, dtype=torch.bool, device=dev)
    stable[: 12000 : 2] = True
    new = scenes.surfel_room(81, 3000, n_objects=8)
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    nx, nop = t(new["xyz"]), t(new["opacity"]).reshape(-1, 1)
    got = fm._temp_points_attach(nx, nop, stable, 0.1)
    # reference route: the stable Gaussians as a map of their own (same activated values)
    rows = stable.nonzero().reshape(-1)
    sub = dict(xyz=fm.xyz[rows], opacity=fm.opacity[rows], scales=fm.scales[rows], rotations=fm.rotations[rows], shs=fm.shs[rows])
    with torch.no_grad():
        cim = mapping.render(settings, sub)["color_index_map"]
    H, W = cam.H, cam.W
    K = torch.tensor([[W / (2.0 * cam.tanfovx), 0.0, cam.cx], [0.0, H / (2.0 * cam.tanfovy), cam.cy], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    want = M.temp_points_attach_indices(nx, nop, settings.viewmatrix.T.contiguous(), K, W, H, cim, fm.xyz[rows], fm.normals(rows), fm.add_depth_thres)
    assert torch.equal(torch.sort(got).values, torch.sort(want).values) and 0 < got.numel() < nx.shape[0]
    before = fm.attach_count
    st = fm.grow(new, new_mapping_call=True, stable_mask=stable)
    assert st["in_place"] and st["attached"] > 0 and st["added"] > 0
    new_rows = st["rows"]
    low = (torch.sigmoid(fm.opacity_raw[new_rows]) < 0.2).reshape(-1)
    assert int(low.sum().item()) > 0 and bool(fm.attach_mask[new_rows][low].all())  # they are in the new call's attach set
    assert fm.attach_count == before + int((torch.sigmoid(fm.opacity_raw[new_rows]) < 0.9).sum().item())


def test_a_new_gaussian_in_a_row_freed_by_a_stable_one_is_unstable(mg):
    """In-place growth hands out spare rows lowest index first — including the row a deleted STABLE Gaussian just freed.  What growth
    adds belongs to the unstable cloud (mapper.py:1438-1466): grow() clears the caller's stable mask on the rows it fills, so the next
    step filters against them and keeps them out of the stable-only render."""
    torch, M = mg
    from dqo_harness import scenes
    from dqo_harness.fused_mapping import FusedMapper
    dev, cam, scene, settings, gt_color, gt_depth, mask = _growth_problem()
    P0 = 12000
    fm = FusedMapper(scene, settings, dev).reserve(4000)
    stable = torch.arange(fm.P, device=dev) < P0     # bench.py's split: the map the run starts from is the stable cloud
    delete = torch.zeros(fm.P, dtype=torch.bool, device=dev)
    delete[5:200:7] = True                           # stable rows, all below every spare row
    st = fm.grow(scenes.surfel_room(83, 3000, n_objects=8), delete_mask=delete, new_mapping_call=True, stable_mask=stable)
    rows = st["rows"]
    reused = rows[rows < P0]
    assert st["in_place"] and reused.numel() == int(delete.sum().item()) and st["added"] > reused.numel()
    assert not bool(stable[rows].any())              # every new Gaussian is unstable, the reused rows included
    assert bool(stable[:P0][~delete[:P0]].all())     # the surviving stable Gaussians stay stable
    # the next step's filter (mapper.py:1356-1357: against the unstable cloud) sees them: the same points again all fall inside
    st2 = fm.grow(dict(xyz=fm.xyz[rows].clone(), scales=torch.exp(fm.scaling_raw[rows]), rotations=fm.rotation_raw[rows].clone(),
                       opacity=torch.sigmoid(fm.opacity_raw[rows]), shs=fm.shs[rows].clone()), new_mapping_call=True, stable_mask=stable)
    assert st2["inside_existing"] == rows.numel() and st2["added"] == 0


def test_a_sharded_map_grows_exactly_like_the_unsharded_map(mg):
    """The per-object job's growth step (FusedMapper.grow with an object gate): every decision — inside an existing Gaussian? on a
    stable surfel's plane? which neighbours set the scale? — judges a candidate against the Gaussians of its OWN object, so two shards
    that hold disjoint object sets take, for their objects, exactly the decisions the map that holds every object takes: the same
    Gaussians are added (bit for bit: position, scale, opacity, rotation) and the same ones deleted, whatever the shard layout."""
    torch, M = mg
    from dqo_harness import mapping, scenes
    from dqo_harness.fused_mapping import FusedMapper
    dev, cam, scene, settings, gt_color, gt_depth, mask = _growth_problem()
    P = 12000
    go = np.asarray(scene["obj_id"], np.int32)
    with torch.no_grad():
        hit = mapping.render(settings, mapping.GaussianParams(scene, dev).activated())["depth_index_map"][0].cpu().numpy()
    po = np.where(hit >= 0, go[np.clip(hit, 0, None)], -1).astype(np.int32)
    new = scenes.surfel_room(84, 6000, n_objects=8)
    objs_a = [0, 2, 5, 7]
    in_a = np.isin(go, objs_a)
    sub = lambda m: {k: (v[m] if hasattr(v, "shape") and v.shape[:1] == (P,) else v) for k, v in scene.items()}
    nsub = lambda m: {k: (np.asarray(v)[m] if hasattr(v, "shape") and v.shape[:1] == (6000,) else v) for k, v in new.items()}
    new_in_a = np.isin(np.asarray(new["obj_id"]), objs_a)
    rng = np.random.default_rng(5)
    delete = rng.uniform(size=P) < 0.01
    half_stable = (np.arange(P) % 3) != 0  # a third of the map is the unstable cloud the filter looks at

    def run(sc, gobj, nw, keep):
        fm = FusedMapper(sc, settings, dev).set_object_gate(gobj, po).reserve(3000)
        n0 = len(gobj)
        stable = torch.zeros(fm.P, dtype=torch.bool, device=dev)
        stable[:n0] = torch.tensor(half_stable[keep], device=dev)
        dm = torch.zeros(fm.P, dtype=torch.bool, device=dev)
        dm[:n0] = torch.tensor(delete[keep], device=dev)
        st = fm.grow(nw, delete_mask=dm, new_mapping_call=True, stable_mask=stable)
        assert st["in_place"]
        alive = fm.alive.bool()
        f = lambda a: a[alive].cpu().numpy()
        rows = np.concatenate([f(fm.xyz), f(fm.scaling_raw), f(fm.opacity_raw), f(fm.rotation_raw), f(fm.shs).reshape(int(alive.sum()), -1)], 1)
        return st, f(fm.gaussian_object), rows

    st_w, obj_w, rows_w = run(scene, go, new, np.ones(P, bool))
    st_a, obj_a, rows_a = run(sub(in_a), go[in_a], nsub(new_in_a), in_a)
    st_b, obj_b, rows_b = run(sub(~in_a), go[~in_a], nsub(~new_in_a), ~in_a)
    assert st_w["added"] > 100 and st_w["inside_existing"] > 0 and st_w.get("attached", 0) > 0
    for k in ("added", "deleted", "inside_existing", "invalid_scale", "attached"):
        assert st_a[k] + st_b[k] == st_w[k], (k, st_a[k], st_b[k], st_w[k])
    canon = lambda r: r[np.lexsort(r.T[::-1])]
    for k in np.unique(obj_w):
        want = canon(rows_w[obj_w == k])
        got = canon(rows_a[obj_a == k]) if k in objs_a else canon(rows_b[obj_b == k])
        assert want.shape == got.shape and np.array_equal(want, got), (int(k), want.shape, got.shape)


def test_gated_attach_on_the_candidates_pixels_equals_the_full_frame_chain(mg):
    """FusedMapper._temp_points_attach(temp_obj=...) renders the stable cloud for the candidates' own pixels only (dqo_attach_pixels: every
    other pixel ownerless in the object gate, per-tile owner sets for the binning) and decides in one launch (dqo_attach_decide).  Checked
    link by link: the two launches against their torch restatements (dqo_mapgrowth.temp_points_pixels /
    temp_points_attach_mask_per_object) bit for bit; the sparse render against the FULL-frame gated render at the candidates' pixels;
    the answer against the reference's boolean-index chain (temp_points_attach_indices with temp_obj / stable_obj) over the full-frame
    render — equal except where the chain's matmul projection and the element-wise one put a candidate into different pixels."""
    torch, M = mg
    from dqo_harness import mapping, scenes
    from dqo_harness.fused_mapping import FusedMapper, tile_object_sets
    dev, cam, scene, settings, gt_color, gt_depth, mask = _growth_problem()
    go = np.asarray(scene["obj_id"], np.int32)
    with torch.no_grad():
        hit = mapping.render(settings, mapping.GaussianParams(scene, dev).activated())["depth_index_map"][0].cpu().numpy()
    po = np.where(hit >= 0, go[np.clip(hit, 0, None)], -1).astype(np.int32)
    fm = FusedMapper(scene, settings, dev).set_object_gate(go, po).reserve(500)
    stable = torch.zeros(fm.P, dtype=torch.bool, device=dev)
    stable[: 12000 : 2] = True
    new = scenes.surfel_room(85, 5000, n_objects=8)
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    nx, nop = t(new["xyz"]), t(new["opacity"]).reshape(-1, 1)
    nx[3::97] += 40.0  # some candidates far outside the image
    nop[7::11] = 0.05  # some below unstable_opacity_low: never attached
    nobj = torch.tensor(np.asarray(new["obj_id"], np.int32), device=dev)
    got = fm._temp_points_attach(nx, nop, stable, 0.1, temp_obj=nobj)
    H, W = cam.H, cam.W
    fx, fy = W / (2.0 * cam.tanfovx), H / (2.0 * cam.tanfovy)
    K = torch.tensor([[fx, 0.0, cam.cx], [0.0, fy, cam.cy], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    w2c = settings.viewmatrix.T.contiguous()
    # (1) the pixels and the sparse gate
    uv, inside = M.temp_points_pixels(nx, w2c, K, W, H)
    assert 0 < int((~inside).sum().item()) < nx.shape[0]
    lin_t = torch.where(inside, uv[:, 1] * W + uv[:, 0], torch.full_like(uv[:, 0], -1))
    lin, sparse, tsets = M.attach_pixels(nx, settings.viewmatrix, fx, fy, cam.cx, cam.cy, W, H, fm.pixel_object)
    assert torch.equal(lin.long(), lin_t)
    sparse_t = torch.full((H * W,), -1, dtype=torch.int32, device=dev)
    sparse_t[lin_t[inside]] = fm.pixel_object.reshape(-1)[lin_t[inside]]
    assert torch.equal(sparse, sparse_t) and int((sparse >= 0).sum().item()) > 100
    assert torch.equal(tsets, tile_object_sets(sparse_t.reshape(H, W)))
    # (2) the sparse render is the full-frame gated render at the candidates' pixels
    sm = stable & fm.alive.bool()
    op, sc, rot = fm.activate()
    data = dict(xyz=torch.where(sm[:, None], fm.xyz, fm._park_position()[None, :]), opacity=op, scales=sc, rotations=rot, shs=fm.shs)
    with torch.no_grad():
        full = mapping.render(settings, data, object_gate=(fm.gaussian_object, fm.pixel_object.reshape(-1)))
        thin = mapping.render(settings, data, object_gate=(fm.gaussian_object, sparse, tsets))
    px = lin_t[inside]
    for k in ("color_index_map", "color_hit_weight", "depth_index_map", "depth", "render"):
        a, b = full[k].reshape(full[k].shape[0], -1)[:, px], thin[k].reshape(thin[k].shape[0], -1)[:, px]
        assert torch.equal(a, b), k
    # (3) the decision: one launch = the masked torch arithmetic, on either render
    want = M.temp_points_attach_mask_per_object(nx, nop, nobj, uv, inside, W, H, full["color_index_map"], full["color_hit_weight"], fm.xyz,
                                                lambda r: fm.normals(r), fm.gaussian_object, fm.add_depth_thres, 0.1)
    for r in (full, thin):
        dec = M.attach_decide(nx, nop, nobj, lin, r["color_index_map"], r["color_hit_weight"], fm.xyz, fm.scaling_raw, fm.rotation_raw,
                              fm.gaussian_object, fm.add_depth_thres, 0.1)
        assert torch.equal(dec.bool(), want)
    assert torch.equal(torch.sort(got).values, want.nonzero().reshape(-1)) and 0 < got.numel() < nx.shape[0]
    # (4) ... and the reference's chain over the full frame
    cim = torch.where((full["color_index_map"] == 0) & (full["color_hit_weight"] == 0), torch.full_like(full["color_index_map"], -1),
                      full["color_index_map"])
    chain = M.temp_points_attach_indices(nx, nop, w2c, K, W, H, cim, fm.xyz, lambda r: fm.normals(r), fm.add_depth_thres, 0.1,
                                         temp_obj=nobj, stable_obj=fm.gaussian_object)
    c_mask = torch.zeros_like(want)
    c_mask[chain] = True
    uv_mm = (nx @ w2c[:3, :3].T + w2c[:3, 3]) @ K.T
    uv_mm = (uv_mm[:, :2] / uv_mm[:, 2:]).long()
    moved = (uv_mm != uv).any(dim=1) & inside  # (the two projections differ in the last bit: a candidate on a pixel edge may move)
    assert not bool(((c_mask != want) & ~moved).any()) and int(moved.sum().item()) < 10


def test_attach_kernels_on_degenerate_candidates(mg):
    """dqo_attach_pixels / dqo_attach_decide with no candidates and with non-finite / huge coordinates: such a candidate is outside the
    image (lin = -1), owns no pixel of the sparse gate and never attaches; nothing is read or written out of bounds.  (A point BEHIND the
    camera projects to a pixel like one in front of it — the reference's get_uv has no depth test, scene/cameras.py:207-214 — and is
    judged like any other: the torch restatement says the same.)"""
    torch, M = mg
    from dqo_harness import mapping, scenes
    dev = torch.device("cuda")
    cam, scene = scenes.make_config(1, P=2000)
    settings = mapping.make_settings(cam, dev)
    H, W = cam.H, cam.W
    fx, fy = W / (2.0 * cam.tanfovx), H / (2.0 * cam.tanfovy)
    K = torch.tensor([[fx, 0.0, cam.cx], [0.0, fy, cam.cy], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    po = torch.zeros((H * W,), dtype=torch.int32, device=dev)
    lin, sparse, tsets = M.attach_pixels(torch.empty((0, 3), device=dev), settings.viewmatrix, fx, fy, cam.cx, cam.cy, W, H, po)
    assert lin.numel() == 0 and bool((sparse == -1).all()) and bool((tsets == 0).all())
    campos = settings.campos.reshape(1, 3).float()
    fwd = settings.viewmatrix[:3, 2].reshape(1, 3).float()  # camera z axis in world coordinates
    inf, nan = float("inf"), float("nan")
    pts = torch.cat([campos - 2.0 * fwd, campos + 2.0 * fwd,
                     torch.tensor([[nan, 0.0, 1.0], [inf, 0.0, 1.0], [0.0, -inf, 1.0], [1e30, 1e30, 1e30]], device=dev)]).contiguous()
    lin, sparse, tsets = M.attach_pixels(pts, settings.viewmatrix, fx, fy, cam.cx, cam.cy, W, H, po)
    uv, inside = M.temp_points_pixels(pts, settings.viewmatrix.T.contiguous(), K, W, H)
    assert torch.equal(lin >= 0, inside) and inside.tolist() == [True, True, False, False, False, False]
    assert torch.equal(lin[:2].long(), uv[:2, 1] * W + uv[:2, 0]) and int((sparse >= 0).sum().item()) == int(torch.unique(lin[:2]).numel())
    P = 5
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    rot = z(P, 4)
    rot[:, 0] = 1
    n = pts.shape[0]
    ids = torch.zeros(n, dtype=torch.int32, device=dev)
    hit = torch.full((1, H, W), -1, dtype=torch.int32, device=dev)
    dec = M.attach_decide(pts, torch.full((n, 1), 0.99, device=dev), ids, lin, hit, z(1, H, W), z(P, 3), z(P, 3), rot,
                          torch.zeros(P, dtype=torch.int32, device=dev), 0.1, 0.1)
    assert int(dec.sum().item()) == 0
    hit[:] = 2  # every pixel shows Gaussian 2, placed on the second candidate: that one attaches, the non-finite ones never do
    sx = (campos + 2.0 * fwd).repeat(P, 1).contiguous()
    dec = M.attach_decide(pts, torch.full((n, 1), 0.99, device=dev), ids, lin, hit, torch.ones((1, H, W), device=dev), sx, z(P, 3), rot,
                          torch.zeros(P, dtype=torch.int32, device=dev), 0.1, 0.1)
    assert dec.tolist()[1:] == [1, 0, 0, 0, 0]
    assert M.attach_decide(pts[:0], z(0, 1), ids[:0], lin[:0], hit, z(1, H, W), z(P, 3), z(P, 3), rot,
                           torch.zeros(P, dtype=torch.int32, device=dev), 0.1, 0.1).numel() == 0


def test_fused_growth_decisions_equal_their_torch_chains(mg):
    """dqo_growth_scales / dqo_growth_inside / dqo_error_maps (one launch each) against the torch chains they replace
    (dqo_mapgrowth.*(fused=False): the reference's statements): identical masks, identical floats."""
    torch, M = mg
    from dqo_harness import mapping, scenes
    dev = torch.device("cuda")
    cam, scene = scenes.make_config(3, P=30000)
    new = scenes.surfel_room(91, 7000, n_objects=8)
    t = lambda a, dt=np.float32: torch.tensor(np.ascontiguousarray(a, dt), device=dev)
    ex, eo = t(scene["xyz"]), t(scene["obj_id"], np.int32)
    er = (torch.exp(t(scene["scales"]).log()).sum(1) - t(scene["scales"]).min(1).values) / 2
    nx, no = t(new["xyz"]), t(new["obj_id"], np.int32)
    nr = (t(new["scales"]).sum(1) - t(new["scales"]).min(1).values) / 2
    eo2 = eo.clone()
    eo2[::17] = -1                    # group-less rows (spare rows of a reserved map)
    for cell in (None, (8.0, 4.0, 8.0)):
        a = M.temp_points_filter_mask_per_object(nx, no, ex, er, eo2, cell=cell, fused=True)
        b = M.temp_points_filter_mask_per_object(nx, no, ex, er, eo2, cell=cell, fused=False)
        assert torch.equal(a, b) and 0 < int(a.sum().item()) < nx.shape[0]
        sa, ia = M.update_geometry_scales_per_object(nx, no, nr, ex, er, eo2, 0.001, 0.05, cell=cell, fused=True)
        sb, ib = M.update_geometry_scales_per_object(nx, no, nr, ex, er, eo2, 0.001, 0.05, cell=cell, fused=False)
        assert torch.equal(ia, ib) and torch.equal(sa, sb)
        assert 0 < int(ia.sum().item()) < nx.shape[0] and float(sa.min().item()) >= float(np.float32(0.001)) and float(sa.max().item()) <= float(np.float32(0.05))
        assert int(((sa > 0.0011) & (sa < 0.0499)).sum().item()) > 100   # (not all clipped: the arithmetic in between is compared too)
    # no existing map / a single new point / an object without neighbours
    s1, i1 = M.update_geometry_scales_per_object(nx, no, nr, ex[:0], er[:0], eo[:0], 0.001, 0.05, fused=True)
    s2, i2 = M.update_geometry_scales_per_object(nx, no, nr, ex[:0], er[:0], eo[:0], 0.001, 0.05, fused=False)
    assert torch.equal(s1, s2) and torch.equal(i1, i2)
    s1, i1 = M.update_geometry_scales_per_object(nx[:1], no[:1], nr[:1], ex, er, eo, 0.001, 0.05, fused=True)
    s2, i2 = M.update_geometry_scales_per_object(nx[:1], no[:1], nr[:1], ex, er, eo, 0.001, 0.05, fused=False)
    assert torch.equal(s1, s2) and torch.equal(i1, i2)
    lone = torch.full_like(no, 63)
    s1, i1 = M.update_geometry_scales_per_object(nx[:50], lone[:50], nr[:50], ex, er, eo, 0.001, 0.05, fused=True)
    s2, i2 = M.update_geometry_scales_per_object(nx[:50], lone[:50], nr[:50], ex, er, eo, 0.001, 0.05, fused=False)
    assert torch.equal(s1, s2) and torch.equal(i1, i2)
    # the error images
    settings = mapping.make_settings(cam, dev)
    with torch.no_grad():
        out = mapping.render(settings, mapping.GaussianParams(scene, dev).activated())
    rng = np.random.default_rng(2)
    gt_color = (out["render"] + t(rng.normal(0, 0.05, out["render"].shape))).contiguous()
    gt_depth = (out["depth"] + t(rng.normal(0, 0.05, out["depth"].shape))).contiguous()
    gt_depth[0, ::7] = 0
    mask = torch.tensor(rng.uniform(size=(cam.H, cam.W)) < 0.8, device=dev)
    for m in (mask, None):
        ca, da = M.error_maps(gt_color, gt_depth, out["render"], out["depth"], out["depth_index_map"], m, fused=True)
        cb, db = M.error_maps(gt_color, gt_depth, out["render"], out["depth"], out["depth_index_map"], m, fused=False)
        assert torch.equal(ca, cb) and torch.equal(da, db) and float(da.max().item()) > 0 and float(ca.max().item()) > 0
