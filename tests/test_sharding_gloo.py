"""N > 1 path on CPU: world_size-2 gloo processes exercise the per-object partition and the packed all-reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dqo_harness import scenes, sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cam, scene = scenes.make_config(3, P=20000)
    mine, assignment = sharding.shard_scene(scene, rank, world)
    # synthetic "render": a per-pixel value that depends only on the owning object, so per-object masked losses add up
    rng = np.random.default_rng(0)
    pix_obj = rng.integers(-1, 8, size=(cam.H, cam.W))
    img = rng.uniform(size=(cam.H, cam.W)).astype(np.float32)
    gt = rng.uniform(size=(cam.H, cam.W)).astype(np.float32)
    my_objs = [k for k, s in assignment.items() if s == rank]
    my_mask = np.isin(pix_obj, my_objs)
    part = np.abs(img - gt)[my_mask].sum()
    red = sharding.PackedAllReduce([("loss_sum", 1), ("pixels", 1), ("n_gauss", 1), ("cam_grad", 6)], "cpu")
    red.put("loss_sum", torch.tensor([part]))
    red.put("pixels", torch.tensor([float(my_mask.sum())]))
    red.put("n_gauss", torch.tensor([float(len(mine["xyz"]))]))
    red.put("cam_grad", torch.arange(6, dtype=torch.float32) * (rank + 1))
    red.reduce()
    # the asynchronous form (bench.py's sharded path): more rounds than staging buffers, the last one must survive finish()
    ar = sharding.PackedAllReduce([("v", 2)], "cpu")
    for it in range(2 * sharding.PackedAllReduce.RING + 1):
        if it % 2:
            ar.put("v", torch.tensor([float(it), float(rank + 1) * (it + 1)]))
            ar.reduce_async()
        else:  # straight from a source tensor
            ar.reduce_async(src=torch.tensor([float(it), float(rank + 1) * (it + 1)]))
    ar.finish()
    tm = sharding.tile_mask_from_pixel_mask(my_mask)
    np.savez(os.path.join(tmp, f"r{rank}.npz"), loss=red.get("loss_sum").numpy(), pixels=red.get("pixels").numpy(),
             n=red.get("n_gauss").numpy(), cam=red.get("cam_grad").numpy(), owned=np.array(sorted(my_objs)), tm=tm,
             total=np.abs(img - gt)[pix_obj >= 0].sum(), npix=(pix_obj >= 0).sum(), mask_sum=my_mask.sum(),
             async_v=ar.get("v").numpy())
    dist.barrier()
    dist.destroy_process_group()


def _object_scene_64x48():
    """64x48 scene with 4 objects (quartiles of the camera-space x coordinate), target = render of a perturbed copy, per-object
    screen masks from the target's depth-hit ids — the bench's construction (bench.py build_problem) at test size."""
    from oracle import oracle_lib as ol
    import util_rast as U
    cam = scenes.Camera(64, 48, 60.0, 60.0, 31.5, 23.5)
    sc = scenes.frustum_cloud(23, 700, cam, zmin=0.5, zmax=1.6)
    xc = sc["xyz"][:, 0] / sc["xyz"][:, 2]
    sc["obj_id"] = np.digitize(xc, np.quantile(xc, [0.25, 0.5, 0.75])).astype(np.int32)
    rng = np.random.default_rng(9)
    pert = dict(sc)
    pert["xyz"] = (sc["xyz"] + rng.normal(0, 0.004, sc["xyz"].shape)).astype(np.float32)
    _, tgt, _ = U.run_oracle(ol, cam, pert)
    hit = tgt["hit_depth"][0]
    pix_obj = np.where(hit >= 0, sc["obj_id"][np.clip(hit, 0, None)], -1)
    return cam, sc, tgt["color"], tgt["depth"], pix_obj


def _object_sums(cam, sc, gt_color, gt_depth, pix_obj, k):
    """Per-OBJECT masked render + loss sums with the oracle: object k's Gaussians alone, its tile mask, its pixel mask.
    Returns [sum |colour err|, mask pixels, sum |depth err|, valid depth pixels]."""
    from oracle import oracle_lib as ol
    from oracle import map_oracle as mo
    import util_rast as U
    sel = sc["obj_id"] == k
    sub = {n: v[sel] for n, v in sc.items()}
    mask = pix_obj == k
    tm = sharding.tile_mask_from_pixel_mask(mask)
    _, r, _ = U.run_oracle(ol, cam, sub, tile_mask=tm)
    tot, cl, dl, _, _ = mo.masked_loss(r["color"], r["depth"], r["hit_depth"], gt_color, gt_depth, mask)
    err = r["depth"].astype(np.float64) - gt_depth.astype(np.float64)
    valid = (r["hit_depth"] != -1) & (gt_depth > 0) & (err < 0.1) & mask[None]
    sums = np.array([np.abs(r["color"].astype(np.float64) - gt_color)[:, mask].sum(), mask.sum(), np.abs(err[valid]).sum(), valid.sum()])
    # consistency of the sums with the oracle's normalised losses
    assert abs(sums[0] / (3 * max(sums[1], 1)) - cl) <= 1e-9 + 1e-6 * cl and abs(sums[2] / max(sums[3], 1) - dl) <= 1e-9 + 1e-6 * dl
    return sums


def _worker_objects(rank, world, port, tmp):
    """Per-object losses of the ranks add up to the unsharded masked loss (L = sum_k L_k(G_k) on disjoint masks)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cam, sc, gt_color, gt_depth, pix_obj = _object_scene_64x48()
    mine, assignment = sharding.shard_scene(sc, rank, world)
    my_objs = sorted(k for k, s in assignment.items() if s == rank)
    assert sorted(np.unique(mine["obj_id"]).tolist()) == my_objs
    part = np.zeros(4)
    for k in my_objs:
        part += _object_sums(cam, mine, gt_color, gt_depth, pix_obj, k)  # from the SHARD's Gaussians only
    red = sharding.PackedAllReduce([("sum_color", 1), ("n_color", 1), ("sum_depth", 1), ("n_depth", 1)], "cpu")
    for name, v in zip(("sum_color", "n_color", "sum_depth", "n_depth"), part):
        red.put(name, torch.tensor([float(v)]))
    red.reduce_async()
    red.finish()
    np.savez(os.path.join(tmp, f"o{rank}.npz"), reduced=red.buf.numpy(), owned=np.array(my_objs))
    dist.barrier()
    dist.destroy_process_group()


def test_per_object_losses_of_two_ranks_sum_to_the_unsharded_loss(tmp_path):
    world = 2
    mp.spawn(_worker_objects, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"o{i}.npz") for i in range(world)]
    np.testing.assert_array_equal(r[0]["reduced"], r[1]["reduced"])
    assert sorted(np.concatenate([r[0]["owned"], r[1]["owned"]]).tolist()) == [0, 1, 2, 3]
    # unsharded: every object in this one process, from the full map
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    cam, sc, gt_color, gt_depth, pix_obj = _object_scene_64x48()
    total = sum(_object_sums(cam, sc, gt_color, gt_depth, pix_obj, k) for k in range(4))
    assert total[1] > 500 and total[3] > 100  # the masks are not empty
    np.testing.assert_allclose(r[0]["reduced"], total, rtol=2e-6)
    # the reported loss over ALL objects = the masked loss of the unsharded job
    got = 0.8 * r[0]["reduced"][0] / (3 * r[0]["reduced"][1]) + r[0]["reduced"][2] / r[0]["reduced"][3]
    want = 0.8 * total[0] / (3 * total[1]) + total[2] / total[3]
    assert abs(got - want) <= 2e-6 * want


def test_object_shards_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{i}.npz") for i in range(world)]
    # every rank sees the same reduced values; they equal the unsharded quantities
    for k in ("loss", "pixels", "n", "cam"):
        np.testing.assert_allclose(r[0][k], r[1][k])
    np.testing.assert_allclose(r[0]["loss"][0], r[0]["total"], rtol=1e-5)
    assert int(r[0]["pixels"][0]) == int(r[0]["npix"]) and int(r[0]["n"][0]) == 20000
    np.testing.assert_allclose(r[0]["cam"], np.arange(6) * 3.0)
    last = 2 * sharding.PackedAllReduce.RING  # reduce_async: both ranks hold the sum of the LAST round
    for i in range(world):
        np.testing.assert_allclose(r[i]["async_v"], [2.0 * last, 3.0 * (last + 1)])
    # partition: disjoint, complete
    owned = np.concatenate([r[0]["owned"], r[1]["owned"]])
    assert sorted(owned.tolist()) == list(range(8)) and len(set(owned.tolist())) == 8
    # tile masks cover the shard's pixels and only tiles that contain them
    assert r[0]["tm"].dtype == np.int32 and r[0]["tm"].shape == ((680 + 15) // 16, (1200 + 15) // 16)


def test_assignment_balanced():
    sizes = {k: n for k, n in enumerate([50000, 30000, 30000, 20000, 10000, 10000, 5000, 5000])}
    for world in (1, 2, 4, 8):
        a = sharding.assign_objects(sizes, world)
        load = [sum(sizes[k] for k, s in a.items() if s == w) for w in range(world)]
        assert sum(load) == sum(sizes.values())
        assert max(load) <= max(max(sizes.values()), 1.34 * sum(load) / world)
    pm = np.zeros((40, 50), bool)
    pm[17, 33] = True
    tm = sharding.tile_mask_from_pixel_mask(pm)
    assert tm.shape == (3, 4) and tm.sum() == 1 and tm[1, 2] == 1


def test_assignment_by_view_work():
    """shard_scene(work=...): objects go to the ranks by the work they put on screen in the current view, not by how many Gaussians
    they store; the assignment is a pure function of (obj_id, work), so every rank computes the same one without an exchange."""
    rng = np.random.default_rng(3)
    obj = rng.integers(0, 8, 4000)
    radii = rng.integers(0, 40, 4000)
    radii[obj >= 6] = 0            # two objects entirely out of view
    radii[obj == 0] *= 4           # one object that fills the screen
    work = sharding.view_work(radii)
    assert (work[radii == 0] == 0).all() and (work[radii > 0] >= 4).all()
    np.testing.assert_array_equal(sharding.view_work(np.array([0, 1, 8, 24])), [0, 4, 9, 25])  # (2r/16 + 2)^2 tiles
    scene = dict(obj_id=obj, xyz=rng.normal(size=(4000, 3)))
    per_obj = np.bincount(obj, weights=work, minlength=8)
    for world in (2, 4):
        parts = [sharding.shard_scene(scene, r, world, work=work) for r in range(world)]
        assert all(p[1] == parts[0][1] for p in parts)  # same assignment on every rank
        assert sum(len(p[0]["obj_id"]) for p in parts) == 4000
        load = np.array([sum(per_obj[k] for k, s in parts[0][1].items() if s == r) for r in range(world)])
        assert load.max() <= max(per_obj.max(), 1.34 * load.sum() / world)
        by_count = sharding.shard_scene(scene, 0, world)[1]
        load_c = np.array([sum(per_obj[k] for k, s in by_count.items() if s == r) for r in range(world)])
        assert load.max() <= load_c.max()  # never worse than the count-balanced assignment on the quantity that costs time
