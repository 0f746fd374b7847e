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
