"""CPU: bench.py's rank resolution — `--gpus N` without a launcher starts N ranks as a child job before anything touches the GPU;
under a launcher WORLD_SIZE must equal --gpus (a mismatch exits non-zero instead of silently measuring one GPU)."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**kw):
    d = dict(gpus=1, as_shard=None, inner=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def test_resolve_world(monkeypatch):
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.resolve_world(_args(gpus=1)) == (0, 1, 0)
    # analysis modes stay one process whatever --gpus says
    assert bench.resolve_world(_args(gpus=4, as_shard="1/4")) == (0, 1, 0)
    assert bench.resolve_world(_args(gpus=4, inner=True)) == (0, 1, 0)
    # no launcher, N > 1: the ranks run as a child job and the parent only relays the exit code
    calls = []
    monkeypatch.setattr(bench, "launch_ranks", lambda a: calls.append(a.gpus) or 7)
    a = _args(gpus=8)
    assert bench.resolve_world(a) is None and a._rc == 7 and calls == [8]
    # under a launcher
    monkeypatch.setenv("WORLD_SIZE", "2"), monkeypatch.setenv("RANK", "1"), monkeypatch.setenv("LOCAL_RANK", "1")
    assert bench.resolve_world(_args(gpus=2)) == (1, 2, 1)
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(_args(gpus=8))
    assert "WORLD_SIZE=2" in str(e.value)


def test_plain_gpus_n_starts_n_ranks_and_relays_the_exit_code():
    """`python bench.py --gpus 2` on a box without a GPU: two ranks are started (each reports that it needs a GPU and exits non-zero),
    the parent never imports the HIP library and exits non-zero as well — it does not fall back to one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=env)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by test_two_rank_rehearsal_on_one_gpu")
    assert r.returncode != 0
    # (the launcher ends the other rank as soon as the first one fails: at least one rank has said why, and its failure summary lists two)
    assert r.stderr.count("bench.py needs a GPU") >= 1, r.stderr[-2000:]
    assert "local_rank: 0" in r.stderr and "local_rank: 1" in r.stderr, r.stderr[-2000:]


def test_world_size_mismatch_exits_nonzero():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 8" in r.stderr


def test_list_split_choices_are_host_logic():
    """FusedMapper.pick_list_split_pair (which lists the blend kernels share between eight waves: by tile count and longest list) and
    bench.parse_list_split — pure host logic, no GPU needed."""
    import types
    sys.path[:0] = [os.path.join(ROOT, "dqo-map_amd")]
    import torch
    from dqo_harness.fused_mapping import FusedMapper
    import bench
    st = types.SimpleNamespace(image_width=1200, image_height=680)  # 75 x 43 = 3225 tiles
    pair = FusedMapper.pick_list_split_pair
    tiles = lambda n: torch.ones(n, dtype=torch.int32)
    assert pair("auto", None, st, 7000) == (2048, 0) and pair("auto", None, st, 3000) == (0, 0)      # a full frame: forward only, long tail only
    assert pair("auto", tiles(2000), st, 3000) == (1024, 0) and pair("auto", tiles(2000), st, 1500) == (0, 0)  # half a frame
    assert pair("auto", tiles(1000), st, 1500) == (1024, 1024) and pair("auto", tiles(700), st, 1500) == (512, 512)
    assert pair("auto", tiles(300), st, 1500) == (256, 256) and pair("auto", tiles(300), st, 900) == (0, 0)   # no list worth cutting
    mask = tiles(3225)
    mask[500:] = 0
    assert pair("auto", mask, st, 5000) == (256, 256)  # masked-out tiles do not count
    assert pair(300, None, st) == (300, 300) and pair(0, None, st) == (0, 0) and pair((512, 0), None, st) == (512, 0)
    for bad in ((512, 256), -1):
        with pytest.raises(ValueError):
            pair(bad, None, st)
    assert FusedMapper.pick_list_split("auto", tiles(300), st, 1500) == 256 and FusedMapper.pick_list_split("auto", None, st, 7000) == 0
    assert bench.parse_list_split("auto") == "auto" and bench.parse_list_split("512") == 512 and bench.parse_list_split("2048,0") == (2048, 0)
