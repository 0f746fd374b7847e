"""GPU: bench.py keeps its contract — one JSON line on stdout with the fields the driver reads, the metric string of BASELINE.json,
a roofline and (full run only) a cpu_baseline object, and a green self check.  Short run of the real script in a child process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-pmc",
                        "--no-aux"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # ONE JSON line, nothing else on stdout
    d = json.loads(lines[0])
    want = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["metric"] == want
    for k, t in (("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
    assert "workload" in d["config"] and d["config"]["selfcheck"] == "ok"
    ro = d["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] == "GB/s" and ro["peak"] == 8000.0
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-4 and "traffic" in ro and ro["kernel"] in d["config"]["kernel_us"]
    # a replayed iteration has neither the scan, nor the placement, nor the loss kernels
    assert not {"tile_scan_kernel", "bin_place_kernel", "loss_reduce_kernel", "loss_grad_kernel"} & set(d["config"]["kernel_us"])


def test_two_rank_rehearsal_on_one_gpu():
    """The N > 1 code path of bench.py end to end — one map sharded by object id over two ranks, hipGraph replay per rank, the packed
    asynchronous all-reduce of the loss sums, every rank's self check — with two processes sharing this box's one GPU and gloo as the
    transport (RCCL refuses two ranks on one device; the collective calls are the same).  A rehearsal of the path, not a measurement."""
    # plain `python bench.py --gpus 2`: the script starts its two ranks itself (bench.launch_ranks), before it touches the GPU
    env = dict(os.environ, DQO_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["shards"] == 2 and d["config"]["selfcheck"] == "ok"
    ar = d["config"]["allreduce"]  # the all-reduce time share
    assert ar["payload_bytes"] == 32 and ar["ms_per_op"] > 0 and 0 < ar["share_of_iteration"] < 10
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["backend"] == "gloo" and d["config"]["devices"] == [0, 0]
    assert "sharded by object id over 2 ranks" in d["config"]["workload"]
    assert 0 < d["config"]["P_shard"] < d["config"]["P"] and d["loss"][0] > 0
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    rep = d["config"]["replicas_reference"]  # every rank stepping the whole unsharded map: the aggregate of two replicas
    assert rep["unit"] == "iter/s" and rep["value"] > 0 and rep["ms_per_step"] > 0
