"""GPU: RCCL itself, on the one GPU a test box has.  Every multi-rank run so far was gloo on one device (RCCL refuses two ranks on one
GPU), so `init_process_group("nccl", device_id=...)`, the first all-reduce, and `PackedAllReduce.reduce_async` in flight beside hipGraph
replays had never met the library they are written for.  A ONE-rank communicator runs anywhere: bench.py --force-collective brings the
nccl (= RCCL) backend up as rank 0 of 1 and runs the timed loop exactly as a rank of an N-rank job does (one graph launch per
iteration, the packed all-reduce of the loss sums started asynchronously after it on a ring of staging buffers, finish(), barrier,
self checks).  Started as a fresh child process before it touches the GPU, like bench.launch_ranks starts the ranks.  New design, no
reference line (the reference is single-GPU: SURVEY.md section 8e)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ipc_legacy", ["unset", "0"])
def test_rccl_one_rank_group_beside_graph_replays(ipc_legacy):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DQO_BENCH_BACKEND", "HSA_ENABLE_IPC_MODE_LEGACY"):
        env.pop(k, None)
    if ipc_legacy != "unset":
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = ipc_legacy
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "1000", "--warmup", "20",
                        "--no-cpu-baseline", "--no-pmc", "--no-aux"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert c["backend"] == "nccl (RCCL)" and c["rccl_ranks"] == 1 and d["n_gpus"] == 1 and c["selfcheck"] == "ok"
    assert ("HSA_ENABLE_IPC_MODE_LEGACY=" + ipc_legacy) in c["backend_note"]
    assert c["graph_unroll"] == 1  # one launch per iteration, the all-reduce behind each: what a rank of an N-rank job does
    ar = c["allreduce"]
    assert ar["payload_bytes"] == 32 and ar["ms_per_op"] > 0
    assert d["steps"] == 1000 and d["value"] > 200 and d["loss"][0] > 0
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rec = dict(ipc_legacy=ipc_legacy, value=d["value"], ms_per_step=d["ms_per_step"], steps=d["steps"], backend=c["backend"],
               backend_note=c["backend_note"], allreduce=ar, loss=d["loss"], selfcheck=c["selfcheck"], graph_unroll=c["graph_unroll"])
    path = os.path.join(out, "rccl_world1.json")
    prev = json.load(open(path)) if os.path.exists(path) else {}
    prev[ipc_legacy] = rec
    json.dump(prev, open(path, "w"), indent=1)
