"""GPU: the reference's optimise LOOP as an N = 2 job — two processes on this box's one GPU, gloo as the transport (RCCL refuses two
ranks on one device; the collective calls are the same): every rank replays its object shard's frame set in the reference's schedule
with the packed all-reduce of the loss sums behind every iteration, and checks itself against the unsharded job."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_window_schedule_as_a_two_rank_job(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "window_gloo_worker.py"), str(r), "2", str(port),
                               str(tmp_path / f"r{r}.json")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-1000:], se[-3000:])
    res = [json.load(open(tmp_path / f"r{r}.json")) for r in range(2)]
    assert sorted(res[0]["objects"] + res[1]["objects"]) == sorted(set(res[0]["objects"] + res[1]["objects"]))  # disjoint shards
    assert res[0]["P_shard"] + res[1]["P_shard"] == 16000 and min(r["P_shard"] for r in res) > 0
    assert all(r["iterations"] == 8 and r["worst_rel_loss_diff"] < 5e-6 and r["confidence_max"] >= 2 for r in res)
