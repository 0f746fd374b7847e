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
