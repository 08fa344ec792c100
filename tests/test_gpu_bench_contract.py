"""bench.py's output contract on a small Grid: exactly ONE JSON line on stdout with the driver's keys, the
`roofline` object of the dominant kernel and the `cpu_baseline` object (the real reference on one host core,
or the CPU restatement when oracle/_ref did not travel)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        text=True, cwd=ROOT, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, pr.stdout[-2000:]
    return json.loads(lines[0])


def test_one_json_line_with_roofline_and_cpu_baseline():
    d = run_bench("--nx", "64", "--steps", "3", "--warmup", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "cell-updates/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "64x64x64" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 64 ** 3 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "cell-updates/s" and c["sample"]


def test_other_workloads_keep_the_contract():
    for args in (("--problem", "blast", "--nx", "48"), ("--smr", "--nx", "32"), ("--integrator", "vl", "--nx", "48"), ("--order", "3", "--nx", "48")):
        d = run_bench(*args, "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
        assert d["value"] > 0 and d["roofline"]["achieved"] > 0 and "workload" in d["config"]
