"""GPU: the bench.py contract -- ONE JSON line with the keys the driver reads, a `roofline` object whose numbers are
self-consistent, a `cpu_baseline` timed by the oracle, and the per-stage lines.  Reduced sizes (2 decoder layers, 8 sequences of
512 tokens): this checks the plumbing, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _common(d, n_gpus=1):
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert k in d and isinstance(d[k], t), (k, d.get(k))
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["n_gpus"] == n_gpus
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert "traffic" in r


def test_joint_line_contract():
    d = _run(["--steps", "2", "--warmup", "1", "--layers", "2", "--batch", "8", "--seq", "512", "--hist", "10", "--pool", "64", "--cpu-budget", "20"])
    _common(d)
    assert d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "user-sequences/sec"
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) <= 0.02 * d["value"]          # value = sequences / measured time
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == d["unit"] and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    a = d["attention"]
    assert a["fwd"]["launches"] == 2 * 2 and a["bwd"]["launches"] == 2 * 2 and 0 < a["fwd"]["frac_of_peak"] < 1       # layers x steps
    assert d["comm"] == {"backend": None, "ranks": 1}


@pytest.mark.parametrize("workload,unit", [("item", "items/sec"), ("user", "user-sequences/sec")])
def test_stage_line_contract(workload, unit):
    d = _run(["--workload", workload, "--steps", "2", "--warmup", "1", "--batch", "16", "--hist", "4", "--cpu-budget", "20"])
    _common(d)
    assert d["unit"] == unit and d["config"]["workload"] == workload
    assert d["cpu_baseline"]["unit"] == unit and d["cpu_baseline"]["value"] > 0


def test_item_c1_line_contract():
    """BASELINE configs[0] (C1): eval forward + reconstruction metrics on the GPU, launch by launch and as one hipGraph replay, beside the
    oracle's CPU figure (round 6: every BASELINE config has a number in the driver's line -- `stages.item_c1`)."""
    d = _run(["--workload", "item_c1", "--cpu-budget", "10"])
    assert d["unit"] == "items/sec" and d["value"] > 0 and d["config"]["workload"] == "item_c1" and d["config"]["per_gpu_batch"] == 16
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    lat = d["latency_us"]
    assert lat["launch_by_launch"] > 0
    g = d["hip_graph"]
    if g["captured"]:
        assert g["bit_identical_to_the_launch_by_launch_call"] and lat["hip_graph"] > 0
    assert 0.0 < d["config"]["eval_mse"] < 1e4          # (random-init heads: the value only has to be a finite masked MSE)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "items/sec" and c["value"] > 0
