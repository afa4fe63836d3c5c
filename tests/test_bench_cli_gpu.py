"""The driver's contract for bench.py, checked on a short run: ONE JSON line on stdout with the agreed keys, the roofline object
measured live, the profiled steps leaving the side streams as they were."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config", ["3", "5"])
def test_bench_prints_one_contract_line(config):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "3", "--warmup", "2", "--no-others",
           "--no-cpu-baseline", "--profile-steps", "1"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["unit"] == "images/s" and d["value"] > 0
    assert abs(d["value"] - d["config"]["crops_per_gpu"] * 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]
    assert "workload" in d["config"] and d["config"]["baseline_config"] == config
    assert d["dtype"] == ("fp8" if config == "5" else "f32")
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["peak"] == (5000.0 if config == "5" else 416.7)       # the roof of the instruction issued: 2500 / 6 bf16 products
    if config != "5":
        assert abs(r["frac_vs_fp32_mfma"] - r["achieved"] / 157.3) < 1e-3
