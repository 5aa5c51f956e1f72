"""bench.py's contract on a small input: ONE JSON line with the metric, the roofline of the dominant kernel and the CPU baseline
(the driver parses exactly this), bit-exactness of the sampled blocks against the oracle included."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "6000000", *extra],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "MiB/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 6000000 / (d["ms_per_step"] * 1e-3) / 2**20) / d["value"] < 0.02
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["bit_exact_vs_gpu"] is True and len(cb["legs"]) == 2 and cb["legs"][0]["cores"] == 1
    assert d["floors"]["coder_floor_ms"] > 0 and d["floors"]["bit_steps_per_lane"] == 8 * 65536
    assert d["predict_phase"]["algorithmic_bytes_per_step"] == 115 * 6000000          # 3 leaves x 17 B + 2 wide leaves x 32 B of record passes
    assert d["reference_stream_model"]["value"] > 0


def test_bench_strong_scaling_flag_and_other_models():
    d = run_bench("--scaling", "strong", "--model", "order012", "--no-cpu-baseline")
    assert d["scaling"] == "strong" and d.get("cpu_baseline") is None and "ONE stream" in d["config"]["workload"]
    d = run_bench("--model", "default", "--no-cpu-baseline", "--no-ref-model")
    assert "k_achash" in d["roofline"]["kernel"] or "k_coder" in d["roofline"]["kernel"]
