"""bench.py's contract on a small input: ONE JSON line with the metric, the roofline of the time-dominant kernel, the CPU baseline
(the driver parses exactly this), bit-exactness of the sampled blocks against the oracle and the full device round trip included;
and the self-launch of the ranks for --gpus N."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_bench(*extra, gpus=1):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--size", "6000000", *extra],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract():
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "MiB/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["encodes_in_flight"] == 4   # (92 blocks: w3_encode_max_in_flight's free-running jobs)
    assert d["value"] > 0 and abs(d["value"] - 6000000 / (d["ms_per_step"] * 1e-3) / 2**20) / d["value"] < 0.02
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # the named kernel is the one with the longest launch, and the predict kernels are candidates
    table = d["roofline_kernels"]
    assert rf["avg_launch_ms"] == max(r["avg_launch_ms"] for r in table)
    names = " ".join(r["kernel"] for r in table)
    for k in ("k_rank_sorted<1>", "k_rank_sorted<2>", "k_partition8<1>", "k_predict_small", "k_apm0<3>", "k_coder_x5"):
        assert k in names, k
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["bit_exact_vs_gpu"] is True and cb["buffers_checked"] == 3 and len(cb["legs"]) == 2 and cb["legs"][0]["cores"] == 1
    assert d["decode"]["roundtrip_all_blocks"] is True and d["decode"]["blocks"] == 92
    assert d["floors"]["coder_floor_ms"] > 0 and d["floors"]["bit_steps_per_lane"] == 8 * 65536
    assert d["one_call_at_a_time"]["ms_per_step"] > 0 and d["floors"]["coder_floor_ms"] == d["one_call_at_a_time"]["kernel_ms_per_step"]["coder_ms"]
    assert d["predict_phase"]["algorithmic_bytes_per_step"] == 115 * 6000000          # 3 leaves x 17 B + 2 wide leaves x 32 B of record passes
    assert d["reference_stream_model"]["value"] > 0 and d["reference_stream_model_value"] == d["reference_stream_model"]["value"]
    assert "other_configs" not in d   # (only at enwik8 size and above)
    # what the driver's record keeps whole is config / roofline / cpu_baseline: the lines a reader must not miss travel there too
    assert d["config"]["reference_stream_model"]["value"] == d["reference_stream_model"]["value"] and d["config"]["decode"]["roundtrip_all_blocks"] is True
    assert "unset" in d["config"]["gpu_max_hw_queues"] or d["config"]["gpu_max_hw_queues"] == os.environ.get("GPU_MAX_HW_QUEUES")
    # the stable pick: longest launch of the one-call-at-a-time leg
    solo = d["roofline"]["solo"]
    assert solo == d["roofline_solo"] and solo["avg_launch_ms"] == max(k["ms"] for k in solo["kernels"]) or abs(solo["avg_launch_ms"] - max(k["ms"] for k in solo["kernels"])) < 1e-3
    assert abs(solo["frac"] - solo["achieved"] / solo["peak"]) < 1e-4 and "k_coder_x4" in " ".join(k["kernel"] for k in solo["kernels"])
    # PCIe-inclusive legs (never part of value)
    hp = d["host_path"]
    assert hp["calls_in_flight_pinned_MiBps"] > 0 and hp["one_call_pinned_MiBps"] > 0 and hp["one_call_pageable_MiBps"] > 0 and hp["h2d_pinned_ms"] > 0 and hp["calls_in_flight"] == 5
    assert d["config"]["host_path"]["calls_in_flight_pinned_MiBps"] == hp["calls_in_flight_pinned_MiBps"]


@pytest.mark.gpu
def test_bench_readings_models_and_sync_mode():
    d = run_bench("--scaling", "weak", "--model", "order012", "--quick", "--hw-queues", "8")
    assert d["scaling"] == "weak" and d.get("cpu_baseline") is None and "per GPU" in d["config"]["workload"] and d["config"]["gpu_max_hw_queues"] == "8"
    assert "host_path" not in d
    d = run_bench("--model", "default", "--quick", "--pipeline", "1")
    assert d["config"]["encodes_in_flight"] == 1
    assert "k_achash" in d["roofline"]["kernel"] or "k_coder" in d["roofline"]["kernel"] or "k_predict_small" in d["roofline"]["kernel"]


@pytest.mark.gpu
def test_bench_self_launch_one_rank_with_exchange():
    """--gpus N with WORLD_SIZE unset starts the ranks itself; on the 1-GPU box: through the launcher path with --force-exchange
    under an explicit one-rank torch.distributed.run, the same command line the self-launch builds for N ranks."""
    import bench
    cmd = bench.launcher_command(1, ["--gpus", "1", "--steps", "2", "--warmup", "1", "--size", "6000000", "--quick", "--force-exchange"], bench.free_port())
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["ranks_seen_by_rccl"] == 1 and "over 1 rank(s)" in d["config"]["exchange"] and d["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_both_readings():
    """The multi-rank control flow end to end on the 1-GPU box: `bench.py --gpus 2` starts two ranks ITSELF (WORLD_SIZE unset), they
    share the one GPU and exchange through gloo (--backend gloo: a rehearsal, its numbers mean nothing): strong reading = ONE stream
    cut over the ranks, weak reading attached, the exchange's totals from both ranks in the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "6000000", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["bytes_total"] == 6000000
    assert d["config"]["ranks_seen_by_rccl"] == 2 and "over 2 rank(s)" in d["config"]["exchange"]
    assert d["weak"]["bytes_total"] == 12000000 and d["weak"]["value"] > 0 and d["value"] > 0
    assert "cpu_baseline" not in d and "decode" not in d   # rank-0, N = 1 extras only


def test_launcher_command_for_n_ranks():
    """CPU: the command bench.py runs for --gpus N when nobody launched it (the driver's own form: one process per GPU,
    127.0.0.1 rendezvous); and the parent never reaches a GPU call before it (self_launch is the first thing main() does)."""
    import bench
    cmd = bench.launcher_command(2, ["--gpus", "2", "--steps", "4"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "2", "--steps", "4"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index("self_launch(") < body.index("import torch")
    a = bench.parse_args(["--gpus", "4"])
    assert a.gpus == 4 and a.scaling == "both" and a.pipeline == 0   # (0 = as many encodes in flight as w3_encode_max_in_flight allows)
