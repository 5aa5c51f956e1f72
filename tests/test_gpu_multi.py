"""The multi-GPU paths of one process: the sharded encode as a stream of steps (w3_encode_sharded_submit / _wait), a failing shard,
and — only where the box has TWO OR MORE devices — the same calls over distinct devices with the RCCL gather, and bench.py --gpus 2
over RCCL.  On the 1-GPU box the multi-device tests skip; the first box with real peers runs them with no new code."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import weath3rb0i_amd as w3
from tests.synth import lcg_text, markov_text

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_devices():
    import torch
    return torch.cuda.device_count()


def shards_of(ctx0, data, bs, k, devices=None):
    import torch
    n = len(data)
    nb = (n + bs - 1) // bs
    out = []
    for r in range(k):
        b0, b1 = C.c_size_t(), C.c_size_t()
        assert ctx0.lib.w3_shard_range(nb, k, r, C.byref(b0), C.byref(b1)) == 0
        t = torch.from_numpy(data[min(b0.value * bs, n):min(b1.value * bs, n)].copy())
        out.append(t.to("cuda:%d" % (devices[r] if devices else 0)))
    return out


def apm012():
    return w3.APM(w3.BestOfTwoModel(w3.BestOfTwoModel(w3.Order0(), w3.Order1()), w3.OrderN(27, 3)))


def run_stream_of_steps(cs, devices, transport, root, oracle=None):
    """steps in flight on every context, gathered out of order; each step's gathered output = the single-context call's"""
    import torch
    from weath3rb0i_amd import _lib as L
    k = len(cs)
    bs = 2048
    inputs = [np.frombuffer(markov_text(300 * 1024 + 77, seed=11), dtype=np.uint8).copy(), np.frombuffer(lcg_text(200 * 1024, seed=12) + markov_text(64 * 1024 + 5, seed=13), dtype=np.uint8).copy(),
              np.frombuffer(markov_text(bs * 2 + 3, seed=14), dtype=np.uint8).copy()]   # (the last one: empty shards when k > 2)
    models = [apm012(), w3.BestOfTwoModel(w3.Order0(), w3.Order1()), w3.init_model()]
    want = {}
    for i, data in enumerate(inputs):
        for m, model in enumerate(models):
            want[(i, m)] = cs[0].encode_blocks(model, data, bs)
    rdev = "cuda:%d" % (devices[root] if devices else 0)
    outs = [(torch.empty(2 * max(len(d) for d in inputs) + 8192, dtype=torch.uint8, device=rdev), torch.zeros(256, dtype=torch.int32, device=rdev)) for _ in range(4)]
    depth = w3.sharded_max_in_flight(cs, models[0], [len(inputs[0]) // k] * k, bs)
    assert depth == 4
    pending = []
    keep = []
    rng = np.random.default_rng(3)
    refused = False
    for step in range(11):
        if len(pending) == depth:
            if not refused:
                with pytest.raises(w3.W3Error) as e:
                    w3.encode_sharded_submit(cs, models[0], keep[-1], bs)
                assert e.value.code == L.W3_E_INVALID
                with pytest.raises(w3.W3Error) as e:   # synchronous entry points are refused while sharded steps are in flight
                    cs[0].encode_blocks(w3.Order0(), inputs[2], bs)
                assert e.value.code == L.W3_E_INVALID
                refused = True
            sj, i, m, slot = pending.pop(int(rng.integers(0, len(pending))))
            totals = w3.encode_sharded_wait(cs, sj, outs[slot][0], outs[slot][1], root=root, transport=transport)
            w_out, w_lens = want[(i, m)]
            assert sum(totals) == len(w_out), (step, i, m)
            assert outs[slot][1][:len(w_lens)].cpu().numpy().astype(np.uint32).tolist() == w_lens.tolist(), (step, i, m)
            assert outs[slot][0][:len(w_out)].cpu().numpy().tobytes() == w_out.tobytes(), (step, i, m)
        i, m = step % len(inputs), step % len(models)
        sh = shards_of(cs[0], inputs[i], bs, k, devices)
        keep.append(sh)
        torch.cuda.synchronize()
        used = {p[3] for p in pending}
        slot = min(s for s in range(4) if s not in used)
        pending.append((w3.encode_sharded_submit(cs, models[m], sh, bs), i, m, slot))
    for sj, i, m, slot in pending:
        totals = w3.encode_sharded_wait(cs, sj, outs[slot][0], outs[slot][1], root=root, transport=transport)
        w_out, w_lens = want[(i, m)]
        assert sum(totals) == len(w_out) and outs[slot][0][:len(w_out)].cpu().numpy().tobytes() == w_out.tobytes(), (i, m)


def test_sharded_stream_of_steps_one_device():
    """w3_encode_sharded_submit / _wait on the 1-GPU box: three contexts on the one device (device-copy gather), then ONE context through
    the RCCL transport (communicator + sizes all-gather really run)."""
    cs = [w3.Context(0) for _ in range(3)]
    try:
        run_stream_of_steps(cs, None, "auto", 0)
        run_stream_of_steps(cs, None, "peer_copy", 2)
        run_stream_of_steps(cs[:1], None, "rccl", 0)
    finally:
        for c in cs:
            c.close()


def test_failing_shard_leaves_no_group_open_and_nothing_in_flight():
    """One shard fails (its context is pinned to the two-phase path and its shard is too short for it): the one-shot sharded call
    returns that shard's error — and the NEXT calls on the same contexts, RCCL transport included, work: no RCCL group was left open,
    no job in flight.  The stream form reports the same error from its wait."""
    import torch
    from weath3rb0i_amd import _lib as L
    cs = [w3.Context(0) for _ in range(2)]
    try:
        bs = 4096
        model = w3.BestOfTwoModel(w3.Order0(), w3.Order1())
        data = np.frombuffer(markov_text(bs * 4 + 5, seed=21), dtype=np.uint8).copy()   # shard 1 = two blocks + 5 bytes ... (k = 2: blocks 2..4)
        nb = 5
        d_out = torch.empty(2 * len(data) + 64 * nb + 64, dtype=torch.uint8, device="cuda")
        d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
        want, wlens = cs[0].encode_blocks(model, data, bs)
        tail = torch.from_numpy(data[bs * 4:].copy()).cuda()    # 5 bytes: below the two-phase path's 8
        head = torch.from_numpy(data[:bs * 4].copy()).cuda()
        cs[1].set_path("twophase")
        with pytest.raises(w3.W3Error) as e:
            w3.encode_blocks_sharded_device(cs, model, [head, tail], bs, d_out, d_lens)
        assert e.value.code == L.W3_E_UNSUPPORTED and "shard 1" in str(e.value)
        sj = w3.encode_sharded_submit(cs, model, [head, tail], bs)   # (the short shard runs inside submit; its error comes from the wait)
        with pytest.raises(w3.W3Error) as e:
            w3.encode_sharded_wait(cs, sj, d_out, d_lens)
        assert e.value.code == L.W3_E_UNSUPPORTED and "shard 1" in str(e.value)
        cs[1].set_path("auto")
        sj = w3.encode_sharded_submit(cs, model, [head, tail], bs)
        totals = w3.encode_sharded_wait(cs, sj, d_out, d_lens)
        assert sum(totals) == len(want) and d_out[:len(want)].cpu().numpy().tobytes() == want.tobytes()
        totals = w3.encode_blocks_sharded_device(cs, model, [head, tail], bs, d_out, d_lens)
        assert sum(totals) == len(want) and d_out[:len(want)].cpu().numpy().tobytes() == want.tobytes()
        totals = w3.encode_blocks_sharded_device(cs[:1], model, [torch.from_numpy(data).cuda()], bs, d_out, d_lens, transport="rccl")
        assert sum(totals) == len(want) and d_out[:len(want)].cpu().numpy().tobytes() == want.tobytes()
        out, lens = cs[1].encode_blocks(model, data, bs)
        assert out.tobytes() == want.tobytes()
    finally:
        for c in cs:
            c.close()


@pytest.mark.skipif("n_devices() < 2", reason="needs two devices: the RCCL gather between real peers")
def test_two_devices_sharded_encode_over_rccl():
    """Two DISTINCT devices: the one-shot sharded call and the stream of steps, RCCL transport (grouped ncclSend / ncclRecv over xGMI)
    and peer copies, root on either device."""
    import torch
    k = min(n_devices(), 4)
    devices = list(range(k))
    cs = [w3.Context(d) for d in devices]
    try:
        bs = 4096
        model = apm012()
        data = np.frombuffer(markov_text(600 * 1024 + 123, seed=31), dtype=np.uint8).copy()
        nb = (len(data) + bs - 1) // bs
        want, wlens = cs[0].encode_blocks(model, data, bs)
        for transport, root in (("auto", 0), ("rccl", k - 1), ("peer_copy", 1)):
            d_out = torch.empty(2 * len(data) + 64 * nb + 64, dtype=torch.uint8, device="cuda:%d" % devices[root])
            d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda:%d" % devices[root])
            sh = shards_of(cs[0], data, bs, k, devices)
            for d in devices:
                torch.cuda.synchronize(d)
            totals = w3.encode_blocks_sharded_device(cs, model, sh, bs, d_out, d_lens, root=root, transport=transport)
            assert sum(totals) == len(want), (transport, root)
            assert d_lens.cpu().numpy().astype(np.uint32).tolist() == wlens.tolist() and d_out[:len(want)].cpu().numpy().tobytes() == want.tobytes(), (transport, root)
        run_stream_of_steps(cs, devices, "rccl", 0)
        run_stream_of_steps(cs, devices, "auto", k - 1)
    finally:
        for c in cs:
            c.close()


@pytest.mark.skipif("n_devices() < 2", reason="needs two devices: bench.py --gpus 2 over RCCL")
def test_two_devices_bench_over_rccl():
    """`bench.py --gpus 2` as the driver runs it (the ranks started by bench.py itself, one per GPU, RCCL exchange): both readings in
    the line, the exchange's totals from both ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", "20000000"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["bytes_total"] == 20000000 and d["config"]["ranks_seen_by_rccl"] == 2
    assert "RCCL" in d["config"]["exchange"] and d["weak"]["bytes_total"] == 40000000 and d["value"] > 0 and d["weak"]["value"] > 0
