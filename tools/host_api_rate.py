"""PCIe-inclusive rates of the host-buffer entry points next to the device-resident one ("report H2D/D2H
separately"; compress() of main.rs:89-113 is file in, file out).

    python3 tools/host_api_rate.py [model] [bytes] [--chunks 0,1908,3815,7630] [--calls 8] [--json FILE]

Legs (1e9 B of the bench's synthetic text by default):
  copies          H2D of the input / D2H of the compressed streams alone, pinned and pageable
  resident        w3_encode_submit / w3_encode_wait over a device-resident input (bench.py's timed loop)
  sync pinned     ONE w3_encode_blocks call at a time (pipelined in pieces inside the call), per piece size
  sync pageable   the same from pageable numpy buffers
  calls in flight w3_encode_host_submit / w3_encode_host_wait, as many calls in flight as the context allows, pinned buffers
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(name="order012apm", n=1_000_000_000, bs=65536, chunks=(0,), calls=8, pageable=True, resident=True, host=None, log=print, tune=0, churn=0):
    import numpy as np
    import torch
    import weath3rb0i_amd as w3
    from tools import synth
    import bench
    res = {"model": name, "bytes": n, "block_size": bs}
    if host is None:
        host = synth.text(n, seed=1)
    model, mname = bench.make_model(w3, name)
    res["context_model"] = mname
    nb = (n + bs - 1) // bs
    if churn:
        # contexts that come and go before the measured one (bench.py's situation: its main context and its decode context have lived
        # and died before the host-path leg): their pipelines create the process's streams; the measured context must not end up with
        # a copy stream on a living predict stream's hardware queue (w3hip.hip: the context's stream comes from the process-wide pool)
        for _ in range(churn):
            c0 = w3.Context(0)
            small = host[: 8 << 20]
            o0, l0 = np.zeros(len(small) * 2 + 8192, dtype=np.uint8), np.zeros((len(small) + bs - 1) // bs, dtype=np.uint32)
            for _ in range(3):
                c0.encode_host_wait(c0.encode_host_submit(model, small, bs, o0, l0))
            c0.close()
    ctx = w3.Context(0)
    if tune:
        ctx.set_tune(tune)
    res["tune"] = tune
    res["gpu_max_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "unset")
    MiB = 2.0 ** 20

    pin_in = torch.empty(n, dtype=torch.uint8).pin_memory()
    pin_in.numpy()[:] = host
    cap = n // 2 + 64 * nb + 4096
    # (a first call sizes everything; its output length tells how much a D2H moves)
    out0, lens0 = ctx.encode_blocks(model, host[: min(n, 64 << 20)], bs)
    ratio = len(out0) / min(n, 64 << 20)
    cap = max(cap, int(n * ratio * 1.25) + 64 * nb + 4096)
    depth_h = ctx.host_max_in_flight(n, bs, model)
    pin_outs = [(torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(nb, dtype=torch.int32).pin_memory()) for _ in range(depth_h)]

    # ---- copies alone
    d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); d_in.copy_(pin_in, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res["h2d_pinned_ms"] = round(min(ts) * 1e3, 2)
    if pageable:
        hp = torch.from_numpy(host)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter(); d_in.copy_(hp); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res["h2d_pageable_ms"] = round(min(ts) * 1e3, 2)

    # ---- device-resident loop (what bench.py times)
    total = None
    if resident:
        depth = ctx.max_in_flight(n, bs, model)
        d_outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(depth)]
        d_lens = [torch.zeros(nb, dtype=torch.int32, device="cuda") for _ in range(depth)]
        d_tot = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(depth)]

        def run(k):
            pend = []
            for i in range(k):
                if len(pend) == depth:
                    ctx.encode_wait(pend.pop(0))
                j = i % depth
                pend.append(ctx.encode_submit(model, d_in, bs, d_outs[j], d_lens[j], d_tot[j]))
            for p in pend:
                ctx.encode_wait(p)
        run(depth + 1)
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(calls); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / calls
        total = int(d_tot[0].item())
        res["resident"] = {"ms_per_call": round(dt * 1e3, 2), "MiBps": round(n / dt / MiB, 1), "in_flight": depth}
        log("device-resident, %d in flight: %.1f ms per call = %.0f MiB/s" % (depth, dt * 1e3, n / dt / MiB))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); pin_outs[0][0][:total].copy_(d_outs[0][:total], non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res["d2h_pinned_ms"] = round(min(ts) * 1e3, 2)
        res["compressed_bytes"] = total
        del d_outs, d_lens, d_tot
    if not resident:
        tot0 = int(n * ratio)
        d_tmp = torch.empty(tot0, dtype=torch.uint8, device="cuda")
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); pin_outs[0][0][:tot0].copy_(d_tmp, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res["d2h_pinned_ms"] = round(min(ts) * 1e3, 2)
        del d_tmp
    del d_in
    torch.cuda.empty_cache()
    log("copies alone: H2D pinned %.1f ms%s, D2H of the streams pinned %s ms" % (res["h2d_pinned_ms"], (", pageable %.1f ms" % res["h2d_pageable_ms"]) if pageable else "", res.get("d2h_pinned_ms")))

    # ---- one synchronous call at a time, pinned buffers, per piece size
    import ctypes as C
    from weath3rb0i_amd import _lib as L
    spec = model.spec()

    def sync_call(in_ptr, out_ptr, out_cap, lens_ptr):
        olen = C.c_size_t()
        rc = ctx.lib.w3_encode_blocks(ctx.h, C.byref(spec), C.c_void_p(in_ptr), n, bs, C.c_void_p(out_ptr), out_cap, C.byref(olen), C.c_void_p(lens_ptr))
        assert rc == 0, (rc, ctx.lib.w3_last_error(ctx.h))
        return olen.value

    res["sync_pinned"] = []
    ref_bytes = None
    for cb in chunks:
        ctx.set_host_chunk_blocks(cb)
        o, ln = pin_outs[0]
        sync_call(pin_in.data_ptr(), o.data_ptr(), cap, ln.data_ptr())
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); tot = sync_call(pin_in.data_ptr(), o.data_ptr(), cap, ln.data_ptr()); ts.append(time.perf_counter() - t0)
        got = bytes(o.numpy()[: min(tot, 1 << 20)]) + bytes(ln.numpy()[:nb].tobytes()[: 1 << 16])
        if ref_bytes is None:
            ref_bytes = (tot, got)
        assert (tot, got) == ref_bytes, "piece size %d changes the output" % cb
        pieces = ctx.timing()["n_parts"]
        res["sync_pinned"].append({"chunk_blocks": cb, "pieces": pieces, "ms_per_call": round(min(ts) * 1e3, 2), "MiBps": round(n / min(ts) / MiB, 1)})
        log("w3_encode_blocks, pinned, pieces of %s blocks (%d pieces): %.1f ms per call = %.0f MiB/s" % (cb or "default", pieces, min(ts) * 1e3, n / min(ts) / MiB))
    ctx.set_host_chunk_blocks(0)

    # ---- the same from pageable memory (default piece size)
    if pageable:
        o = np.empty(cap, dtype=np.uint8); ln = np.empty(nb, dtype=np.uint32)
        sync_call(host.ctypes.data, o.ctypes.data, cap, ln.ctypes.data)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); tot = sync_call(host.ctypes.data, o.ctypes.data, cap, ln.ctypes.data); ts.append(time.perf_counter() - t0)
        assert (tot, bytes(o[: min(tot, 1 << 20)]) + bytes(ln[:nb].tobytes()[: 1 << 16])) == ref_bytes
        res["sync_pageable"] = {"ms_per_call": round(min(ts) * 1e3, 2), "MiBps": round(n / min(ts) / MiB, 1)}
        log("w3_encode_blocks, pageable: %.1f ms per call = %.0f MiB/s" % (min(ts) * 1e3, n / min(ts) / MiB))
        del o, ln

    # ---- calls in flight, pinned
    def run_host(k):
        pend = []
        last = None
        for i in range(k):
            if len(pend) == depth_h:
                last = ctx.encode_host_wait(pend.pop(0))
            o, ln = pin_outs[i % depth_h]
            pend.append(ctx.encode_host_submit(model, pin_in, bs, o, ln))
        for p in pend:
            last = ctx.encode_host_wait(p)
        return last
    run_host(depth_h + 1)
    t0 = time.perf_counter(); tot = run_host(calls); dt = (time.perf_counter() - t0) / calls
    o, ln = pin_outs[(calls - 1) % depth_h]
    assert (tot, bytes(o.numpy()[: min(tot, 1 << 20)]) + bytes(ln.numpy()[:nb].tobytes()[: 1 << 16])) == ref_bytes, "calls in flight: output differs from the synchronous call's"
    res["in_flight_pinned"] = {"ms_per_call": round(dt * 1e3, 2), "MiBps": round(n / dt / MiB, 1), "calls_in_flight": depth_h, "calls": calls}
    log("w3_encode_host_submit / wait, pinned, %d calls in flight: %.1f ms per call = %.0f MiB/s (PCIe in and out included)" % (depth_h, dt * 1e3, n / dt / MiB))
    res["compressed_ratio"] = round(tot / n, 4)
    ctx.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model", nargs="?", default="order012apm")
    ap.add_argument("bytes", nargs="?", type=float, default=1e9)
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--calls", type=int, default=8)
    ap.add_argument("--json", default="")
    ap.add_argument("--no-pageable", action="store_true")
    ap.add_argument("--no-resident", action="store_true")
    ap.add_argument("--tune", type=int, default=0, help="W3_OPT_TUNE (bit 16 = 65536: the copies on two streams of their own)")
    ap.add_argument("--churn", type=int, default=0, help="contexts created, used and closed before the measured one")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this run (0 = leave the environment alone)")
    a = ap.parse_args()
    if a.hw_queues:
        os.environ["GPU_MAX_HW_QUEUES"] = str(a.hw_queues)   # (before anything initialises the HIP runtime)
    res = measure(a.model, int(a.bytes), chunks=[int(c) for c in a.chunks.split(",")], calls=a.calls, pageable=not a.no_pageable, resident=not a.no_resident, tune=a.tune, churn=a.churn)
    txt = json.dumps(res)
    print(txt)
    if a.json:
        open(a.json, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
