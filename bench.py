#!/usr/bin/env python3
"""bench.py — encode throughput of the weath3rb0i hot path on MI355X.

A "step" is one pass of the hot path (predict + arithmetic-code + pack every
64 KiB block of the shard; for N>1 also the RCCL gather of the per-GPU streams
to rank 0) over one batch of synthetic enwik-shaped input already resident in
HBM.  One process per GPU; for N>1 launch with torch.distributed.run.

Prints ONE JSON line on rank 0 (contract in the task statement): metric /
value (whole-job MiB/s) / roofline (dominant kernel vs the HBM roof) /
cpu_baseline (the CPU oracle = C restatement of the reference, timed on the
host cores over a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_model(w3, name):
    if name == "order0":
        return w3.Order0(), "Order0"
    if name == "order01":
        return w3.BestOfTwoModel(w3.Order0(), w3.Order1()), "BestOfTwo(Order0,Order1)"
    if name == "order012":
        return (w3.BestOfTwoModel(w3.BestOfTwoModel(w3.Order0(), w3.Order1()), w3.OrderN(27, 3)),
                "BestOfTwo(BestOfTwo(Order0,Order1),OrderN(27,3))")
    if name == "default":
        return w3.init_model(), "OrderNEntropy(11,3,ACHistory(8,book1))"
    if name == "order012apm":  # BASELINE configs[1] with its "single APM mixer" (build-defined APM)
        return w3.APM(make_model(w3, "order012")[0]), "APM(" + make_model(w3, "order012")[1] + ",order0 ctx,rate 7)"
    if name == "fullcm":       # BASELINE configs[2]: Counter orders 0/1/2 + slot-state orders 1-4 + two APM stages
        return w3.full_cm(), "APM(APM(BestOfTwo^6(Order0,Order1,OrderN(27,3),Slot1..4(2^14 cells)),o0,7),o1,6)"
    raise SystemExit("unknown --model " + name)


def make_oracle_model(orc, name):
    if name == "order0":
        return orc.Order0()
    if name == "order01":
        return orc.BestOfTwoModel(orc.Order0(), orc.Order1())
    if name == "order012":
        return orc.BestOfTwoModel(orc.BestOfTwoModel(orc.Order0(), orc.Order1()), orc.OrderN(27, 3))
    if name == "order012apm":
        return orc.APM(make_oracle_model(orc, "order012"))
    if name == "fullcm":
        m = make_oracle_model(orc, "order012")
        for order in (1, 2, 3, 4):
            m = orc.BestOfTwoModel(m, orc.SlotModel(order, 14))
        return orc.APM(orc.APM(m, orc.APM_ORDER0, 7), orc.APM_ORDER1, 6)
    return orc.OrderNEntropy(11, 3, orc.ACHistory(8, orc.StationaryModel.for_book1()))


# per input byte, (Counter leaves, slot-state leaves, APM stages) of each bench model
MODEL_SHAPE = {"order0": (1, 0, 0), "order01": (2, 0, 0), "order012": (3, 0, 0), "default": (1, 0, 0), "order012apm": (3, 0, 1),
               "fullcm": (3, 4, 2)}


def cpu_baseline(name, sample, block_size, budget_s=15.0):
    """Oracle (kind="port": C restatement of the reference CPU path) on a bounded sample, all host cores."""
    from oracle import pyoracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("W3_CPU_THREADS", "16")))  # a 1-GPU box's CPU share is 16 cores
    m = make_oracle_model(orc, name)
    # calibrate in two stages (thread start-up and table allocation dominate tiny probes), then size for ~budget_s
    nbytes = min(len(sample), 4 * block_size * cores)
    for target in (2.0, budget_s):
        t0 = time.time()
        out, lens = orc.encode_blocks(m, sample[:nbytes], block_size, nthreads=cores)
        dt = max(time.time() - t0, 1e-3)
        if target == budget_s and dt >= 0.5 * budget_s:
            break
        nbytes = int(min(len(sample), max(nbytes, nbytes / dt * target))) // block_size * block_size
    t0 = time.time()
    out, lens = orc.encode_blocks(m, sample[:nbytes], block_size, nthreads=cores)
    dt = time.time() - t0
    return {"value": round(nbytes / dt / 2**20, 3), "unit": "MiB/s", "cores": cores, "kind": "port",
            "sample": "first %d bytes (%d blocks) of rank 0's shard, %d threads, %.1f s" % (nbytes, len(lens), cores, dt)}, out, lens, nbytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="order012apm", help="order0 | order01 | order012 | default | order012apm (BASELINE configs[1]) | fullcm (configs[2])")
    ap.add_argument("--size", type=int, default=1_000_000_000, help="input bytes PER GPU (enwik9-class = 1e9)")
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--path", default="auto", help="auto | generic | twophase")
    ap.add_argument("--parts", type=int, default=0, help="block ranges pipelined inside one encode call (W3_OPT_PARTS): 0 = auto, 1..4")
    ap.add_argument("--coder", default="x4", help="two-phase coder kernel: x4 (default) | x3 | x2 | fast | robust")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--force-exchange", action="store_true", help="run the RCCL exchange step even with 1 rank (rehearsal)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    exchange = world > 1 or args.force_exchange
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import weath3rb0i_amd as w3
    from tools import synth

    bs = args.block_size
    n = args.size
    nb = (n + bs - 1) // bs
    model, model_name = make_model(w3, args.model)
    ctx = w3.Context(local_rank)
    ctx.set_path(args.path)
    ctx.set_parts(args.parts)
    ctx.set_coder(args.coder)

    # rank r owns chunks [r*chunks, (r+1)*chunks) of one global seeded stream (weak scaling: n bytes per GPU)
    chunks_per_rank = (n + (1 << 20) - 1) >> 20
    host = synth.text(n, seed=args.seed, chunk0=rank * chunks_per_rank, nthreads=max(1, min(16, (os.cpu_count() or 8) // max(1, world))))
    d_in = torch.from_numpy(host).cuda()
    # output buffers are double-buffered: the exchange of step k (RCCL, its own stream) overlaps the encode of step k+1
    nbuf = 2 if exchange else 1
    d_outs = [torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    d_lenss = [torch.zeros(nb, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    d_out, d_lens = d_outs[0], d_lenss[0]
    stream = torch.cuda.current_stream().cuda_stream

    from weath3rb0i_amd import shard
    gather_buf = torch.empty(int(world * n * 0.75) + 4096, dtype=torch.uint8, device="cuda") if (exchange and rank == 0) else None
    gathered = {"pending": [], "k": 0}

    def step():
        k = gathered["k"] % nbuf
        gathered["k"] += 1
        ctx.encode_blocks_device(model, d_in, bs, d_outs[k], d_lenss[k], d_total, stream=stream)
        if exchange:
            # the one exchange step: sizes all-gather + grouped send/recv of the packed streams to rank 0 (RCCL).
            # The previous step's transfers must have landed before rank 0's gather buffer is reused.
            shard.wait_all(gathered["pending"])
            allb, alll, totals, reqs = shard.gather_streams(d_outs[k], int(d_total.item()), d_lenss[k], dst=0, out=gather_buf, async_op=True)
            gathered["pending"] = reqs
            gathered["bytes"] = sum(totals)

    def sync():
        if exchange:
            shard.wait_all(gathered["pending"])
            gathered["pending"] = []
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_timing(True)
    kern_ms = {"predict_ms": 0.0, "achash_ms": 0.0, "slot_ms": 0.0, "apm_ms": 0.0, "coder_ms": 0.0, "pack_ms": 0.0, "generic_ms": 0.0}
    coder_bytes = 0
    launches = 0
    parts = 1
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = ctx.timing()  # events were recorded on the launch stream; the encode call already synchronised it
        for k in kern_ms:
            kern_ms[k] += tm[k]
        coder_bytes += tm["coder_bytes"]
        launches += max(1, tm["n_coder_launches"])
        path = tm["path"]
        parts = max(1, tm["n_parts"])
    sync()
    dt = time.perf_counter() - t0
    ctx.set_timing(False)

    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    total_out = int(d_total.item())

    if rank == 0:
        ratio = total_out / n
        ms_per_step = dt / args.steps * 1e3
        value = world * n * args.steps / dt / 2**20
        # dominant kernel = the longest-running one of this model's two-phase kernels (or the fused lane-per-block kernel)
        traffic = None
        ncnt, nslot, napm = MODEL_SHAPE[args.model]
        if path == 2:
            # algorithmic HBM bytes per step: coder = streams + input + compressed bytes; k_apm0 = L streams +
            # input + its output stream; k_slot = per leaf and input byte 2 nibbles x (96 B read + 96 B written) + input + stream
            cands = [("k_coder_x3 (mix + recurrence + output wavefronts)", kern_ms["coder_ms"] / launches, coder_bytes / launches)]
            if napm == 1:
                cands.append(("k_apm0<%d> (APM stage: wave per block, table in LDS)" % (ncnt + nslot), kern_ms["apm_ms"] / (args.steps * parts),
                              n * (16 * (ncnt + nslot) + 1 + 16) / parts))
            if nslot:
                cands.append(("k_slot (slot-state leaves: lane per block, hash map in HBM; all batches of a step)",
                              kern_ms["slot_ms"] / args.steps, nslot * n * (1 + 16 + 2 * 192)))
            if kern_ms["achash_ms"] > 0:   # ACHistory leaves: one byte read + 8 key bytes written per input byte (+ the 8 MiB prefix table)
                cands.append(("k_achash (ACHistory keys of every step, 16-bit prefix table)", kern_ms["achash_ms"] / (args.steps * parts),
                              n * 9 / parts + (8 << 20)))
            dom_name, dom_ms, dom_bytes = max(cands, key=lambda c: c[1])
            try:  # PMC-measured HBM bytes of this kernel for this exact config (profiles/, separate rocprofv3 --pmc passes)
                for tj in json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))["entries"]:
                    if tj["config"] == {"model": args.model, "bytes_per_gpu": n, "block_size": bs} and dom_name.startswith(tj["kernel"]):
                        traffic = int(tj["traffic_bytes_per_step"] / parts)   # per launch, like `achieved`
            except (OSError, KeyError, ValueError):
                pass
        else:
            dom_ms = kern_ms["generic_ms"] / args.steps
            # A = 1 + c (stream write) + 64 B of Counter RMW per table model and input byte
            # + 384 B per slot-state leaf (2 nibbles x 96-B cell read + write) + 48 B per APM stage (8 x (4 B read + 2 B write))
            dom_bytes = n * (1 + ratio + 64 * ncnt + 384 * nslot + 48 * napm)
            dom_name = "k_cm" if (nslot or napm) else "k_generic"
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        res = {
            "metric": "encode MiB/s, 64 KiB blocks, bit-exact vs CPU ref",
            "value": round(value, 2), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "enwik9-shaped synthetic text (tools/synth.c seed %d), %d bytes per GPU, %d-byte blocks, model %s"
                       % (args.seed, n, bs, model_name), "bytes_per_gpu": n, "block_size": bs, "blocks_per_gpu": nb,
                       "model": model_name, "path": {1: "generic", 2: "twophase"}.get(path, str(path)), "compressed_ratio": round(ratio, 4),
                       "ranges_per_call": parts,
                       "exchange": "all_gather sizes + grouped send/recv to rank 0 (RCCL), overlapped with the next step's encode" if world > 1 else "none (1 GPU)"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": int(dom_bytes),
                         "launches_per_step": parts if path == 2 else 1},
            "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in kern_ms.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, cout, clens, cn = cpu_baseline(args.model, host, bs)
            # the baseline run doubles as a bit-exactness check of the timed GPU output
            nchk = len(clens)
            kl = (gathered["k"] - 1) % nbuf   # buffers of the last timed step
            g_lens = d_lenss[kl][:nchk].cpu().numpy().astype(np.uint32)
            g_out = d_outs[kl][: int(g_lens.sum())].cpu().numpy()
            cb["bit_exact_vs_gpu"] = bool(np.array_equal(g_lens, clens) and np.array_equal(g_out, cout))
            res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
