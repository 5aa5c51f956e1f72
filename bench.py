#!/usr/bin/env python3
"""bench.py — encode throughput of the weath3rb0i hot path on MI355X.

A "step" is one pass of the hot path (predict + arithmetic-code + pack every
block of the rank's shard; for N>1 also the RCCL gather of the per-GPU streams
to rank 0) over one batch of synthetic input already resident in HBM.  One
process per GPU; for N>1 launch with torch.distributed.run.

--scaling weak   (default) every GPU gets --size bytes of its own: N x 1e9 bytes in aggregate.
--scaling strong ONE global stream of --size bytes is cut into contiguous block ranges
                 (shard.byte_range), BASELINE.json configs[3]: "enwik9 sharded across 8 GPUs".
At N = 1 the two are the same run.

Prints ONE JSON line on rank 0 (contract in the task statement): metric /
value (whole-job MiB/s) / roofline (dominant kernel vs the HBM roof) /
cpu_baseline (the CPU oracle = C restatement of the reference, timed on the
host cores over a bounded sample of the same workload, two legs) / floors (the
block-count-independent coder chain that bounds strong scaling).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SHADER_GHZ = 2.4        # MI355X_MICROARCH.md: max clock; the lone coder waves run at it (profiles/r2_xstep_bench.txt)


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


def host_cores():
    """Host cores this process can really use: the scheduler affinity, cut to the cgroup's CPU quota when there is one
    (a 1-GPU box shows all 256 cores of the node in its affinity mask but is allotted 16: 256 oracle threads then ran
    SLOWER than one).  Returns (cores to use, what limited them)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    cores, why = aff, "sched_getaffinity"
    if quota is not None and quota < cores:
        cores, why = max(1, int(quota + 0.5)), "cgroup cpu quota %.1f of %d visible" % (quota, aff)
    elif aff > 64:
        cores, why = 16, "no cgroup quota readable; %d cores visible, a 1-GPU box is allotted 16" % aff
    if os.environ.get("W3_CPU_THREADS"):
        cores, why = max(1, min(aff, int(os.environ["W3_CPU_THREADS"]))), "W3_CPU_THREADS"
    return cores, why


def cpu_baseline(name, sample, block_size, budget_s=8.0):
    """Oracle (kind="port": C restatement of the reference CPU path) on a bounded sample of rank 0's shard, two legs:
    (i) ONE thread, the sample as ONE stream — the reference's actual mode (Cargo.toml:14-15, main.rs:89-113);
    (ii) block-parallel over every host core this process may use (one block per task, fresh model + coder each).
    `value` is leg (ii), the stronger baseline.  Returns (dict, leg-ii streams, lens, bytes covered)."""
    from oracle import pyoracle as orc
    cores, cores_why = host_cores()

    def sized(run, n0, unit):
        """calibrate in two stages (start-up and table allocation dominate tiny probes), then size for ~budget_s"""
        nbytes = max(unit, min(len(sample), n0) // unit * unit)
        for target in (1.0, budget_s):
            t0 = time.time()
            run(nbytes)
            dt = max(time.time() - t0, 1e-3)
            if target == budget_s and dt >= 0.5 * budget_s:
                return nbytes
            nbytes = max(unit, int(min(len(sample), max(nbytes, nbytes / dt * target))) // unit * unit)
        return nbytes

    # (i) single thread, whole stream (a fresh model per run: the oracle's models are stateful)
    one = lambda k: orc.encode_stream(make_oracle_model(orc, name), sample[:k].tobytes())
    n1 = sized(one, 1 << 20, 1 << 16)
    t0 = time.time()
    s1 = one(n1)
    dt1 = time.time() - t0
    # (ii) block-parallel
    par = lambda k: orc.encode_blocks(make_oracle_model(orc, name), sample[:k], block_size, nthreads=cores)
    n2 = sized(par, 4 * block_size * cores, block_size)
    t0 = time.time()
    out, lens = par(n2)
    dt2 = time.time() - t0
    legs = [{"mode": "single thread, whole sample as one stream (the reference's mode)", "value": round(n1 / dt1 / 2**20, 3), "unit": "MiB/s",
             "cores": 1, "sample_bytes": n1, "seconds": round(dt1, 2), "ns_per_bit": round(dt1 * 1e9 / (8 * n1), 2),
             "compressed_ratio": round(len(s1) / n1, 4)},
            {"mode": "block-parallel, one %d-byte block per task" % block_size, "value": round(n2 / dt2 / 2**20, 3), "unit": "MiB/s",
             "cores": cores, "sample_bytes": n2, "seconds": round(dt2, 2)}]
    return {"value": legs[1]["value"], "unit": "MiB/s", "cores": cores, "cores_chosen_by": cores_why, "kind": "port",
            "sample": "first %d bytes (%d blocks) of rank 0's shard, %d threads, %.1f s" % (n2, len(lens), cores, dt2),
            "legs": legs}, out, lens, n2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="order012apm", help="order0 | order01 | order012 | default | order012apm (BASELINE configs[1]) | fullcm (configs[2])")
    ap.add_argument("--size", type=int, default=1_000_000_000, help="input bytes: per GPU (weak) or in all (strong); enwik9-class = 1e9")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--data", default="text", choices=["text", "mixed"], help="text = enwik-shaped, mixed = Silesia-shaped (BASELINE configs[4] with --block-size 262144 --size 211938580)")
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--path", default="auto", help="auto | generic | twophase")
    ap.add_argument("--parts", type=int, default=0, help="EXPERIMENTAL block ranges pipelined inside one encode call (W3_OPT_PARTS): 0 = auto, 1..4")
    ap.add_argument("--coder", default="x4", help="two-phase coder kernel: x4 (default) | x3 | x2 | fast | robust")
    ap.add_argument("--variant", default="", help="experiments: comma-separated W3_OPT_VARIANT names (Context.set_variant), e.g. no_side_stream")
    ap.add_argument("--no-verify", action="store_true", help="switch W3_OPT_VERIFY off (sampled ballot-round re-prediction of every predict phase; on by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-model", action="store_true", help="skip the extra order012 measurement (the largest model whose streams are entirely the reference's)")
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
    from weath3rb0i_amd import shard

    bs = args.block_size
    gen = synth.text if args.data == "text" else synth.mixed
    nthreads = max(1, min(16, (os.cpu_count() or 8) // max(1, world)))
    if args.scaling == "strong":
        # one global stream of --size bytes; rank r codes the contiguous block range shard.byte_range gives it.
        # The generator works in 1 MiB chunks from a chunk index: generate the covering chunks, slice the range out.
        n_global = args.size
        lo, hi = shard.byte_range(rank, world, n_global, bs)
        c0 = lo >> 20
        host = gen(max(hi - (c0 << 20), 1), seed=args.seed, chunk0=c0, nthreads=nthreads)[lo - (c0 << 20): hi - (c0 << 20)]
        n = hi - lo
    else:
        # rank r owns chunks [r*chunks, (r+1)*chunks) of one global seeded stream (weak scaling: n bytes per GPU)
        n = args.size
        n_global = n * world
        chunks_per_rank = (n + (1 << 20) - 1) >> 20
        host = gen(n, seed=args.seed, chunk0=rank * chunks_per_rank, nthreads=nthreads)
    nb = (n + bs - 1) // bs
    model, model_name = make_model(w3, args.model)
    ctx = w3.Context(local_rank)
    ctx.set_path(args.path)
    ctx.set_parts(args.parts)
    ctx.set_coder(args.coder)
    ctx.set_verify(not args.no_verify)
    if args.variant:
        ctx.set_variant(*args.variant.split(","))

    d_in = torch.from_numpy(np.ascontiguousarray(host)).cuda()
    # output buffers are double-buffered: the exchange of step k (RCCL, its own stream) overlaps the encode of step k+1
    nbuf = 2 if exchange else 1
    d_outs = [torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    d_lenss = [torch.zeros(max(nb, 1), dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    gather_buf = torch.empty(int(n_global * 0.75) + 4096, dtype=torch.uint8, device="cuda") if (exchange and rank == 0) else None
    gathered = {"pending": [], "k": 0}

    def step(mdl=None):
        k = gathered["k"] % nbuf
        gathered["k"] += 1
        ctx.encode_blocks_device(mdl or model, d_in, bs, d_outs[k], d_lenss[k], d_total, stream=stream)
        if exchange:
            # the one exchange step: sizes all-gather + grouped send/recv of the packed streams to rank 0 (RCCL).
            # The previous step's transfers must have landed before rank 0's gather buffer is reused.
            shard.wait_all(gathered["pending"])
            allb, alll, totals, reqs = shard.gather_streams(d_outs[k], int(d_total.item()), d_lenss[k][:nb], dst=0, out=gather_buf, async_op=True)
            gathered["pending"] = reqs
            gathered["bytes"] = sum(totals)

    def sync():
        if exchange:
            shard.wait_all(gathered["pending"])
            gathered["pending"] = []
            dist.barrier()
        torch.cuda.synchronize()

    def timed(mdl, steps):
        ctx.set_timing(True)
        kern = {"predict_ms": 0.0, "achash_ms": 0.0, "slot_ms": 0.0, "apm_ms": 0.0, "coder_ms": 0.0, "pack_ms": 0.0, "generic_ms": 0.0}
        extra = {"coder_bytes": 0, "predict_bytes": 0, "launches": 0, "parts": 1, "path": 0}
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(mdl)
            tm = ctx.timing()  # events were recorded on the launch stream; the encode call already synchronised it
            for k in kern:
                kern[k] += tm[k]
            extra["coder_bytes"] += tm["coder_bytes"]
            extra["predict_bytes"] += tm["predict_bytes"]
            extra["launches"] += max(1, tm["n_coder_launches"])
            extra["path"] = tm["path"]
            extra["parts"] = max(1, tm["n_parts"])
        sync()
        dt = time.perf_counter() - t0
        ctx.set_timing(False)
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), kern, extra

    for _ in range(args.warmup):
        step()
    dt, kern_ms, ex = timed(model, args.steps)
    coder_bytes, launches, parts, path = ex["coder_bytes"], ex["launches"], ex["parts"], ex["path"]
    total_out = int(d_total.item())
    last_buf = (gathered["k"] - 1) % nbuf   # buffers of the last timed step

    ref_model = None
    g_lens_keep = g_out_keep = None
    if world == 1 and not exchange and not args.no_ref_model and args.model != "order012" and args.data == "text":
        # the largest model whose streams are entirely the reference's (no build-defined node): same input, 3 steps, outside the
        # timed region (the main model's output is kept aside for the bit-exactness check first)
        if not args.no_cpu_baseline:
            g_lens_keep = d_lenss[last_buf].clone()
            g_out_keep = d_outs[last_buf][:total_out].clone()
        m2, m2_name = make_model(w3, "order012")
        step(m2)
        dt2, k2, _ = timed(m2, 3)
        ref_model = {"model": m2_name, "value": round(n * 3 / dt2 / 2**20, 2), "unit": "MiB/s", "ms_per_step": round(dt2 / 3 * 1e3, 3),
                     "kernel_ms_per_step": {k: round(v / 3, 3) for k, v in k2.items()}, "compressed_ratio": round(int(d_total.item()) / n, 4),
                     "note": "every node of this model is the reference's (Order0/Order1/OrderN + OpinionMixer2): its block streams are the reference's streams"}

    if rank == 0:
        ratio = total_out / max(n, 1)
        ms_per_step = dt / args.steps * 1e3
        value = n_global * args.steps / dt / 2**20
        # dominant kernel = the longest-running one of this model's two-phase kernels (or the fused lane-per-block kernel)
        traffic = traffic_src = None
        ncnt, nslot, napm = MODEL_SHAPE[args.model]
        if path == 2:
            # algorithmic HBM bytes per step: coder = streams + input + compressed bytes; k_apm0 = L streams +
            # input + its output stream; k_slot = per leaf and input byte 2 nibbles x (96 B read + 96 B written) + input + stream
            cands = [("k_coder_%s (mix + recurrence + output wavefronts)" % args.coder, kern_ms["coder_ms"] / launches, coder_bytes / launches)]
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
            # PMC-measured HBM bytes of this kernel for this exact config: NOT measured in this run — separate rocprofv3 --pmc passes
            # (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE), kept under profiles/ and looked up here
            for fn in ("r2_traffic.json", "r1_traffic.json"):
                try:
                    for tj in json.load(open(os.path.join(ROOT, "profiles", fn)))["entries"]:
                        if tj["config"] == {"model": args.model, "bytes_per_gpu": n, "block_size": bs} and dom_name.startswith(tj["kernel"]) and traffic is None:
                            traffic = int(tj["traffic_bytes_per_step"] / parts)   # per launch, like `achieved`
                            traffic_src = "profiles/" + fn + " (rocprofv3 --pmc passes of this config; a constant looked up, not measured in this run)"
                except (OSError, KeyError, ValueError):
                    pass
        else:
            dom_ms = kern_ms["generic_ms"] / args.steps
            # A = 1 + c (stream write) + 64 B of Counter RMW per table model and input byte
            # + 384 B per slot-state leaf (2 nibbles x 96-B cell read + write) + 48 B per APM stage (8 x (4 B read + 2 B write))
            dom_bytes = n * (1 + ratio + 64 * ncnt + 384 * nslot + 48 * napm)
            dom_name = "k_cm" if (nslot or napm) else "k_generic"
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        coder_ms = kern_ms["coder_ms"] / args.steps
        steps_per_lane = 8 * min(bs, max(n, 1))
        workload = ("%s synthetic (tools/synth.c seed %d), %s scaling: %s, %d-byte blocks, model %s"
                    % ("enwik9-shaped text" if args.data == "text" else "Silesia-shaped mix", args.seed, args.scaling,
                       ("%d bytes per GPU (%d in all)" % (n, n_global)) if args.scaling == "weak" else ("ONE stream of %d bytes cut over %d GPU(s)" % (n_global, world)),
                       bs, model_name))
        # w3_timing.predict_bytes also carries the APM stages' bytes: take the single ORDER0 stage's out again; models with slot
        # leaves or several stages get no predict-phase figure
        predict_phase = None
        if path == 2 and kern_ms["predict_ms"] > 0 and nslot == 0 and napm <= 1:
            pb = ex["predict_bytes"] / args.steps - (n * (16 * ncnt + 17) if napm == 1 else 0)
            pms = kern_ms["predict_ms"] / args.steps
            predict_phase = {"algorithmic_bytes_per_step": int(pb), "achieved_GBps": round(pb / pms / 1e6, 1),
                             "frac_of_hbm_peak": round(pb / pms / 1e6 / HBM_PEAK_GBPS, 4)}
        res = {
            "metric": "encode MiB/s, %d KiB blocks, bit-exact vs CPU ref" % (bs >> 10),
            "value": round(value, 2), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": n, "bytes_total": n_global, "block_size": bs, "blocks_per_gpu": nb,
                       "context_model": model_name, "path": {1: "generic", 2: "twophase"}.get(path, str(path)), "compressed_ratio": round(ratio, 4),
                       "ranges_per_call": parts,
                       "exchange": "all_gather sizes + grouped send/recv to rank 0 (RCCL), overlapped with the next step's encode" if exchange else "none (1 GPU)"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": int(dom_bytes),
                         "launches_per_step": parts if path == 2 else 1},
            "kernel_ms_per_step": {k: round(v / args.steps, 3) for k, v in kern_ms.items()},
            # the predict phase is several kernels on two streams (their launch durations overlap); as a whole: algorithmic bytes
            # (per leaf input + 16-byte stream, plus 8 B written + 8 B read per record pass of a wide leaf) over the phase's time
            "predict_phase": predict_phase,
            # the coder is ONE dependent chain of 8 x block_size bit-steps per lane: its time does not shrink with the block count,
            # so it is the floor of a strong-scaled run (predict / APM / pack scale with the bytes per GPU)
            "floors": {"coder_floor_ms": round(coder_ms, 3) if path == 2 else None, "bit_steps_per_lane": steps_per_lane,
                       "cycles_per_bit_step": round(coder_ms * 1e-3 * SHADER_GHZ * 1e9 / steps_per_lane, 1) if path == 2 else None,
                       "clock_ghz_assumed": SHADER_GHZ,
                       "strong_scaling_projection_ms": ({str(g): round(coder_ms + (ms_per_step - coder_ms) / g, 2) for g in (1, 2, 4, 8)}
                                                        if (world == 1 and path == 2) else None),
                       "note": "projection for ONE stream of bytes_total cut over g GPUs = coder_floor + (ms_per_step - coder_floor) / g, gather overlapped; not a measurement"},
        }
        if ref_model:
            res["reference_stream_model"] = ref_model
        if world == 1 and not args.no_cpu_baseline:
            cb, cout, clens, cn = cpu_baseline(args.model, host, bs)
            # the baseline run doubles as a bit-exactness check of the timed GPU output
            nchk = len(clens)
            src_lens = g_lens_keep if g_lens_keep is not None else d_lenss[last_buf]
            g_lens = src_lens[:nchk].cpu().numpy().astype(np.uint32)
            src_out = g_out_keep if g_out_keep is not None else d_outs[last_buf]
            g_out = src_out[: int(g_lens.sum())].cpu().numpy()
            cb["bit_exact_vs_gpu"] = bool(np.array_equal(g_lens, clens) and np.array_equal(g_out, cout))
            res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
