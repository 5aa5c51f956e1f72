#!/usr/bin/env python3
"""bench.py — encode throughput of the weath3rb0i hot path on MI355X.

A "step" is one pass of the hot path (predict + APM + arithmetic-code + pack every block of the rank's shard; for N>1 also the
RCCL gather of the per-GPU streams to rank 0) over one batch of synthetic input already resident in HBM.  One process per GPU.

    python3 bench.py --gpus N --steps K --warmup W

N > 1 with WORLD_SIZE unset: this process starts the N ranks itself (`python -m torch.distributed.run`, as a child process,
before anything here touches the GPU), relays rank 0's JSON line and exits with the child's code.  Under a launcher
(WORLD_SIZE set) it is one rank.

Steps are PIPELINED (--pipeline 2, default): w3_encode_submit / w3_encode_wait keep two encodes in flight, so step k+1's predict
phase runs beside step k's APM and coder kernels; the timed region holds exactly K steps, bracketed by barrier + synchronize
(nothing of the warm-up is still running when it starts, everything of step K has landed when it ends).  --pipeline 1: one
synchronous call per step.

--scaling both (default)  value = the STRONG reading: ONE stream of --size bytes cut into contiguous block ranges over the N GPUs
                          (BASELINE.json configs[3]: "enwik9 sharded across 8"); for N > 1 the WEAK reading (--size bytes per GPU)
                          is timed too and attached as "weak".  At N = 1 the two are the same run.
--scaling strong | weak   that reading only.

Prints ONE JSON line on rank 0 (contract in the task statement): metric / value (whole-job MiB/s) / roofline (the kernel with
the longest launch vs the HBM roof, every kernel's figure in roofline_kernels) / cpu_baseline (the CPU oracle = C restatement of
the reference, -O3 -march=native, timed on the host cores over a bounded sample, two legs) / decode (device round trip of EVERY
block of the last step's output) / other_configs (BASELINE configs[2] and configs[4] shapes, 3 steps each) / floors.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The library asks nothing of the environment (its pipeline stages use streams of different priority levels, which HIP maps to
# different hardware queues: INTEGRATION.md).  --hw-queues N sets GPU_MAX_HW_QUEUES=N for THIS run (it must be in the environment before
# the HIP runtime initialises, hence the early look at argv); whatever is in effect is recorded in config.gpu_max_hw_queues.
for _i, _a in enumerate(sys.argv):
    if _a == "--hw-queues" and _i + 1 < len(sys.argv) and int(sys.argv[_i + 1]) > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[_i + 1]
    elif _a.startswith("--hw-queues=") and int(_a.split("=")[1]) > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = _a.split("=")[1]

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
NSETS = 4                # output buffer sets of a regime = the deepest submit / wait pipeline (W3_MAX_JOBS)
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
    # the reference's best published configurations (wave-per-block predict kernel, w3_predict_wave.h)
    if name == "ac26":         # bin/entropy-hashing-ac/main.rs:21-25, enwik7.log:115: best ratio on enwik7
        return w3.OrderNEntropy(26, 3, w3.ACHistory(23, w3.StationaryModel.for_enwik7())), "OrderNEntropy(26,3,ACHistory(23,enwik7))"
    if name == "ac20":         # book1.log:139
        return w3.OrderNEntropy(20, 3, w3.ACHistory(17, w3.StationaryModel.for_book1())), "OrderNEntropy(20,3,ACHistory(17,book1))"
    if name == "ordern32_1":   # bin/ordern/enwik7.log:163: best plain OrderN
        return w3.OrderN(32, 1), "OrderN(32,1)"
    if name == "ordern22_2":
        return w3.OrderN(22, 2), "OrderN(22,2)"
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
    if name == "ac26":
        return orc.OrderNEntropy(26, 3, orc.ACHistory(23, orc.StationaryModel.for_enwik7()))
    if name == "ac20":
        return orc.OrderNEntropy(20, 3, orc.ACHistory(17, orc.StationaryModel.for_book1()))
    if name == "ordern32_1":
        return orc.OrderN(32, 1)
    if name == "ordern22_2":
        return orc.OrderN(22, 2)
    return orc.OrderNEntropy(11, 3, orc.ACHistory(8, orc.StationaryModel.for_book1()))


WAVE_MODELS = {"ac26": 65, "ac20": 65, "ordern32_1": 0, "ordern22_2": 0}   # key bytes per input byte (1 read + 32 written + 32 read) of the hash kernel

# per input byte, (time-ordered Counter leaves, wide Counter leaves, slot-state leaves, APM stages) of each bench model
MODEL_SHAPE = {"ac26": (1, 0, 0, 0), "ac20": (1, 0, 0, 0), "ordern32_1": (1, 0, 0, 0), "ordern22_2": (1, 0, 0, 0), "order0": (1, 0, 0, 0), "order01": (1, 1, 0, 0), "order012": (1, 2, 0, 0), "default": (1, 0, 0, 0), "order012apm": (1, 2, 0, 1),
               "fullcm": (1, 2, 4, 2)}


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
    """Oracle (kind="port": C restatement of the reference CPU path, built -O3 -march=native on this host) on a bounded sample of
    rank 0's shard, two legs:
    (i) ONE thread, the sample as ONE stream — the reference's actual mode (Cargo.toml:14-15, main.rs:89-113);
    (ii) block-parallel over every host core this process may use (one block per task, fresh model + coder each).
    `value` is leg (ii), the stronger baseline.  Returns (dict, leg-ii streams, lens, bytes covered)."""
    from oracle import native
    so = native.build_native()   # before pyoracle's first import: it loads W3_ORACLE_SO
    if so:
        os.environ["W3_ORACLE_SO"] = so
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
            "build": "gcc -O3 -march=native on this host" if so else "gcc -O3 (portable build; the native one could not be made here)",
            "sample": "first %d bytes (%d blocks) of rank 0's shard, %d threads, %.1f s" % (n2, len(lens), cores, dt2),
            "legs": legs}, out, lens, n2


def launcher_command(n_gpus, argv, port):
    """The command that starts the N ranks of this bench (what the driver runs itself for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_gpus, argv):
    """WORLD_SIZE unset and --gpus N > 1: start the ranks as a CHILD process (never exec: this process must not touch the GPU, and a
    GPU process must not be replaced) and relay their output."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(launcher_command(n_gpus, argv, free_port()), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in p.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if p.returncode != 0 or len(lines) != 1:
        print("bench.py: the %d-rank run failed (exit code %d, %d JSON lines)" % (n_gpus, p.returncode, len(lines)), file=sys.stderr)
        raise SystemExit(p.returncode or 1)
    print(lines[0], flush=True)
    raise SystemExit(0)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="order012apm", help="order0 | order01 | order012 | default | order012apm (BASELINE configs[1]) | fullcm (configs[2]) | ac26 | ac20 | ordern32_1 | ordern22_2 (the reference's best published configurations)")
    ap.add_argument("--size", type=int, default=1_000_000_000, help="input bytes: of the one stream (strong) / per GPU (weak); enwik9-class = 1e9")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"])
    ap.add_argument("--data", default="text", choices=["text", "mixed"], help="text = enwik-shaped, mixed = Silesia-shaped (BASELINE configs[4] with --block-size 262144 --size 211938580)")
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--path", default="auto", help="auto | generic | twophase")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2, 3, 4], help="encodes in flight (w3_encode_submit / w3_encode_wait): 0 = as many as w3_encode_max_in_flight allows for the shard "
                    "(default: 4 up to 4,096 blocks, 3 up to 12,288, 2 beyond), 1 = one synchronous call per step")
    ap.add_argument("--coder", default="x4", help="two-phase coder kernel: x4 (default; pipelined and half-CU runs use x5 in its place) | x5 | x3 | x2 | fast | robust")
    ap.add_argument("--variant", default="", help="experiments: comma-separated W3_OPT_VARIANT names (Context.set_variant), e.g. no_side_stream, half_cu, full_cu")
    ap.add_argument("--tune", type=int, default=0, help="W3_OPT_TUNE bit mask (scheduling experiments)")
    ap.add_argument("--verify", type=int, default=1, help="W3_OPT_VERIFY: v / 256 of the blocks are re-predicted with ballot rounds per call (1 = default sample)")
    ap.add_argument("--no-verify", action="store_true", help="switch W3_OPT_VERIFY off (sampled ballot-round re-prediction of every predict phase; on by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-model", action="store_true", help="skip the extra order012 measurement (the largest model whose streams are entirely the reference's)")
    ap.add_argument("--no-decode", action="store_true", help="skip the device round trip of the last step's whole output")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 3-step runs of BASELINE configs[2] / configs[4]")
    ap.add_argument("--quick", action="store_true", help="= --no-cpu-baseline --no-ref-model --no-decode --no-other-configs --no-host-path (profiling runs)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this run (0 = leave the environment alone: HIP's default of 4 per priority level)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the PCIe-inclusive legs (host_path: w3_encode_blocks / w3_encode_host_submit from pinned and pageable memory)")
    ap.add_argument("--force-exchange", action="store_true", help="run the RCCL exchange step even with 1 rank (rehearsal)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI, one GPU per rank (the measured path).  gloo = REHEARSAL of the multi-rank control flow on a box with "
                         "fewer GPUs than ranks: the ranks share the GPUs there are (rank r on device r mod count) and the exchange is staged through host "
                         "memory; its numbers mean nothing")
    a = ap.parse_args(argv)
    if a.quick:
        a.no_cpu_baseline = a.no_ref_model = a.no_decode = a.no_other_configs = a.no_host_path = True
    return a


class Regime:
    """One reading (strong or weak) on this rank: the shard, its device buffers and the step loop."""

    def __init__(self, env, scaling, size, data_kind, block_size, seed):
        import numpy as np
        import torch
        from tools import synth
        from weath3rb0i_amd import shard
        self.env, self.scaling, self.bs = env, scaling, block_size
        rank, world = env["rank"], env["world"]
        gen = synth.text if data_kind == "text" else synth.mixed
        nthreads = max(1, min(16, (os.cpu_count() or 8) // max(1, world)))
        if scaling == "strong":
            # one global stream of --size bytes; rank r codes the contiguous block range shard.byte_range gives it.
            # The generator works in 1 MiB chunks from a chunk index: generate the covering chunks, slice the range out.
            self.n_global = size
            lo, hi = shard.byte_range(rank, world, size, block_size)
            c0 = lo >> 20
            self.host = gen(max(hi - (c0 << 20), 1), seed=seed, chunk0=c0, nthreads=nthreads)[lo - (c0 << 20): hi - (c0 << 20)]
            self.n = hi - lo
        else:
            # rank r owns chunks [r*chunks, (r+1)*chunks) of one global seeded stream (weak scaling: n bytes per GPU)
            self.n = size
            self.n_global = size * world
            chunks_per_rank = (size + (1 << 20) - 1) >> 20
            self.host = gen(size, seed=seed, chunk0=rank * chunks_per_rank, nthreads=nthreads)
        n = self.n
        self.nb = (n + block_size - 1) // block_size
        self.d_in = torch.from_numpy(np.ascontiguousarray(self.host)).cuda()
        # one output buffer set per encode in flight (2 .. 4); the exchange of step k (RCCL, its own stream) overlaps step k+1
        self.nsets = NSETS
        self.d_outs = [torch.empty(n + n // 4 + 64 * self.nb + 1024, dtype=torch.uint8, device="cuda") for _ in range(NSETS)]
        self.d_lenss = [torch.zeros(max(self.nb, 1), dtype=torch.int32, device="cuda") for _ in range(NSETS)]
        self.d_totals = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(NSETS)]
        self.gather_buf = (torch.empty(int(self.n_global * 0.75) + 4096, dtype=torch.uint8, device="cuda")
                           if (env["exchange"] and rank == 0 and not env.get("host_staged")) else None)
        self.reqs = [[] for _ in range(NSETS)]   # outstanding exchange requests per buffer set
        self.gathered = None         # (rank totals) of the last exchange
        self.last_buf = 0
        self.stream = torch.cuda.current_stream().cuda_stream

    def release(self):
        import torch
        self.d_in = self.d_outs = self.d_lenss = self.d_totals = self.gather_buf = None
        torch.cuda.empty_cache()

    def _exchange(self, k):
        """the one exchange step: sizes all-gather + grouped send/recv of the packed streams to rank 0 (RCCL), asynchronous"""
        from weath3rb0i_amd import shard
        other = (k - 1) % self.nsets
        shard.wait_all(self.reqs[other])   # rank 0's gather buffer is reused: the previous step's transfers must have landed
        self.reqs[other] = []
        total = int(self.d_totals[k].item())
        if self.env.get("host_staged"):   # gloo rehearsal: the same exchange on host copies
            _, _, totals, reqs = shard.gather_streams(self.d_outs[k][:total].cpu(), total, self.d_lenss[k][:self.nb].cpu(), dst=0, async_op=True)
        else:
            _, _, totals, reqs = shard.gather_streams(self.d_outs[k], total, self.d_lenss[k][:self.nb], dst=0, out=self.gather_buf, async_op=True)
        self.reqs[k] = reqs
        self.gathered = totals

    def run_steps(self, ctx, model, steps, pipeline, acc=None):
        """`steps` encodes of this shard (+ exchange), at most `pipeline` in flight; acc: dict that sums w3_timing fields"""
        from weath3rb0i_amd import shard
        pending = []
        pipeline = min(pipeline, max(ctx.max_in_flight(self.n, self.bs, model), self.env.get("force_depth", 0))) if pipeline > 1 else pipeline
        for i in range(steps):
            k = i % self.nsets
            if self.reqs[k]:
                shard.wait_all(self.reqs[k])   # buffer set k is about to be overwritten
                self.reqs[k] = []
            if pipeline == 1:
                ctx.encode_blocks_device(model, self.d_in, self.bs, self.d_outs[k], self.d_lenss[k], self.d_totals[k], stream=self.stream)
                self._done(ctx, k, acc)
            else:
                if len(pending) >= pipeline:   # the oldest job's slot (and, with NSETS == the deepest pipeline, its buffer set) is needed
                    job0, k0 = pending.pop(0)
                    ctx.encode_wait(job0)
                    self._done(ctx, k0, acc)
                pending.append((ctx.encode_submit(model, self.d_in, self.bs, self.d_outs[k], self.d_lenss[k], self.d_totals[k], stream=self.stream), k))
        for job0, k0 in pending:
            ctx.encode_wait(job0)
            self._done(ctx, k0, acc)

    def _done(self, ctx, k, acc):
        self.last_buf = k
        if acc is not None:
            tm = ctx.timing()
            for key, v in tm.items():
                if isinstance(v, list):
                    acc[key] = [a + b for a, b in zip(acc.get(key, [0.0] * len(v)), v)]
                elif key in ("path", "n_wide", "n_parts"):
                    acc[key] = v
                else:
                    acc[key] = acc.get(key, 0) + v
        if self.env["exchange"]:
            self._exchange(k)

    def sync(self):
        import torch
        import torch.distributed as dist
        from weath3rb0i_amd import shard
        if self.env["exchange"]:
            for k in range(self.nsets):
                shard.wait_all(self.reqs[k])
                self.reqs[k] = []
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, ctx, model, steps, warmup, pipeline):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; MAX over ranks.  -> (seconds, timing sums)"""
        import torch
        import torch.distributed as dist
        if pipeline > 1 and not getattr(self, "_primed", None) == (id(ctx), pipeline):
            # every job slot of the submit / wait pipeline allocates its workspace at its first use: one untimed round over all of them
            # first, so that W = 1 warm-up step does not leave allocations of the other slots inside the timed region
            self.run_steps(ctx, model, pipeline, pipeline)
            self._primed = (id(ctx), pipeline)
        if warmup:
            self.run_steps(ctx, model, warmup, pipeline)
        ctx.set_timing(True)
        acc = {}
        self.sync()
        t0 = time.perf_counter()
        self.run_steps(ctx, model, steps, pipeline, acc)
        self.sync()
        dt = time.perf_counter() - t0
        ctx.set_timing(False)
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if self.env.get("host_staged") else "cuda")
        if self.env["world"] > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), acc


def kernel_table(model, n, bs, steps, acc, coder_name):
    """Every two-phase kernel of the model: (name, average launch ms, ALGORITHMIC bytes per launch) — DESIGN.md §4.
    Launch durations come from hipEvents on each kernel's own launch stream (w3_timing); kernels of different streams (or of
    the other encode in flight) overlap, so the durations add up to more than the step."""
    nsmall, nwide, nslot, napm = MODEL_SHAPE[model]
    rows = []
    L = nsmall + nwide + nslot
    per = lambda key: acc.get(key, 0.0) / steps
    rows.append(("k_coder_%s (mix + recurrence + output wavefronts)" % coder_name, per("coder_ms") / max(1, acc.get("n_coder_launches", steps) / steps),
                 acc.get("coder_bytes", 0) / max(1, acc.get("n_coder_launches", steps))))
    if napm >= 1:
        rows.append(("k_apm0<%d> (APM stage: two wavefronts per block, table in LDS)%s" % (L, "" if napm == 1 else " + k_apm1 (ORDER1 stage) + k_mix"),
                     per("apm_ms"), n * (16 * L + 1 + 16) + (napm - 1) * n * (8 + 16 + 16)))
    if nslot and acc.get("slot_ms", 0) > 0:
        if acc.get("n_slot_launches", 0) > 0:
            # the table walk: per leaf and input byte 1 (input) + 16 (stream) + 2 nibbles x (96 B of Cell read + 96 B written) — SURVEY section 8(d)
            rows.append(("k_slot (slot-state leaves: lane per block, hash map in HBM; all batches of a step)", per("slot_ms"), nslot * n * (1 + 16 + 2 * 192)))
        else:
            # the sorted replay (w3_slot2.h; NO table in HBM): per leaf and input byte 1 (input) + 2 nibble events x (8 B record written by
            # k_slot_events, two 8-bit sort passes of 8 read + 8 written each (2^14 Cells), 8 read and 8 written by k_slot_replay)
            rows.append(("k_slot_events + k_slot_sort<0|1> + k_slot_replay (slot-state leaves: events sorted by Cell, replayed with the open Cell in LDS; no table in HBM)",
                         per("slot_ms"), nslot * n * (1 + 2 * (8 + 32 + 8 + 8))))
    if acc.get("achash_ms", 0) > 0:   # ACHistory leaves: one byte read + 8 key bytes written per input byte (+ the 8 MiB prefix table)
        rows.append(("k_achash (ACHistory keys of every step, 16-bit prefix table)", per("achash_ms"), n * 9 + (8 << 20)))
    for w in range(int(acc.get("n_wide", 0))):
        # wide leaf w: leaf order = Order1-shaped first in every bench model; the order-2 leaf refines the Order1 leaf's records
        o2 = w >= 1
        rows.append(("k_rank_sorted<%d> (Counter rounds inside sorted groups, 16-byte scatter)" % (2 if o2 else 1), acc["rank_ms"][w] / steps, n * (8 + 16)))
        rows.append(("k_partition8<%d> (stable 8-bit partition through LDS tiles)" % (3 if o2 else 1), acc["part_ms"][w] / steps, n * (1 + (8 if o2 else 1) + 8)))
    if model in WAVE_MODELS:   # one wavefront per block, 64 steps per round, Counter table in HBM: per step 8 input bytes read (the lane's
        # window), 2 written, one 8-byte slot read and written; + the 32-bit hashes of a hashed history
        rows.append(("k_predict_wave (+ k_achash32: ACHistory hashes of every step)" if WAVE_MODELS[model] else "k_predict_wave (wave per block, Counter table in HBM)",
                     per("predict_ms"), n * (8 * (8 + 2 + 16) + WAVE_MODELS[model])))
    if acc.get("small_ms", 0) > 0 and nsmall:
        rows.append(("k_predict_small<8> (time-ordered Counter leaf, table in LDS)", per("small_ms") / nsmall, n * 17))
    rows.append(("k_scan_lens + k_pack (wave prefix scan + compaction copy)", per("pack_ms"), 2 * (acc.get("coder_bytes", 0) / steps - n * (16 * (1 if napm else L) + 1))))
    return [r for r in rows if r[1] > 0]


def lookup_traffic(model, n, bs, kernel_name):
    """PMC-measured HBM bytes per launch of this kernel for this exact config: NOT measured in this run — separate rocprofv3 --pmc
    passes (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE), kept under profiles/ and looked up here."""
    for fn in ("r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
        try:
            for tj in json.load(open(os.path.join(ROOT, "profiles", fn)))["entries"]:
                if tj["config"] == {"model": model, "bytes_per_gpu": n, "block_size": bs} and kernel_name.startswith(tj["kernel"]):
                    return int(tj["traffic_bytes_per_step"]), "profiles/" + fn + " (rocprofv3 --pmc passes of this config; a constant looked up, not measured in this run)"
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def short_run(w3, model_name, data_kind, size, bs, seed, steps, env, ctx=None):
    """short line of another BASELINE configuration (a fresh context after the main run has released its memory, or the main run's own)"""
    import torch
    rg = Regime(env, "weak", size, data_kind, bs, seed)
    own_ctx = ctx is None
    if own_ctx:
        ctx = w3.Context(env["local_rank"])
    try:
        model, mname = make_model(w3, model_name)
        pipeline = ctx.max_in_flight(rg.n, bs, model)   # (2 for the full CM: pipelined below 7,000 blocks — the sorted replay —, synchronous inside submit beyond)
        dt, acc = rg.timed(ctx, model, steps, 1, pipeline)
        rows = kernel_table(model_name, rg.n, bs, steps, acc, "x4" if pipeline == 1 else "x5")
        dom = max(rows, key=lambda r: r[1])
        res = {"context_model": mname, "data": "enwik9-shaped text" if data_kind == "text" else "Silesia-shaped mix", "bytes": rg.n, "block_size": bs,
               "blocks": rg.nb, "steps": steps, "pipeline": pipeline, "value": round(rg.n * steps / dt / 2**20, 2), "unit": "MiB/s",
               "ms_per_step": round(dt / steps * 1e3, 3), "compressed_ratio": round(int(rg.d_totals[rg.last_buf].item()) / max(rg.n, 1), 4),
               "dominant_kernel": dom[0], "dominant_ms": round(dom[1], 3), "dominant_frac_of_hbm_peak": round(dom[2] / (dom[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
               "kernel_ms_per_step": {k: round(acc.get(k, 0.0) / steps, 3) for k in ("predict_ms", "slot_ms", "apm_ms", "coder_ms", "pack_ms")}}
    finally:
        if own_ctx:
            ctx.close()
        rg.release()
        torch.cuda.empty_cache()
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, sys.argv[1:])   # (does not return)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    exchange = world > 1 or args.force_exchange
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    env = {"rank": rank, "world": world, "local_rank": local_rank, "exchange": exchange, "host_staged": rehearsal}

    import weath3rb0i_amd as w3

    bs = args.block_size
    model, model_name = make_model(w3, args.model)
    ctx = w3.Context(local_rank)
    ctx.set_path(args.path)
    ctx.set_coder(args.coder)
    ctx.set_verify(0 if args.no_verify else args.verify)
    ctx.set_tune(args.tune)
    if args.variant:
        ctx.set_variant(*args.variant.split(","))

    readings = ["strong", "weak"] if (args.scaling == "both" and world > 1) else ["strong" if args.scaling == "both" else args.scaling]
    results = {}
    rg = None
    for rd in readings:
        if rg is not None:
            rg.release()
        rg = Regime(env, rd, args.size, args.data, bs, args.seed)
        # encodes in flight for THIS reading's shard (a strong-scaled shard may be small enough for four free-running jobs)
        pipeline = args.pipeline if args.pipeline else ctx.max_in_flight(rg.n, bs, model)
        if args.tune & 8192:
            env["force_depth"] = 3   # (experiment: free-running jobs beyond 12,288 blocks)
        pipeline = min(pipeline, max(ctx.max_in_flight(rg.n, bs, model), env.get("force_depth", 0))) if pipeline > 1 else pipeline
        dt, acc = rg.timed(ctx, model, args.steps, args.warmup, pipeline)
        results[rd] = {"dt": dt, "acc": acc, "n": rg.n, "n_global": rg.n_global, "nb": rg.nb, "pipeline": pipeline,
                       "ratio": int(rg.d_totals[rg.last_buf].item()) / max(rg.n, 1),
                       "exchange_totals": list(rg.gathered) if rg.gathered else None}
    main_rd = readings[0]
    r0 = results[main_rd]
    if len(readings) > 1:
        # the main line's per-kernel figures belong to the FIRST reading; the regime object left alive is the last one's: nothing
        # below (extras run only at world == 1) needs it
        pass
    dt, acc, n, n_global, nb, pipeline = r0["dt"], r0["acc"], r0["n"], r0["n_global"], r0["nb"], r0["pipeline"]
    path = int(acc.get("path", 0))
    coder_name = args.coder if (args.coder != "x4" or (pipeline == 1 and "half_cu" not in args.variant)) else "x5"

    extras = world == 1 and not exchange and rank == 0
    ref_model = decode = sync_line = None
    kept = []   # (compressed stream, lens) of the live output buffers of the timed steps
    if extras:
        for k in sorted({(rg.last_buf - d) % rg.nsets for d in range(min(max(pipeline, 1), args.steps, rg.nsets))}):
            tot = int(rg.d_totals[k].item())
            kept.append((rg.d_outs[k][:tot].clone(), rg.d_lenss[k][:nb].clone()))
        if pipeline >= 2 and not args.no_ref_model:
            # the same workload with ONE synchronous call per step (3 steps, outside the timed region): what a lone call takes, and the
            # coder's launch with nothing beside it — the floor of a strong-scaled run
            dts, accs = rg.timed(ctx, model, 3, 1, 1)
            sync_line = {"value": round(n * 3 / dts / 2**20, 2), "unit": "MiB/s", "ms_per_step": round(dts / 3 * 1e3, 3),
                         "kernel_ms_per_step": {k: round(accs.get(k, 0.0) / 3, 3) for k in ("predict_ms", "apm_ms", "coder_ms", "pack_ms")},
                         "note": "w3_encode_blocks_device, one call at a time (k_coder_x4, full kernel shapes)"}
            if int(accs.get("path", 0)) == 2:
                sync_line["_rows"] = kernel_table(args.model, n, bs, 3, accs, "x4" if args.coder == "x4" else args.coder)
        if not args.no_ref_model and args.model != "order012" and args.data == "text":
            # the largest model whose streams are entirely the reference's (no build-defined node): same input, 3 steps, outside the
            # timed region (the main model's outputs are kept aside for the checks first)
            m2, m2_name = make_model(w3, "order012")
            dt2, a2 = rg.timed(ctx, m2, 3, 1, pipeline)
            ref_model = {"model": m2_name, "value": round(n * 3 / dt2 / 2**20, 2), "unit": "MiB/s", "ms_per_step": round(dt2 / 3 * 1e3, 3), "pipeline": pipeline,
                         "kernel_ms_per_step": {k: round(a2.get(k, 0.0) / 3, 3) for k in ("predict_ms", "apm_ms", "coder_ms", "pack_ms")},
                         "compressed_ratio": round(int(rg.d_totals[rg.last_buf].item()) / n, 4),
                         "note": "every node of this model is the reference's (Order0/Order1/OrderN + OpinionMixer2): its block streams are the reference's streams"}
    small_lines = None
    if extras and not args.no_other_configs and args.model == "order012apm" and args.data == "text" and args.size >= 100_000_000:
        # Small inputs of the bench model (w3_encode_max_in_flight = 4 free-running jobs): configs[1] at its literal enwik8 size, and ONE
        # RANK'S SHARE of the stream at 8 GPUs — the strong reading's per-GPU work, measured here on one GPU; 8 x its rate (minus the
        # exchange, which overlaps the next step) is what an 8-GPU strong-scaled run can reach.  On the main run's context (its job
        # workspaces are large enough already), outside the timed region.
        small_lines = []
        for label, size in (("configs[1] at enwik8 size", 100_000_000), ("one rank's share of the stream at 8 GPUs (strong reading)", args.size // 8 // bs * bs)):
            try:
                r = short_run(w3, args.model, "text", size, bs, args.seed, 12, env, ctx=ctx)
                r["what"] = label
                if "share" in label:
                    r["projected_8gpu_strong_MiBps"] = round(8 * r["value"], 1)
                    r["note"] = "projection = 8 x this rate: the ranks code disjoint block ranges with no data-path collective; the gather of ~0.38 x bytes to rank 0 overlaps the next step"
                small_lines.append(r)
            except Exception as e:
                small_lines.append({"what": label, "error": str(e)[:300]})
    # release the encoder's workspaces (two jobs' worth) before the decoder and the other configurations need the memory
    ctx.close()
    host = rg.host
    d_in_keep = rg.d_in if extras else None
    rg.d_outs = rg.d_lenss = rg.d_totals = rg.gather_buf = None
    torch.cuda.empty_cache()

    if extras and not args.no_decode:
        # independent full check + decode rate: EVERY block of the last timed step's output through the device decoder (lane per
        # block, k_generic_nl / k_cm_nl: no kernel in common with the predict phase), compared with the input on the device
        dctx = w3.Context(local_rank)
        try:
            comp, lens = kept[-1]
            back = torch.empty(n, dtype=torch.uint8, device="cuda")
            ddts = []
            for _ in range(2):   # the first call also allocates the lanes' tables (tens of GB): the second one is the rate
                back.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dctx.decode_blocks_device(model, comp, lens, bs, n, back)
                torch.cuda.synchronize()
                ddts.append(time.perf_counter() - t0)
            decode = {"value": round(n / ddts[1] / 2**20, 2), "unit": "MiB/s", "seconds": round(ddts[1], 3), "first_call_seconds": round(ddts[0], 3), "blocks": nb,
                      "roundtrip_all_blocks": bool(torch.equal(back, d_in_keep)),
                      "note": "w3_decode_blocks_device over the whole output of the last timed step, outside the timed region (second of two calls)"}
            del back
        finally:
            dctx.close()
            torch.cuda.empty_cache()
        if not decode["roundtrip_all_blocks"]:   # a rate for streams that do not decode is not a result: no line is printed
            raise SystemExit("bench.py: the device round trip of the last timed step's output FAILED (model %s, %d bytes): encoder and decoder disagree" % (args.model, n))

    host_path = None
    if extras and not args.no_host_path:
        # PCIe-inclusive ("report H2D/D2H separately"; never part of `value`): the same input from HOST memory —
        # compress() of main.rs:89-113 is file in, file out.  tools/host_api_rate.py, on a context of its own.
        try:
            from tools import host_api_rate
            hp = host_api_rate.measure(args.model, n, bs, chunks=(0,), calls=max(4, min(args.steps, 8)), pageable=True, resident=False, host=host, log=lambda *_: None)
            host_path = {"calls_in_flight_pinned_MiBps": hp["in_flight_pinned"]["MiBps"], "calls_in_flight_pinned_ms_per_call": hp["in_flight_pinned"]["ms_per_call"],
                         "calls_in_flight": hp["in_flight_pinned"]["calls_in_flight"],
                         "one_call_pinned_MiBps": hp["sync_pinned"][0]["MiBps"], "one_call_pinned_ms": hp["sync_pinned"][0]["ms_per_call"], "one_call_pieces": hp["sync_pinned"][0]["pieces"],
                         "one_call_pageable_MiBps": hp.get("sync_pageable", {}).get("MiBps"), "one_call_pageable_ms": hp.get("sync_pageable", {}).get("ms_per_call"),
                         "h2d_pinned_ms": hp["h2d_pinned_ms"], "h2d_pageable_ms": hp.get("h2d_pageable_ms"), "d2h_pinned_ms": hp.get("d2h_pinned_ms"),
                         "bytes": n, "compressed_ratio": hp.get("compressed_ratio"), "unit": "MiB/s",
                         "note": "input and output in HOST memory, PCIe both ways inside the timed region; calls_in_flight = w3_encode_host_submit / w3_encode_host_wait "
                                 "(the next call's H2D and the previous call's D2H overlap this call's encode); one_call = ONE synchronous w3_encode_blocks at a time, "
                                 "pipelined in pieces inside the call (it cannot hide its first copy in, its last coder chain and its last copy out); outputs byte-identical"}
        except Exception as e:   # an extra must not lose the main line
            host_path = {"error": str(e)[:300]}

    res = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n_global * args.steps / dt / 2**20
        nsmall, nwide, nslot, napm = MODEL_SHAPE[args.model]
        ncnt = nsmall + nwide
        ratio = r0["ratio"]
        if path == 2:
            rows = kernel_table(args.model, n, bs, args.steps, acc, coder_name)
        else:
            # A = 1 + c (stream write) + 64 B of Counter RMW per table model and input byte
            # + 384 B per slot-state leaf (2 nibbles x 96-B cell read + write) + 48 B per APM stage (8 x (4 B read + 2 B write))
            rows = [("k_cm" if (nslot or napm) else "k_generic", acc.get("generic_ms", 0.0) / args.steps, n * (1 + ratio + 64 * ncnt + 384 * nslot + 48 * napm))]
        rows = rows or [("none", 1e-9, 0)]
        table = []
        for name, ms, nbytes in rows:
            tr, src = lookup_traffic(args.model, n, bs, name)
            table.append({"kernel": name, "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(nbytes),
                          "achieved": round(nbytes / (ms * 1e-3) / 1e9, 2), "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                          "traffic": tr, "traffic_source": src})
        dom = max(table, key=lambda r: r["avg_launch_ms"])   # the time-dominant kernel
        # The same choice among the kernels of the ONE-CALL-AT-A-TIME leg (no other call's kernels beside any launch): stable from round to round,
        # where the overlapped pick flips with whatever the pipeline stretches (round 2: k_apm0, round 3: the coder beside the rank kernels).
        solo = None
        if sync_line and sync_line.get("_rows"):
            srows = sync_line.pop("_rows")
            sname, sms, sbytes = max(srows, key=lambda r: r[1])
            str_, ssrc = lookup_traffic(args.model, n, bs, sname)
            solo = {"bound": "hbm", "kernel": sname, "achieved": round(sbytes / (sms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(sbytes / (sms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), "traffic": str_, "traffic_source": ssrc, "avg_launch_ms": round(sms, 4),
                    "algorithmic_bytes_per_launch": int(sbytes),
                    "kernels": [{"kernel": r[0].split(" (")[0], "ms": round(r[1], 3), "frac": round(r[2] / (r[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)} for r in srows],
                    "chosen_by": "longest average launch of the one-call-at-a-time leg (3 synchronous calls: no other CALL's kernels beside any launch; inside a call the "
                                 "time-ordered leaf's kernel runs on the side stream beside the first rank kernel, which is why that launch is the longest)"}
        elif sync_line:
            sync_line.pop("_rows", None)
        coder_ms = acc.get("coder_ms", 0.0) / args.steps
        steps_per_lane = 8 * min(bs, max(n, 1))
        readings_txt = {"strong": "ONE stream of %d bytes cut over %d GPU(s)" % (r0["n_global"], world),
                        "weak": "%d bytes per GPU (%d in all)" % (n, n_global)}[main_rd]
        workload = ("%s synthetic (tools/synth.c seed %d), %s scaling: %s, %d-byte blocks, model %s, %s"
                    % ("enwik9-shaped text" if args.data == "text" else "Silesia-shaped mix", args.seed, main_rd, readings_txt, bs, model_name,
                       "%d encodes in flight (w3_encode_submit / w3_encode_wait)" % pipeline if pipeline >= 2 else "one synchronous call per step"))
        # the predict phase is several kernels; as a whole: algorithmic bytes (per leaf input + 16-byte stream, plus 8 B written + 8 B read
        # per record pass of a wide leaf) over the phase's time.  w3_timing.predict_bytes also carries the APM stages' bytes: take the
        # single ORDER0 stage's out again; models with slot leaves or several stages get no predict-phase figure
        predict_phase = None
        if path == 2 and acc.get("predict_ms", 0) > 0 and nslot == 0 and napm <= 1:
            pb = acc["predict_bytes"] / args.steps - (n * (16 * ncnt + 17) if napm == 1 else 0)
            pms = acc["predict_ms"] / args.steps
            predict_phase = {"algorithmic_bytes_per_step": int(pb), "ms": round(pms, 3), "achieved_GBps": round(pb / pms / 1e6, 1),
                             "frac_of_hbm_peak": round(pb / pms / 1e6 / HBM_PEAK_GBPS, 4)}
        exch = "none (1 GPU)"
        if exchange:
            exch = ("all_gather sizes + grouped send/recv to rank 0 (%s) over %d rank(s), overlapped with the next step's encode; last step gathered %s bytes (per rank %s)"
                    % ("gloo through host memory: REHEARSAL, ranks share GPUs" if rehearsal else "RCCL", dist.get_world_size(),
                       sum(r0["exchange_totals"] or [0]), r0["exchange_totals"]))
        res = {
            "metric": "encode MiB/s, %d KiB blocks, bit-exact vs CPU ref" % (bs >> 10),
            "value": round(value, 2), "unit": "MiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": main_rd, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "bytes_per_gpu": n, "bytes_total": n_global, "block_size": bs, "blocks_per_gpu": nb,
                       "context_model": model_name, "path": {1: "generic", 2: "twophase"}.get(path, str(path)), "compressed_ratio": round(ratio, 4),
                       "encodes_in_flight": pipeline, "exchange": exch, "ranks_seen_by_rccl": dist.get_world_size() if exchange else 1,
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default: 4 per priority level)")},
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": dom["frac"], "traffic": dom["traffic"], "traffic_source": dom["traffic_source"],
                         "avg_launch_ms": dom["avg_launch_ms"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "launches_per_step": 1,
                         "chosen_by": "longest average launch among the step's kernels (hipEvents on each kernel's launch stream)",
                         "note": ("with several encodes in flight the coder's launch is stretched by the kernels beside it (one latency chain per lane: %s ms with nothing "
                                  "beside it, floors.coder_floor_ms) — the kernels that own the chip are in roofline_kernels (k_apm0, k_rank_sorted) and whole_step is the step's "
                                  "algorithmic bytes over its time" % (sync_line["kernel_ms_per_step"]["coder_ms"] if sync_line else "?")) if pipeline >= 2 and "coder" in dom["kernel"] else None,
                         "solo": solo,
                         "kernels": [{"kernel": r["kernel"].split(" (")[0], "ms": r["avg_launch_ms"], "frac": r["frac"]} for r in table]},
            "roofline_solo": solo,
            "roofline_kernels": table,
            "whole_step": {"algorithmic_bytes": int(sum(r["algorithmic_bytes_per_launch"] for r in table)),
                           "achieved_GBps": round(sum(r["algorithmic_bytes_per_launch"] for r in table) / (ms_per_step * 1e-3) / 1e9, 1),
                           "frac_of_hbm_peak": round(sum(r["algorithmic_bytes_per_launch"] for r in table) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "kernel_ms_per_step": {k: round(acc.get(k, 0.0) / args.steps, 3) for k in ("predict_ms", "achash_ms", "slot_ms", "apm_ms", "coder_ms", "pack_ms", "generic_ms")},
            "predict_phase": predict_phase,
            # the coder is ONE dependent chain of 8 x block_size bit-steps per lane: its time does not shrink with the block count,
            # so it is the floor of a strong-scaled run (predict / APM / pack scale with the bytes per GPU)
            "floors": {"coder_floor_ms": (sync_line["kernel_ms_per_step"]["coder_ms"] if sync_line else round(coder_ms, 3)) if path == 2 else None,
                       "coder_launch_ms_in_this_run": round(coder_ms, 3) if path == 2 else None,
                       "bit_steps_per_lane": steps_per_lane,
                       "cycles_per_bit_step": (round((sync_line["kernel_ms_per_step"]["coder_ms"] if sync_line else coder_ms) * 1e-3 * SHADER_GHZ * 1e9 / steps_per_lane, 1)
                                               if path == 2 else None),
                       "clock_ghz_assumed": SHADER_GHZ,
                       "note": "coder_floor_ms = the coder's launch with nothing beside it (synchronous call): 8 x block_size dependent steps per lane, it does not "
                               "shrink with the block count; with two encodes in flight its launch stretches beside the next step's rank kernels"},
        }
        if "weak" in results and main_rd != "weak":
            rw = results["weak"]
            res["weak"] = {"value": round(rw["n_global"] * args.steps / rw["dt"] / 2**20, 2), "unit": "MiB/s", "ms_per_step": round(rw["dt"] / args.steps * 1e3, 3),
                           "bytes_per_gpu": rw["n"], "bytes_total": rw["n_global"], "steps": args.steps, "warmup": args.warmup,
                           "note": "weak reading: every GPU codes its own --size bytes; same K steps, same barriers"}
        if sync_line:
            res["one_call_at_a_time"] = sync_line
            res["config"]["one_call_at_a_time"] = {"value": sync_line["value"], "unit": "MiB/s", "ms_per_step": sync_line["ms_per_step"]}
        if ref_model:
            res["reference_stream_model"] = ref_model
            res["reference_stream_model_value"] = ref_model["value"]
            res["config"]["reference_stream_model"] = {"model": ref_model["model"], "value": ref_model["value"], "unit": "MiB/s", "ms_per_step": ref_model["ms_per_step"],
                                                       "note": "same input, every node the reference's own (no build-defined APM): these block streams ARE the reference's streams"}
        if decode:
            res["decode"] = decode
            res["config"]["decode"] = {"value": decode["value"], "unit": "MiB/s", "roundtrip_all_blocks": decode["roundtrip_all_blocks"]}
        if host_path:
            res["host_path"] = host_path
            res["config"]["host_path"] = {k: host_path[k] for k in ("calls_in_flight_pinned_MiBps", "one_call_pinned_MiBps", "one_call_pageable_MiBps", "h2d_pinned_ms", "d2h_pinned_ms") if k in host_path}

    if extras and not args.no_other_configs and args.model == "order012apm" and args.data == "text" and args.size >= 100_000_000:
        # BASELINE configs[2] (full CM) at the same size and configs[4]'s shape (Silesia-sized mix, 256 KiB blocks, the hash-map model)
        d_in_keep = None
        rg.release()
        oc = []
        for mname, kind, size, obs, nsteps in (("fullcm", "text", args.size, bs, 3), ("fullcm", "mixed", 211_938_580, 262144, 3), ("fullcm", "text", 100_000_000, bs, 8)):
            try:
                oc.append(short_run(w3, mname, kind, size, obs, args.seed, nsteps, env))
                if size == 100_000_000:
                    oc[-1]["what"] = "configs[2] at its literal enwik8 size"
            except Exception as e:   # an extra must not lose the main line
                oc.append({"context_model": mname, "data": kind, "error": str(e)[:300]})
        res["other_configs"] = oc
        if small_lines is not None:
            res["small_inputs"] = small_lines

    if rank == 0:
        if extras and not args.no_cpu_baseline:
            cb, cout, clens, cn = cpu_baseline(args.model, host, bs)
            # the baseline run doubles as a bit-exactness check of the timed GPU output: every output buffer of the timed steps that is still live
            nchk = len(clens)
            ok = True
            for comp, lens in kept:
                g_lens = lens[:nchk].cpu().numpy().astype(np.uint32)
                g_out = comp[: int(g_lens.sum())].cpu().numpy()
                ok = ok and bool(np.array_equal(g_lens, clens) and np.array_equal(g_out, cout))
            cb["bit_exact_vs_gpu"] = ok
            cb["buffers_checked"] = len(kept)
            res["cpu_baseline"] = cb
        print(json.dumps(res), flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
