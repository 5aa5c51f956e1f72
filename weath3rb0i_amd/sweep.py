"""Parameter sweeps on the device — the reference's research drivers `src/bin/ordern/main.rs:9-58` (one launch for every
configuration), `src/bin/entropy-hashing-ac/main.rs` and `src/bin/entropy-hashing-huff/main.rs` (one counting-sink encode per
configuration: every one of them runs on the two-phase path since round 3).

The reference runs `OrderN::new(ctx_bits, alignment_bits)` over one file for ctx_bits 8..=30 x alignment_bits 0..=4,
three times each, prints `[ordern] [ctx: B, align: A] csize: N (ratio: r), ctime: t (t/bit per bit)` for the fastest run
and tracks the best parameters per ctx_bits and overall (`best`/`params`, levels = 2).  Here ALL configurations run in one
device launch (`w3_sweep_ordern`: lanes = configurations x blocks, counting sink ACStats as in the reference's `compress`,
`main.rs:66-80`) on the block container: csize = (sum of the blocks' bit counts) / 8, each block
coded alone with a fresh model — so the figures are those of the reference run per block, not of its single
whole-file stream; ctime is the launch's time divided by the number of configurations.  Same line format, same tie rule (a later configuration replaces the best on equality,
`main.rs:21-26`: `if res > best[i] { continue }`).
"""
import sys
import time

from . import models


def exec_one(ctx, data, block_size, ctx_bits, alignment_bits, repeats=3, out=print):
    """`exec` (`bin/ordern/main.rs:42-66`): fastest of `repeats` runs; returns (csize, seconds)."""
    model = models.OrderN(ctx_bits, alignment_bits)
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        _, lens = ctx.encode_blocks(model, data, block_size)
        dt = time.perf_counter() - t0
        res = int(lens.sum())
        if best is None or dt < best[1]:
            best = (res, dt)
    res, dt = best
    nbits = max(1, len(data) * 8)
    out("[ordern] [ctx: %2d, align: %d] csize: %d (ratio: %.3f), ctime: %.3fms (%.3fns per bit)"
        % (ctx_bits, alignment_bits, res, res / max(1, len(data)), dt * 1e3, dt * 1e9 / nbits))
    return res, dt


def sweep_ordern(ctx, data, block_size=65536, ctx_bits=range(8, 31), alignment_bits=range(0, 5), repeats=3, out=print):
    """`main` (`bin/ordern/main.rs:9-40`).  Returns (global best csize, (ctx_bits, alignment_bits), {(B, A): csize})."""
    levels = 2
    best = [len(data)] * levels
    params = [(0, 0)] * levels
    table = {}
    pre = {}
    configs = [(b, a) for b in ctx_bits for a in alignment_bits if a <= b]
    if hasattr(ctx, "sweep_ordern") and configs:   # every configuration x block in ONE launch, fastest of `repeats`
        dt_best = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            bits = ctx.sweep_ordern(data, block_size, configs)
            dt = time.perf_counter() - t0
            dt_best = dt if dt_best is None else min(dt_best, dt)
        for c, row in zip(configs, bits):
            pre[c] = (int(row.astype("uint64").sum()) // 8, dt_best / len(configs))   # ACStats::result(): bits / 8 (helpers.rs:70-73)
    for b in ctx_bits:
        best[1] = len(data)
        params[1] = (0, 0)
        for a in alignment_bits:
            if a > b:          # OrderN needs alignment_bits <= bits_in_context (the crate would underflow `bits - align`)
                continue
            if (b, a) in pre:
                res, dt = pre[(b, a)]
                out("[ordern] [ctx: %2d, align: %d] csize: %d (ratio: %.3f), ctime: %.3fms (%.3fns per bit)"
                    % (b, a, res, res / max(1, len(data)), dt * 1e3, dt * 1e9 / max(1, len(data) * 8)))
            else:
                res, _ = exec_one(ctx, data, block_size, b, a, repeats, out)
            table[(b, a)] = res
            for i in range(levels):
                if res > best[i]:
                    continue
                best[i] = res
                params[i] = (b, a)
        out("-> best: %d for [ctx: %d, align: %d]" % (best[1], params[1][0], params[1][1]))
    out("-> gloabl best: %d for [ctx: %d, align: %d]" % (best[0], params[0][0], params[0][1]))   # (sic, main.rs:35)
    return best[0], params[0], table


def _stats_csize(ctx, model, data, block_size, repeats):
    """ACStats csize (helpers.rs:70-73: bits / 8) of `model` over the block container, fastest of `repeats` device runs."""
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        bits = ctx.encode_stats(model, data, block_size)
        dt = time.perf_counter() - t0
        res = int(bits.astype("uint64").sum()) // 8
        if best is None or dt < best[1]:
            best = (res, dt)
    return best


def sweep_entropy_ac(ctx, data, block_size=65536, ctx_bits=range(8, 31), alignment_bits=range(0, 5), repeats=1, out=print):
    """`src/bin/entropy-hashing-ac/main.rs:11-48`: OrderNEntropy(ctx_bits, alignment_bits, ACHistory(ctx_bits - alignment_bits,
    StationaryModel::new(buf))) for ctx_bits 8..=30 x alignment_bits 0..=4, line format of `exec` (:64-73), best per ctx_bits and
    overall with the reference's tie rule.  Every configuration runs on the two-phase path (hashes of up to 8 bits: k_achash +
    k_predict_small; wider: k_achash32 + k_predict_wave) through the counting sink.  Returns (best csize, (B, A), {(B, A): csize})."""
    levels = 2
    best = [len(data)] * levels
    params = [(0, 0)] * levels
    table = {}
    station = models.StationaryModel.new(data)
    for b in ctx_bits:
        best[1] = len(data)
        params[1] = (0, 0)
        for a in alignment_bits:
            if a > b:
                continue
            model = models.OrderNEntropy(b, a, models.ACHistory(b - a, station))
            res, dt = _stats_csize(ctx, model, data, block_size, repeats)
            out("[eh-ac] [ctx: %2d, align: %d] csize: %d (ratio %.3f), ctime: %.3fms (%.3fns per bit)"
                % (b, a, res, res / max(1, len(data)), dt * 1e3, dt * 1e9 / max(1, len(data) * 8)))
            table[(b, a)] = res
            for i in range(levels):
                if res > best[i]:
                    continue
                best[i] = res
                params[i] = (b, a)
        out("-> best: %d for [ctx: %d, align: %d]" % (best[1], params[1][0], params[1][1]))
    out("-> gloabl best: %d for [ctx: %d, align: %d]" % (best[0], params[0][0], params[0][1]))   # (sic, main.rs:43)
    return best[0], params[0], table


def sweep_entropy_huff(ctx, data, block_size=65536, rem_huff_sizes=range(7, 13), huff_sizes=range(8, 16), ctx_bits=range(8, 31), repeats=1, out=print):
    """`src/bin/entropy-hashing-huff/main.rs:11-50`: OrderNEntropy(ctx_bits, 0, HuffHistory::new(buf, huff_size, rem_huff_size)), three levels
    of best.  The tables come from w3_huff_tables (ties among equal counts in ascending symbol order: INTEGRATION.md)."""
    levels = 3
    best = [len(data)] * levels
    params = [(0, 0, 0)] * levels
    table = {}
    for r in rem_huff_sizes:
        best[1] = len(data)
        params[1] = (0, 0, 0)
        for h in huff_sizes:
            best[2] = len(data)
            params[2] = (0, 0, 0)
            try:
                hist = models.HuffHistory.new(data, h, r)
            except models.W3Error:   # the reference panics when the length limit is too small for the alphabet (package_merge.rs)
                out("[eh-huff] [rem_hsize: %2d, hsize: %2d] length limit too small for the alphabet" % (r, h))
                continue
            for b in ctx_bits:
                res, dt = _stats_csize(ctx, models.OrderNEntropy(b, 0, hist), data, block_size, repeats)
                out("[eh-huff] [rem_hsize: %2d, hsize: %2d, ctx: %2d] csize: %d (ratio: %.3f), ctime: %.3fms (%.3fns per bit)"
                    % (r, h, b, res, res / max(1, len(data)), dt * 1e3, dt * 1e9 / max(1, len(data) * 8)))
                table[(r, h, b)] = res
                for i in range(levels):
                    if res > best[i]:
                        continue
                    best[i] = res
                    params[i] = (r, h, b)
            out("-> best: %d for [rem_hsize: %d, hsize: %d, ctx: %d]" % ((best[2],) + params[2]))
        out("--> best: %d for [rem_hsize: %d, hsize: %d, ctx: %d]" % ((best[1],) + params[1]))
    out("---> global best: %d for [rem_hsize: %d, hsize: %d, ctx: %d]" % ((best[0],) + params[0]))
    return best[0], params[0], table


def main(argv=None):
    import argparse
    from .api import Context
    ap = argparse.ArgumentParser(description="parameter sweeps of the reference's research drivers on the GPU, block container: "
                                             "ordern (bin/ordern), eh-ac (bin/entropy-hashing-ac), eh-huff (bin/entropy-hashing-huff)")
    ap.add_argument("path")
    ap.add_argument("--driver", default="ordern", choices=["ordern", "eh-ac", "eh-huff"])
    ap.add_argument("--block-size", type=int, default=65536)
    ap.add_argument("--ctx-bits", default="8:30", help="lo:hi inclusive")
    ap.add_argument("--align-bits", default="0:4", help="lo:hi inclusive")
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args(argv)
    data = open(args.path, "rb").read()
    lo, hi = (int(x) for x in args.ctx_bits.split(":"))
    alo, ahi = (int(x) for x in args.align_bits.split(":"))
    ctx = Context(0)
    try:
        if args.driver == "ordern":
            sweep_ordern(ctx, data, args.block_size, range(lo, hi + 1), range(alo, ahi + 1), args.repeats)
        elif args.driver == "eh-ac":
            sweep_entropy_ac(ctx, data, args.block_size, range(lo, hi + 1), range(alo, ahi + 1), args.repeats)
        else:
            sweep_entropy_huff(ctx, data, args.block_size, ctx_bits=range(lo, hi + 1), repeats=args.repeats)
    finally:
        ctx.close()


if __name__ == "__main__":
    sys.exit(main())
