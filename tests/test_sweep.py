"""The OrderN parameter sweep (weath3rb0i_amd/sweep.py, the reference's bin/ordern/main.rs): best-tracking logic on the
CPU with a stand-in context, and the device csizes against the oracle."""
import numpy as np
import pytest

from weath3rb0i_amd import sweep
from tests.synth import markov_text


class FakeCtx:
    """encode_blocks() with a known csize per (bits, align): csize = table value, one block."""

    def __init__(self, table):
        self.table = table

    def encode_blocks(self, model, data, block_size):
        spec = model.spec()
        nd = spec.nodes[0]
        return None, np.array([self.table[(nd.bits, nd.align)]], dtype=np.uint32)


def test_best_tracking_follows_the_reference_rule():
    """bin/ordern/main.rs:16-37: per-ctx best resets every ctx_bits; on equal csize the LATER configuration wins
    (`if res > best[i] { continue }`); a configuration no better than the input length still replaces (0, 0) only if <=."""
    table = {(8, 0): 50, (8, 1): 40, (8, 2): 40, (9, 0): 45, (9, 1): 39, (9, 2): 60, (10, 0): 39, (10, 1): 70, (10, 2): 39}
    lines = []
    best, params, got = sweep.sweep_ordern(FakeCtx(table), bytes(100), 64, range(8, 11), range(0, 3), repeats=1, out=lines.append)
    assert got == table
    assert (best, params) == (39, (10, 2))                       # 39 three times: the last one stays
    assert [l for l in lines if l.startswith("-> best")] == [
        "-> best: 40 for [ctx: 8, align: 2]", "-> best: 39 for [ctx: 9, align: 1]", "-> best: 39 for [ctx: 10, align: 2]"]
    assert lines[-1] == "-> gloabl best: 39 for [ctx: 10, align: 2]"
    assert lines[0].startswith("[ordern] [ctx:  8, align: 0] csize: 50 (ratio: 0.500), ctime: ")


def test_alignment_above_context_bits_is_skipped():
    lines = []
    _, _, got = sweep.sweep_ordern(FakeCtx({(2, a): 7 for a in range(3)}), bytes(10), 64, range(2, 3), range(0, 5), repeats=1, out=lines.append)
    assert sorted(got) == [(2, 0), (2, 1), (2, 2)]


@pytest.mark.gpu
def test_sweep_csizes_match_the_oracle(oracle):
    import weath3rb0i_amd as w3
    data = markov_text(3 * 4096 + 100, seed=51)
    ctx = w3.Context(0)
    try:
        best, params, got = sweep.sweep_ordern(ctx, data, 4096, range(8, 13), range(0, 5), repeats=1, out=lambda s: None)
    finally:
        ctx.close()
    want = {}
    for (b, a) in got:   # csize = (sum of the blocks' ACStats bit counts) / 8 (helpers.rs:70-73)
        want[(b, a)] = sum(oracle.encode_stats_bits(oracle.OrderN(b, a), data[o:o + 4096]) for o in range(0, len(data), 4096)) // 8
    assert got == want
    assert best == min(want.values()) and want[params] == best


@pytest.mark.gpu
def test_entropy_hashing_sweeps_match_the_oracle(oracle):
    """bin/entropy-hashing-ac and bin/entropy-hashing-huff as sweeps over the counting sink: csizes equal the oracle's ACStats per block,
    hashes of 5 .. 16 bits (k_achash / k_huffkeys with k_predict_small, k_achash32 / k_huffkeys<true> with k_predict_wave), best per level
    with the reference's tie rule, the reference's line format."""
    import weath3rb0i_amd as w3
    data = markov_text(3 * 4096 + 100, seed=52)
    bs = 4096
    blocks = [data[o:o + bs] for o in range(0, len(data), bs)]
    ctx = w3.Context(0)
    try:
        lines = []
        best, params, got = sweep.sweep_entropy_ac(ctx, data, bs, range(8, 20, 4), range(0, 4), repeats=1, out=lines.append)
        want = {}
        for (b, a) in got:
            want[(b, a)] = sum(oracle.encode_stats_bits(oracle.OrderNEntropy(b, a, oracle.ACHistory(b - a, oracle.StationaryModel(buf=data))), blk)
                               for blk in blocks) // 8
        assert got == want
        assert best == min(want.values()) and want[params] == best
        assert lines[0].startswith("[eh-ac] [ctx:  8, align: 0] csize: %d (ratio " % want[(8, 0)])
        assert lines[-1] == "-> gloabl best: %d for [ctx: %d, align: %d]" % (best, params[0], params[1])
        lines = []
        best, params, got = sweep.sweep_entropy_huff(ctx, data, bs, range(9, 11), range(10, 12), range(8, 21, 6), repeats=1, out=lines.append)
        want = {}
        for (r, h, b) in got:
            want[(r, h, b)] = sum(oracle.encode_stats_bits(oracle.OrderNEntropy(b, 0, oracle.HuffHistory(data, h, r)), blk) for blk in blocks) // 8
        assert got == want and len(got) == 12
        assert best == min(want.values()) and want[params] == best
        assert lines[-1].startswith("---> global best: %d for [rem_hsize: " % best)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_counting_sink_and_sweep_and_export_match_the_oracle(oracle):
    """A14 (ACStats) on every coder kernel and both paths; the one-launch sweep per (configuration, block); the context
    statistics export against the oracle's Counter table semantics."""
    import weath3rb0i_amd as w3
    from tests.synth import lcg_text
    data = markov_text(5 * 8192 + 333, seed=52) + bytes(9000) + lcg_text(7000, seed=9)
    bs = 8192
    blocks = [data[o:o + bs] for o in range(0, len(data), bs)]
    ctx = w3.Context(0)
    try:
        cases = [("order0", w3.Order0, oracle.Order0), ("best012", lambda: w3.BestOfTwoModel(w3.BestOfTwoModel(w3.Order0(), w3.Order1()), w3.OrderN(27, 3)),
                                                        lambda: oracle.BestOfTwoModel(oracle.BestOfTwoModel(oracle.Order0(), oracle.Order1()), oracle.OrderN(27, 3))),
                 ("ordern_14_4", lambda: w3.OrderN(14, 4), lambda: oracle.OrderN(14, 4)),
                 ("o012apm", lambda: w3.APM(w3.BestOfTwoModel(w3.Order0(), w3.Order1())), lambda: oracle.APM(oracle.BestOfTwoModel(oracle.Order0(), oracle.Order1())))]
        for name, dev, orc in cases:
            want = [oracle.encode_stats_bits(orc(), blk) for blk in blocks]
            for path in ("auto", "generic"):
                ctx.set_path(path)
                for coder in (("x4", "x3", "x2", "fast", "robust") if path == "auto" else ("x4",)):
                    ctx.set_coder(coder)
                    got = ctx.encode_stats(dev(), data, bs)
                    assert got.tolist() == want, (name, path, coder)
        ctx.set_path("auto"); ctx.set_coder("x4")
        configs = [(8, 0), (11, 3), (12, 4), (19, 3), (22, 2), (27, 3), (30, 1)]
        got = ctx.sweep_ordern(data, bs, configs)
        for c, row in zip(configs, got):
            assert row.tolist() == [oracle.encode_stats_bits(oracle.OrderN(*c), blk) for blk in blocks], c
        # export: Counter (n0, n1) per context after the whole input as one stream == a direct replay of Counter::update
        n0, n1 = ctx.export_counters(w3.OrderN(11, 3), data[:20000])
        c0 = np.zeros(1 << 11, dtype=np.int64); c1 = np.zeros(1 << 11, dtype=np.int64)
        hist = t = 0
        for byte in data[:20000]:
            for j in range(8):
                bit = (byte >> (7 - j)) & 1
                cx = (((hist & 0xFF) << 3) | (t & 7)) if t else 0
                if bit: c1[cx] += 1
                else: c0[cx] += 1
                if (c1[cx] if bit else c0[cx]) == 65535:
                    c0[cx] = (c0[cx] >> 1) + (c0[cx] & 1); c1[cx] = (c1[cx] >> 1) + (c1[cx] & 1)
                hist = (hist << 1) | bit; t += 1
        assert n0.astype(np.int64).tolist() == c0.tolist() and n1.astype(np.int64).tolist() == c1.tolist()
    finally:
        ctx.close()
