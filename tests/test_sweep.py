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
    for (b, a) in got:
        _, lens = oracle.encode_blocks(oracle.OrderN(b, a), data, 4096, nthreads=8)
        want[(b, a)] = int(lens.sum())
    assert got == want
    assert best == min(want.values()) and want[params] == best
