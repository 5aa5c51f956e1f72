"""GPU parity of the CM path (slot-state leaves, APM chain; k_cm and its encoder fast paths) against the CPU
oracle, byte for byte, through the C ABI.  The models are BUILD-DEFINED: the oracle here is
this build's own restatement, pinned by the committed digests in tests/golden/cm_streams.json."""
import hashlib
import json

import numpy as np
import pytest

import weath3rb0i_amd as w3
from tests.synth import lcg_text, markov_text, mixed_bytes
from tests.test_cm_cpu import GOLDEN, golden_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = w3.Context(0)
    yield c
    c.close()


def o012(m):
    return m.BestOfTwoModel(m.BestOfTwoModel(m.Order0(), m.Order1()), m.OrderN(27, 3))


def full(m, slot, log_cells=14):
    t = o012(m)
    for order in (1, 2, 3, 4):
        t = m.BestOfTwoModel(t, slot(order, log_cells))
    return m.APM(m.APM(t, 0, 7), 1, 6)


def pair(oracle, name):
    o = oracle
    table = {
        "slot0": (lambda: w3.SlotModel(0, 4), lambda: o.SlotModel(0, 4)),
        "slot1": (lambda: w3.SlotModel(1, 10), lambda: o.SlotModel(1, 10)),
        "slot2": (lambda: w3.SlotModel(2, 12), lambda: o.SlotModel(2, 12)),
        "slot3_tiny_table": (lambda: w3.SlotModel(3, 1), lambda: o.SlotModel(3, 1)),     # constant eviction
        "slot7": (lambda: w3.SlotModel(7, 14), lambda: o.SlotModel(7, 14)),
        "slot2_hashmap_sized": (lambda: w3.SlotModel(2, w3.HashMap.new(1 << 20)), lambda: o.SlotModel(2, 13)),
        "apm0_order0": (lambda: w3.APM(w3.Order0()), lambda: o.APM(o.Order0())),
        "apm1_order0_r3": (lambda: w3.APM(w3.Order0(), w3.APM.ORDER1, 3), lambda: o.APM(o.Order0(), o.APM_ORDER1, 3)),
        "apm_rate15": (lambda: w3.APM(w3.Order1(), 0, 15), lambda: o.APM(o.Order1(), 0, 15)),
        "apm_rate1": (lambda: w3.APM(w3.Order1(), 0, 1), lambda: o.APM(o.Order1(), 0, 1)),
        "o012_apm": (lambda: w3.APM(o012(w3)), lambda: o.APM(o012(o))),
        "apm_chain4": (lambda: w3.APM(w3.APM(w3.APM(w3.APM(w3.Order0(), 0, 7), 1, 6), 0, 5), 1, 4),
                       lambda: o.APM(o.APM(o.APM(o.APM(o.Order0(), 0, 7), 1, 6), 0, 5), 1, 4)),
        "slot_mix": (lambda: w3.BestOfTwoModel(w3.SlotModel(2, 12), w3.BestOfTwoModel(w3.Order0(), w3.SlotModel(1, 12))),
                     lambda: o.BestOfTwoModel(o.SlotModel(2, 12), o.BestOfTwoModel(o.Order0(), o.SlotModel(1, 12)))),
        "apm_frozen": (lambda: w3.APM(w3.FrozenModel(w3.Order0())), lambda: o.APM(o.FrozenModel(o.Order0()))),
        "apm_main_default": (lambda: w3.APM(w3.init_model()),
                             lambda: o.APM(o.OrderNEntropy(11, 3, o.ACHistory(8, o.StationaryModel.for_book1())))),
        "full_cm": (lambda: w3.full_cm(), lambda: full(o, o.SlotModel)),
        "full_cm_small_tables": (lambda: full(w3, w3.SlotModel, 8), lambda: full(o, o.SlotModel, 8)),
    }
    return table[name]


NAMES = ["slot0", "slot1", "slot2", "slot3_tiny_table", "slot7", "slot2_hashmap_sized", "apm0_order0", "apm1_order0_r3", "apm_rate15",
         "apm_rate1", "o012_apm", "apm_chain4", "slot_mix", "apm_frozen", "apm_main_default", "full_cm", "full_cm_small_tables"]




def check(ctx, oracle, name, data, bs, decode=True):
    dev, orc = pair(oracle, name)
    out, lens = ctx.encode_blocks(dev(), data, bs)
    want, wlens = oracle.encode_blocks(orc(), data, bs, nthreads=8)
    assert lens.tolist() == wlens.tolist(), name
    assert out.tobytes() == want.tobytes(), name
    if len(data) >= 8:
        # encode runs on the two-phase path (Counter kernels, slot leaves by sorted replay — few blocks here —, k_apm0 / k_apm1);
        # k_slot (hash map in HBM, the form for many blocks) and the lane-per-block k_cm must agree with it
        assert ctx.timing()["path"] == 2, name
        if "slot" in name or "full_cm" in name:
            ctx.set_variant("slot_table")
            try:
                out3, lens3 = ctx.encode_blocks(dev(), data, bs)
                assert ctx.timing()["path"] == 2
            finally:
                ctx.set_variant()
            assert lens3.tolist() == wlens.tolist() and out3.tobytes() == want.tobytes(), name + " (k_slot)"
        ctx.set_path("generic")
        try:
            out2, lens2 = ctx.encode_blocks(dev(), data, bs)
            assert ctx.timing()["path"] == 1
        finally:
            ctx.set_path("auto")
        assert lens2.tolist() == wlens.tolist() and out2.tobytes() == want.tobytes(), name + " (k_cm)"
    if decode:
        assert ctx.decode_blocks(dev(), out, lens, bs, len(data)).tobytes() == bytes(data), name
        ctx.set_variant("decode_lane")   # the lane-per-block decoder (k_cm_nl), where the default is k_decode_spec (sixteen lanes per block)
        try:
            assert ctx.decode_blocks(dev(), out, lens, bs, len(data)).tobytes() == bytes(data), name + " (lane-per-block decoder)"
        finally:
            ctx.set_variant()
        if bs < 262144:                  # (256 KiB blocks: a decode is a latency chain of 2 M steps per block — the two-bit form is covered at the smaller sizes)
            ctx.set_tune(16384)          # k_decode_spec with two bits per speculated group (a tested variant)
            try:
                assert ctx.decode_blocks(dev(), out, lens, bs, len(data)).tobytes() == bytes(data), name + " (k_decode_spec, two-bit groups)"
            finally:
                ctx.set_tune(0)
        ctx.set_tune(262144)             # k_decode_spec with the nibble-major table formats of large batches (bucketed exact maps, APM tables by nibble group)
        try:
            assert ctx.decode_blocks(dev(), out, lens, bs, len(data)).tobytes() == bytes(data), name + " (k_decode_spec, nibble-major table formats)"
            if bs < 262144:
                ctx.set_tune(524288)            # the general kernel where an instance specialised for the model's shape would run (round-3 formats)
                assert ctx.decode_blocks(dev(), out, lens, bs, len(data)).tobytes() == bytes(data), name + " (k_decode_spec, general kernel)"
                ctx.set_tune(262144 | 524288)   # ... and with the nibble-major formats
                assert ctx.decode_blocks(dev(), out, lens, bs, len(data)).tobytes() == bytes(data), name + " (k_decode_spec, nibble-major, general kernel)"
        finally:
            ctx.set_tune(0)
    return out, lens


@pytest.mark.parametrize("name", NAMES)
def test_cm_models_encode_decode(ctx, oracle, name):
    data = markov_text(30000, seed=41) + lcg_text(6000, seed=4) + bytes(700) + mixed_bytes(5000, seed=6)
    check(ctx, oracle, name, data, 8192)


def test_cm_golden_digests_on_device(ctx):
    """The committed digests (tests/golden/cm_streams.json) straight from the device: no oracle in the loop."""
    want = json.load(open(GOLDEN))
    models = {"slot2": lambda: w3.SlotModel(2, 12), "apm0_order0": lambda: w3.APM(w3.Order0()), "o012_apm": lambda: w3.APM(o012(w3)),
              "full_cm": lambda: w3.full_cm()}
    for iname, data in golden_inputs().items():
        for mname, mk in models.items():
            out, lens = ctx.encode_blocks(mk(), data, len(data))
            assert [len(out), hashlib.sha256(out.tobytes()).hexdigest()] == want[iname][mname], (iname, mname)


def test_cm_edge_blocks(ctx, oracle):
    cases = {
        "zeros64k": (bytes(65536), 65536), "ones64k": (b"\xff" * 65536, 65536), "alt55": (b"\x55" * 5000, 4096), "single": (b"A", 4096),
        "two": (b"AB", 1), "bs_plus_1": (lcg_text(4097, seed=5), 4096), "bs_minus_1": (lcg_text(4095, seed=6), 4096),
        "ragged": (lcg_text(3 * 4096 + 17, seed=7), 4096),
        "random": (np.random.default_rng(1).integers(0, 256, 20000, dtype=np.uint8).tobytes(), 4096),
        "lanes_65": (markov_text(65 * 512, seed=8), 512),           # more than one wavefront, last one ragged
    }
    for cname, (data, bs) in cases.items():
        for name in ("slot2", "o012_apm", "full_cm_small_tables"):
            check(ctx, oracle, name, data, bs)
    out, lens = ctx.encode_blocks(w3.full_cm(), b"", 4096)
    assert len(out) == 0 and len(lens) == 0


def test_cm_256k_blocks_mixed(ctx, oracle):
    """BASELINE configs[4] shape: mixed text/binary, 256 KiB blocks (hash-map stress: 2^19 nibble contexts per leaf)."""
    data = mixed_bytes(600000, seed=5)
    check(ctx, oracle, "full_cm", data, 262144)


def test_cm_full_block_64k(ctx, oracle):
    data = markov_text(3 * 65536 + 1000, seed=77)
    check(ctx, oracle, "full_cm", data, 65536)
    check(ctx, oracle, "o012_apm", data, 65536)


@pytest.mark.parametrize("name", ["o012_apm", "apm1_order0_r3", "apm_chain4"])
def test_apm_twophase_shapes(ctx, oracle, name):
    """k_apm0 / k_apm1 corner cases: more blocks than one workgroup's waves, ragged last block, lengths that are not a
    multiple of the 8-position round, every previous-byte group present (random bytes), one giant group (constant
    bytes), group boundaries inside a round (short alternating runs)."""
    rng = np.random.default_rng(9)
    runs = b"".join(bytes([int(v)]) * int(r) for v, r in zip(rng.integers(0, 256, 3000), rng.integers(1, 12, 3000)))
    cases = [
        (markov_text(11 * 4096 + 1234, seed=12), 4096),
        (rng.integers(0, 256, 30011, dtype=np.uint8).tobytes(), 8192),
        (b"\x00" * 20003, 8192), (b"ab" * 9001, 4096), (runs, 4096),
        (markov_text(2 * 65536 + 777, seed=13), 65536),
        (lcg_text(5, seed=3), 4096), (lcg_text(9, seed=3), 8), (lcg_text(4, seed=3), 4096),
    ]
    for data, bs in cases:
        check(ctx, oracle, name, data, bs, decode=False)


def test_apm_predict_blocks(ctx, oracle):
    """Model::predict of an APM chain for every step, straight from the predict + APM kernels."""
    data = markov_text(40000, seed=21)
    for name in ("o012_apm", "apm_chain4"):
        dev, orc = pair(oracle, name)
        p = ctx.predict_blocks(dev(), data, 16384)
        want = np.concatenate([oracle.predict_all(orc(), data[o:min(o + 16384, 40000)]) for o in range(0, 40000, 16384)])
        assert np.array_equal(p, want), name


def test_apm_robust_coder_handback(ctx, oracle):
    """A lowered accumulator limit makes the fast coder hand blocks to the robust coder, which reads the APM's final stream."""
    data = markov_text(50000, seed=5)
    dev, orc = pair(oracle, "o012_apm")
    want, wlens = oracle.encode_blocks(orc(), data, 8192, nthreads=8)
    ctx.set_acc_limit(19)
    try:
        out, lens = ctx.encode_blocks(dev(), data, 8192)
        assert ctx.timing()["n_recoded_blocks"] > 0
    finally:
        ctx.set_acc_limit(46)
    assert lens.tolist() == wlens.tolist() and out.tobytes() == want.tobytes()


def test_slot_tables_in_batches(ctx, oracle):
    """When the hash maps of all blocks do not fit the device budget, k_slot runs in equal batches of blocks
    (tables zero-filled per batch); forced here with the tuning hook."""
    data = markov_text(130 * 512 + 77, seed=14)
    ctx.set_slot_budget_mb(40)                         # slot2: 2^12 cells x 128 B = 512 KiB per block -> 64 blocks per batch
    ctx.set_variant("slot_table")                      # (at this block count the default is the sorted replay, which has no tables)
    try:
        check(ctx, oracle, "slot2", data, 512, decode=False)
        assert ctx.timing()["path"] == 1                   # (check() ends on its k_cm cross-check)
        dev, _ = pair(oracle, "slot2")
        ctx.set_variant("slot_table")                  # (check() leaves the defaults behind)
        ctx.set_timing(True)
        ctx.encode_blocks(dev(), data, 512)
        assert ctx.timing()["n_slot_launches"] == 3
    finally:
        ctx.set_timing(False)
        ctx.set_slot_budget_mb(0)
        ctx.set_variant()


def test_slot_sorted_replay_shapes(ctx, oracle):
    """The sorted replay of the slot-state leaves (w3_slot2.h) where its bookkeeping has corners: Cells that fill one sort bin
    (2^8 and fewer: a single pass), leaves of different table sizes in one launch (one and two passes mixed), 2^16 Cells, a block
    longer than 64 KiB (events past 2^17), ragged last blocks of 1 .. 9 bytes, constant input (every event of a context in one Cell)."""
    text = markov_text(3 * 8192 + 5, seed=91)
    models = {
        "cells_2^8": (lambda: w3.SlotModel(2, 8), lambda: oracle.SlotModel(2, 8)),
        "cells_2^9": (lambda: w3.SlotModel(1, 9), lambda: oracle.SlotModel(1, 9)),
        "cells_2^16": (lambda: w3.SlotModel(3, 16), lambda: oracle.SlotModel(3, 16)),
        "mixed_sizes": (lambda: w3.BestOfTwoModel(w3.SlotModel(1, 4), w3.BestOfTwoModel(w3.SlotModel(2, 12), w3.SlotModel(3, 8))),
                        lambda: oracle.BestOfTwoModel(oracle.SlotModel(1, 4), oracle.BestOfTwoModel(oracle.SlotModel(2, 12), oracle.SlotModel(3, 8)))),
    }
    cases = [(text, 8192), (markov_text(2 * 131072 + 9, seed=92), 131072), (b"\x00" * 9000, 4096), (text[:8192 + 1], 8192), (text[:4096 * 2 + 9], 4096)]
    for mname, (dev, orc) in models.items():
        for data, bs in cases:
            want, wlens = oracle.encode_blocks(orc(), data, bs, nthreads=8)
            for variant in ("slot_sorted", "slot_table"):
                ctx.set_variant(variant)
                try:
                    out, lens = ctx.encode_blocks(dev(), data, bs)
                finally:
                    ctx.set_variant()
                assert lens.tolist() == wlens.tolist() and out.tobytes() == want.tobytes(), (mname, len(data), bs, variant)


def test_apm_models_submit_wait_pipeline(ctx, oracle):
    """w3_encode_submit / w3_encode_wait with APM chains (ORDER0 and ORDER1 stages) and the half-CU kernel shapes: two calls in
    flight, outputs identical to the oracle's; a spec with slot-state leaves runs synchronously inside submit."""
    import torch
    data = markov_text(500 * 1024 + 17, seed=33)
    host = np.frombuffer(data, dtype=np.uint8)
    n, bs = len(host), 512
    nb = (n + bs - 1) // bs
    d_in = torch.from_numpy(host.copy()).cuda()
    bufs = [(torch.empty(2 * n + 64 * nb + 64, dtype=torch.uint8, device="cuda"), torch.zeros(nb, dtype=torch.int32, device="cuda"),
             torch.zeros(1, dtype=torch.int64, device="cuda")) for _ in range(2)]
    torch.cuda.synchronize()
    assert ctx.max_in_flight(n, bs, w3.full_cm()) == 2 and ctx.max_in_flight(n, bs) == 4
    for name in ("o012_apm", "apm_chain4", "slot2", "full_cm_small_tables", "slot_mix"):   # (slot leaves: pipelined too at this block count — sorted replay)
        dev, orc = pair(oracle, name)
        want, wlens = oracle.encode_blocks(orc(), data, bs, nthreads=8)
        jobs = [ctx.encode_submit(dev(), d_in, bs, *bufs[k]) for k in range(2)]
        for k in (1, 0):   # any order
            ctx.encode_wait(jobs[k])
            d_out, d_lens, d_total = bufs[k]
            assert d_lens.cpu().numpy().astype(np.uint32).tolist() == wlens.tolist(), name
            assert d_out[: int(d_total.item())].cpu().numpy().tobytes() == want.tobytes(), name


def test_cm_unstaged_kernel_still_agrees(ctx, oracle):
    """k_cm (cells in global memory, used beyond 8 slot leaves) against k_cm_staged and the oracle."""
    data = markov_text(20000, seed=15) + mixed_bytes(6000, seed=16)
    ctx.set_variant("cm_unstaged")
    try:
        for name in ("slot_mix", "full_cm_small_tables"):
            check(ctx, oracle, name, data, 4096)
    finally:
        ctx.set_variant()


def test_cm_reference_container(ctx, oracle):
    """w30i + length + ONE stream (main.rs:89-144) with a CM model."""
    d = markov_text(20000, seed=3)
    got = ctx.compress(d, w3.full_cm())
    assert got == oracle.compress(full(oracle, oracle.SlotModel), d)
    assert ctx.decompress(got, w3.full_cm()) == d
