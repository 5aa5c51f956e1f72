"""CPU-side checks of the BUILD-DEFINED CM pieces (SURVEY §8 A15/A16/A19 ii-v): the product's host-side
tables against the reference's CSV digests and the oracle, the oracle's slot model / APM against small
hand-computed cases and committed golden digests, spec validation of the new nodes.  No GPU."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import weath3rb0i_amd as w3
from weath3rb0i_amd import _lib as L
from tests.synth import lcg_text, markov_text, mixed_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# sha256 of docs/state_table/state_table.csv in the reference
STATE_TABLE_CSV_SHA256 = "7adb99832083a81c80fe60a092f748aca8e8e8ed4a4b750a63d58b6b2a045e3e"


def test_product_state_table_matches_reference_csv_digest():
    rows = w3.StateTable.rows()
    csv = "state,tr0,tr1,prob\n" + "".join("%d,%d,%d,%d\n" % (i, r[1], r[2], r[0]) for i, r in enumerate(rows))
    assert hashlib.sha256(csv.encode()).hexdigest() == STATE_TABLE_CSV_SHA256
    # spot rows quoted in SURVEY §8(c)3
    assert rows[1].tolist() == [32768, 3, 993] and rows[4].tolist() == [21845, 6, 997] and rows[3962].tolist() == [64079, 2235, 3962]


def test_state_table_trait_surface(oracle):
    for s in (0, 1, 2, 3, 500, 3962):
        for bit in (0, 1):
            assert w3.StateTable.next(s, bit) == oracle.lib.w3o_st_next(s, bit)
        assert w3.StateTable.p(s) == oracle.lib.w3o_st_p(s)
    assert w3.StateTable.next4([0, 1, 2, 3], 0b1010) == [2, 3, 2973, 4]
    assert w3.StateTable.p4([0, 4, 3962, 3]) == [32768, 21845, 64079, 32768]


def test_stretch_squash_product_equals_oracle(oracle):
    st = np.zeros(4096, dtype=np.int16)
    sq = np.zeros(4095, dtype=np.uint16)
    assert L.load().w3_stretch_squash(st.ctypes.data_as(C.c_void_p), sq.ctypes.data_as(C.c_void_p)) == 0
    assert [oracle.squash(d) for d in range(-2047, 2048)] == sq.tolist()
    assert [oracle.stretch(q << 4) for q in range(4096)] == st.tolist()
    # shape: logistic in 1/256 nat, symmetric, monotone, inverse pair
    assert sq[2047] == 32768 and sq[0] == 22 and sq[-1] == 65514
    assert all(int(sq[2047 + d]) + int(sq[2047 - d]) == 65536 for d in range(1, 2048))
    assert np.all(np.diff(sq.astype(np.int64)) >= 0) and np.all(np.diff(st.astype(np.int64)) >= 0)
    for d in (-2000, -300, -1, 0, 1, 77, 1500):
        assert abs(int(st[int(sq[d + 2047]) >> 4]) - d) <= 8 + abs(d) // 16
    import math
    assert max(abs(int(sq[d + 2047]) - 65536 / (1 + math.exp(-d / 256))) for d in range(-2047, 2048)) <= 1.0


def test_hashmap_sizing_rule(oracle):
    for size in (96, 97, 1 << 20, 1536 << 10, (1 << 24) - 1, 1 << 24, 3 << 30):
        assert w3.HashMap.new(size).log_cell_count == oracle.lib.w3o_hashmap_log_cell_count(size)


def test_slot_model_first_steps_by_hand(oracle):
    """Empty table: every state is 0 -> p = 32768 and the state walks the entry nodes (naive.rs:25-27)."""
    m = oracle.SlotModel(2, 10)
    seen = []
    for bit in (1, 0, 1, 1):
        seen.append(m.predict())
        m.update(bit)
    assert seen == [32768] * 4
    # the same nibble context again (second byte starts with the same two-byte history? no: order 2 sees the
    # first byte now) -> fresh slot, still 32768; an order-0 model returns to the SAME slots every byte
    m0 = oracle.SlotModel(0, 4)
    ps = []
    for _ in range(3):
        for bit in (0, 1, 1, 0, 0, 0, 0, 1):
            ps.append(m0.predict())
            m0.update(bit)
    assert ps[:8] == [32768] * 8
    st = w3.StateTable
    # second visit: first-bit state after one 0 is state 1 (p = 32768); third visit: next(1, 0) = 3 -> p(3)
    assert ps[8] == st.p(1) and ps[16] == st.p(st.next(1, 0))
    # second bit of the first nibble was reached through nib_ctx = 0 and saw a 1: state 2, then next(2, 1)
    assert ps[9] == st.p(2) and ps[17] == st.p(st.next(2, 1))


def test_slot_replacement_policy(oracle):
    """Tag miss: candidates 1,0,2,3 by least observations of the first-bit state; tag stored; states cleared."""
    import ctypes
    cell = oracle.Cell()
    assert oracle.lib.w3o_cell_get_slot(C.byref(cell), 0x0000) == 3           # empty cell: tag 0 hits id 3
    assert oracle.lib.w3o_cell_get_slot(C.byref(cell), 0x0ABC) == 1           # the reference's PoC answer on a miss
    # a model on a 2-cell table must evict: after many distinct contexts every slot is in use and predictions stay valid
    m = oracle.SlotModel(3, 1)
    data = markov_text(4096, 5)
    c = oracle.encode_stream(m, data)
    back = oracle.decode_stream(oracle.SlotModel(3, 1), c, len(data))
    assert bytes(back) == bytes(data)
    assert [oracle.st_conf(s) for s in (0, 1, 2, 3, 4, 993, 3962)] == [0, 1, 1, 2, 3, 2, 45]


def test_apm_identity_at_start_and_learning(oracle):
    """A fresh APM row is the identity map (within interpolation error); it then moves towards the coded bits."""
    m = oracle.APM(oracle.FrozenModel(oracle.Order0()), oracle.APM_ORDER0, 4)   # input p is always 32768
    p0 = m.predict()
    assert abs(p0 - 32768) <= 64      # stretch is quantised to p >> 4, buckets are 128 apart
    for _ in range(40):                     # 5 zero bytes: row c0=1 sees five 0s
        m.update(0)
    assert m.predict() < p0
    # chain of two
    m2 = oracle.APM(oracle.APM(oracle.Order0()), oracle.APM_ORDER1, 6)
    data = markov_text(3000, 2)
    assert bytes(oracle.decode_stream(oracle.APM(oracle.APM(oracle.Order0()), oracle.APM_ORDER1, 6), oracle.encode_stream(m2, data), len(data))) == bytes(data)


def cm_model(oracle, name):
    o = oracle
    if name == "slot2":
        return o.SlotModel(2, 12)
    if name == "apm0_order0":
        return o.APM(o.Order0())
    if name == "o012_apm":
        return o.APM(o.BestOfTwoModel(o.BestOfTwoModel(o.Order0(), o.Order1()), o.OrderN(27, 3)))
    if name == "full_cm":
        m = o.BestOfTwoModel(o.BestOfTwoModel(o.Order0(), o.Order1()), o.OrderN(27, 3))
        for order in (1, 2, 3, 4):
            m = o.BestOfTwoModel(m, o.SlotModel(order, 14))
        return o.APM(o.APM(m, o.APM_ORDER0, 7), o.APM_ORDER1, 6)
    raise KeyError(name)


GOLDEN = os.path.join(ROOT, "tests", "golden", "cm_streams.json")


def golden_inputs():
    return {"lcg_text_20000_s12345": bytes(lcg_text(20000, 12345)), "markov_16384_s3": bytes(markov_text(16384, 3)),
            "mixed_12000_s9": bytes(mixed_bytes(12000, 9)), "zeros_5000": bytes(5000), "ff_3000": b"\xff" * 3000}


def test_cm_golden_digests(oracle):
    """Committed digests of the build-defined streams (made by tests/golden/make_cm_golden.py from the oracle):
    any change to the definitions is a format break and must be deliberate."""
    want = json.load(open(GOLDEN))
    for iname, data in golden_inputs().items():
        for mname in ("slot2", "apm0_order0", "o012_apm", "full_cm"):
            c = oracle.encode_stream(cm_model(oracle, mname), data)
            assert [len(c), hashlib.sha256(bytes(c)).hexdigest()] == want[iname][mname], (iname, mname)


def test_spec_validation_of_cm_nodes():
    assert w3.full_cm().spec().n_nodes == 3 + 2 + 4 * 2 + 2
    assert w3.APM(w3.SlotModel(2, w3.HashMap.new(1536 << 10))).spec().n_nodes == 2
    for bad in (lambda: w3.SlotModel(8, 12), lambda: w3.SlotModel(1, 0), lambda: w3.SlotModel(1, 25), lambda: w3.APM(w3.Order0(), 2, 7),
                lambda: w3.APM(w3.Order0(), 0, 0), lambda: w3.APM(w3.Order0(), 0, 16)):
        with pytest.raises(w3.W3Error) as e:
            bad().spec()
        assert e.value.code == L.W3_E_INVALID
    # APM below a BestOfTwo is well-formed but only the root chain is implemented on the device
    with pytest.raises(w3.W3Error) as e:
        w3.BestOfTwoModel(w3.APM(w3.Order0()), w3.Order1()).spec()
    assert e.value.code == L.W3_E_UNSUPPORTED
    with pytest.raises(TypeError):
        w3.FrozenModel(w3.SlotModel(1, 10))
