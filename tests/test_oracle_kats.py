"""Pins the CPU oracle to the reference's own known-answer vectors.

Vectors are re-expressed from the reference's unit tests:
  entropy_coding/io.rs:107-227, entropy_coding/arithmetic_coder_tests.rs:44-161,
  docs/state_table/*.csv (sha256), docs/hashslots.md:69-139, bin/cmp/main.rs:14-24.
"""
import hashlib

import numpy as np
import pytest

from tests.synth import lcg_text


# ---------------------------------------------------------------- io.rs tests
def test_read_bits(oracle):  # io.rs:107-125
    r = oracle.ACReader(bytes([0b01010101, 0b10101010]))
    truth = [0, 1] * 4 + [1, 0] * 4
    assert [r.read_bit() for _ in range(16)] == truth
    assert [r.read_bit() for _ in range(16)] == [0] * 16  # past EOF


def test_read_u32(oracle):  # io.rs:127-143
    r = oracle.ACReader(b"\xde\xad\xbe\xef")
    assert r.read_u32() == 0xDEADBEEF
    assert [r.read_bit() for _ in range(16)] == [0] * 16
    r = oracle.ACReader(b"\xde\xad")
    assert r.read_u32() == 0xDEAD0000
    assert [r.read_bit() for _ in range(16)] == [0] * 16


def _writer(oracle, ops, flush):
    w = oracle.ACWriter()
    for op, cnt in ops:
        for _ in range(cnt):
            w.inc_parity() if op == "p" else w.write_bit(op)
    w.flush(flush)
    return w.bytes()


@pytest.mark.parametrize("ops,flush,truth", [
    ([(1, 3), ("p", 3), (0, 5)], 0xFFFFFFFF, bytes([0b11101110, 0b00011111])),   # io.rs:145-155
    ([(1, 3), ("p", 6), (0, 2)], 0xFFFFFFFF, bytes([0b11101111, 0b11011111])),   # io.rs:157-167
    ([(1, 8)], 0xDEADBEEF, bytes([0xFF, 0xDE])),                                  # io.rs:169-177 flush_aligned
    ([(1, 7)], 0x00ADBEEF, bytes([0xFE])),                                        # io.rs:179-187 flush_unaligned
    ([(1, 4), ("p", 2)], 0x00ADBEEF, bytes([0b11110110])),                        # io.rs:189-198
    ([(1, 4), ("p", 10)], 0x00ADBEEF, bytes([0b11110111, 0b11111110])),           # io.rs:200-209
    ([(1, 7), ("p", 7)], 0x00ADBEEF, bytes([0b11111110, 0b11111110])),            # io.rs:211-220
    ([], 0xDEADBEEF, bytes([0xDE])),                                              # io.rs:222-227 flush_only
])
def test_acwriter_vectors(oracle, ops, flush, truth):
    assert _writer(oracle, ops, flush) == truth


# ------------------------------------------------- arithmetic_coder_tests.rs
def _ac_compress(oracle, data, probs):  # arithmetic_coder_tests.rs:8-23
    w = oracle.ACWriter()
    ac = oracle.ArithmeticCoder.new_coder()
    r = oracle.ACReader(data)
    for p in probs:
        ac.encode(r.read_bit(), p, w)
    ac.flush(w)
    return w.bytes()


def _ac_decompress(oracle, comp, probs):  # arithmetic_coder_tests.rs:25-42
    r = oracle.ACReader(comp)
    ac = oracle.ArithmeticCoder.new_decoder(r)
    w = oracle.ACWriter()
    for p in probs:
        w.write_bit(ac.decode(p, r))
    w.flush(0)
    out = w.bytes()
    assert out[-1] == 0  # aligned flush(0) appends exactly one 0x00
    return out[:-1]


@pytest.mark.parametrize("name,data,probs,clen", [
    ("best_model_zeroes", b"\x00" * (1 << 15), [0], 1),
    ("best_model_ones_15", b"\xff" * (1 << 15), [65535], 1),
    ("best_model_ones_16", b"\xff" * (1 << 16), [65535], 2),
    ("best_model_alternating", b"\x55" * 1024, [0, 65535], 1),
    ("worst_model_zeroes", b"\x00" * 16, [65535], 16 * 16 + 1),
    ("worst_model_ones", b"\xff" * 16, [0], 32 * 16 + 1),
    ("worst_model_alternating", b"\x55" * 16, [65535, 0], 24 * 16 + 1),
    ("no_model", b"\xaa\x55" * 64, [1 << 15], 128 + 1),
    ("half_good_model", b"\x55" * 128, [1 << 15, 65535], 64),
    ("half_bad_model", b"\x55" * 128, [1 << 15, 0], 16 * 128 + 64 + 1),
])
def test_ac_lengths_and_roundtrip(oracle, name, data, probs, clen):
    nbits = len(data) * 8
    pr = (probs * (nbits // len(probs) + 1))[:nbits]
    comp = _ac_compress(oracle, data, pr)
    assert len(comp) == clen, name
    assert _ac_decompress(oracle, comp, pr) == data


# ---------------------------------------------------------------- state table
ST_SHA = "7adb99832083a81c80fe60a092f748aca8e8e8ed4a4b750a63d58b6b2a045e3e"
AUX_SHA = "8340ec4895b39de9c6fa67faa5b190fe1d68b872ee809c248f3350bd47d37bd2"


def test_state_table_csv_digest(oracle):
    assert hashlib.sha256(oracle.state_table_csv().encode()).hexdigest() == ST_SHA
    assert hashlib.sha256(oracle.state_table_csv(aux=True).encode()).hexdigest() == AUX_SHA


def test_state_table_spot_rows(oracle):
    t = oracle.state_table()  # prob, next0, next1
    rows = {1: (3, 993, 32768), 3: (4, 995, 32768), 4: (6, 997, 21845), 3962: (2235, 3962, 64079)}
    for s, (n0, n1, p) in rows.items():
        assert (int(t[s][1]), int(t[s][2]), int(t[s][0])) == (n0, n1, p)
    assert t[:, 0].min() == 1456 and t[:, 0].max() == 64079
    assert t[:, 1:].max() < 3963
    s = [0, 1, 3, 4]
    out = (oracle.C.c_uint16 * 4)()
    oracle.lib.w3o_st_next4((oracle.C.c_uint16 * 4)(*s), 0b1010, out)
    assert list(out) == [oracle.lib.w3o_st_next(0, 1), oracle.lib.w3o_st_next(1, 0),
                         oracle.lib.w3o_st_next(3, 1), oracle.lib.w3o_st_next(4, 0)]


# ---------------------------------------------------------------- hashslots.md
def test_slot_index_table(oracle):
    """docs/hashslots.md:61-133: rel_idx=(1<<bit_id)-1+nib_ctx, abs=((rel+15*id)*3)>>1, odd->mask even->shift."""
    C = oracle.C
    spot = {(0, 0, 0): (0, 0), (0, 1, 1): (3, 0), (0, 3, 7): (21, 0), (1, 0, 0): (22, 1), (1, 3, 7): (43, 1),
            (2, 0, 0): (45, 0), (2, 3, 7): (66, 0), (3, 0, 0): (67, 1), (3, 2, 1): (73, 1), (3, 3, 7): (88, 1)}
    seen = 0
    for sid in range(4):
        for bit_id in range(4):
            for nib_ctx in range(1 << bit_id):
                a, p = C.c_uint32(), C.c_int()
                oracle.lib.w3o_slot_get_idx(sid, bit_id, nib_ctx, C.byref(a), C.byref(p))
                rel = (1 << bit_id) - 1 + nib_ctx
                assert a.value == ((rel + sid * 15) * 3) >> 1
                assert p.value == ((rel + sid * 15) * 3) & 1
                if (sid, bit_id, nib_ctx) in spot:
                    assert (a.value, p.value) == spot[(sid, bit_id, nib_ctx)]
                seen += 1
    assert seen == 60


def test_cell_slot_codec(oracle):
    C = oracle.C
    cell = oracle.Cell()
    rng = np.random.default_rng(3)
    want = {}
    for sid in range(4):
        for bit_id in range(4):
            for nib_ctx in range(1 << bit_id):
                v = int(rng.integers(0, 4096))
                want[(sid, bit_id, nib_ctx)] = v
                oracle.lib.w3o_slot_set_state(C.byref(cell), sid, bit_id, nib_ctx, v)
    for k, v in want.items():  # neighbours preserved (hashmap.rs:99-112)
        assert oracle.lib.w3o_slot_get_state(C.byref(cell), *k) == v
    # slot byte ranges, hashslots.md:136-139
    cell2 = oracle.Cell()
    for bit_id in range(4):
        for nib_ctx in range(1 << bit_id):
            oracle.lib.w3o_slot_set_state(C.byref(cell2), 1, bit_id, nib_ctx, 0xFFF)
    b = bytes(cell2.slots)
    assert b[:22] == bytes(22) and b[22] == 0x0F and b[23:45] == b"\xff" * 22 and b[45:] == bytes(45)
    # tags: big-endian packed, id 3 = lowest 12 bits; miss -> slot 1 (hashmap.rs:43-68)
    cell.hashes[:] = (C.c_uint8 * 6)(0xAB, 0xC1, 0x23, 0x45, 0x67, 0x89)
    assert oracle.lib.w3o_cell_get_slot(C.byref(cell), 0x789) == 3
    assert oracle.lib.w3o_cell_get_slot(C.byref(cell), 0x456) == 2
    assert oracle.lib.w3o_cell_get_slot(C.byref(cell), 0xFFFF123) == 1
    assert oracle.lib.w3o_cell_get_slot(C.byref(cell), 0xABC) == 0
    assert oracle.lib.w3o_cell_get_slot(C.byref(cell), 0x000) == 1
    assert oracle.lib.w3o_hashmap_log_cell_count(1 << 30) == 23  # floor(30 - log2(96))
    assert oracle.lib.w3o_hashmap_cell_index(0xF000000000000000, 4) == 15
    nib = 0b1011
    st = (C.c_uint16 * 4)(1, 2, 3, 4)
    oracle.lib.w3o_slot_set_nib(C.byref(cell), 2, nib, st)
    out = (C.c_uint16 * 4)()
    oracle.lib.w3o_slot_get_nib(C.byref(cell), 2, nib, out)
    assert list(out) == [1, 2, 3, 4]
    assert oracle.lib.w3o_slot_get_state(C.byref(cell), 2, 3, nib >> 1) == 4


# ------------------------------------------------- models (cross-check digests)
def test_counter(oracle):
    m = oracle.Order0()
    assert m.predict() == 32768
    # all-zero 64 KiB block triggers the halving at count 65535 (counter.rs:22-25)
    assert oracle.encode_stream(oracle.Order0(), bytes(65536)) == b"\xff" * 16
    assert oracle.encode_stream(oracle.Order0(), b"\xff" * 65536) == b"\x00" * 16 + b"\x01"


def test_survey_cross_check_digests(oracle):
    d = lcg_text(65536)
    assert hashlib.sha256(d).hexdigest() == "87a3c717b538e0b5a76a8c07b5543ef5a129ffc3c48d78bcc1d4cbeb1123b169"
    cases = [
        (oracle.Order0(), 43693, "37791f2604aaa6d6a81de79dbeb80ef09e25581d28c95f5cd5b411848aa7c64d"),
        (oracle.Order1(), 49588, "5bceb3081cb21048473d483534b6f1cc901e1ac04fb1858ea303c994cf1fb7dc"),
        (oracle.BestOfTwoModel(oracle.Order0(), oracle.Order1()), 45259,
         "19da6b5f594baf1395037f6ea2c497a8a0740806e3ceede028d20e7b6c09b518"),
    ]
    for m, n, sha in cases:
        s = oracle.encode_stream(m, d)
        assert len(s) == n and hashlib.sha256(s).hexdigest() == sha
    assert oracle.encode_stream(oracle.Order0(), d[:16]).hex() == "9196f5dfb6ca8dcfbf0bbba9e4cf8adc3f"


def test_order0_equiv_ordern_11_3(oracle):  # bin/cmp/main.rs:14-24
    d = lcg_text(20000, seed=99)
    assert oracle.encode_stream(oracle.Order0(), d) == oracle.encode_stream(oracle.OrderN(11, 3), d)
    assert oracle.encode_stream(oracle.Order1(), d) == oracle.encode_stream(oracle.OrderN(19, 3), d)
    assert oracle.encode_stream(oracle.OrderN(16, 3), d) == oracle.encode_stream(
        oracle.OrderNEntropy(16, 3, oracle.RawHistory()), d)


def test_stationary_and_ac_history(oracle):
    sm = oracle.StationaryModel(b"\x80" * 100)  # bit position 0 is always 1, others 0
    t = sm.table
    assert t[0] > 60000 and all(x < 2000 for x in t[1:])
    assert oracle.StationaryModel.for_book1().table == [1, 50188, 62497, 15819, 22545, 31499, 22988, 29616]
    h = oracle.ACHistory(8, oracle.StationaryModel.for_book1())
    assert h.hash() == 255  # all-zero history: 0 takes the UPPER sub-interval (arithmetic_coder.rs:45-48) -> eight 1 bits
    vals = set()
    for b in lcg_text(64, seed=5):
        for s in range(7, -1, -1):
            h.update((b >> s) & 1)
            v = h.hash()
            assert 0 <= v < 256
            vals.add(v)
    assert len(vals) > 8


def test_roundtrips_all_models(oracle):
    d = lcg_text(3000, seed=42) + bytes(100) + b"\xff" * 50
    mk = [
        lambda: oracle.Order0(), lambda: oracle.Order1(), lambda: oracle.OrderN(27, 3), lambda: oracle.OrderN(12, 0),
        lambda: oracle.OrderN(14, 4), lambda: oracle.OrderNEntropy(11, 3, oracle.ACHistory(8, oracle.StationaryModel.for_book1())),
        lambda: oracle.FrozenModel(oracle.Order0()),
        lambda: oracle.BestOfTwoModel(oracle.BestOfTwoModel(oracle.Order0(), oracle.Order1()), oracle.OrderN(27, 3)),
    ]
    for f in mk:
        s = oracle.encode_stream(f(), d)
        assert oracle.decode_stream(f(), s, len(d)) == d
        c = oracle.compress(f(), d)
        assert c[:4] == b"w30i" and int.from_bytes(c[4:12], "big") == len(d) and c[12:] == s
        assert oracle.decompress(f(), c) == d
    # frozen model never trains: every p is 32768 -> 1 bit per bit + flush byte
    assert len(oracle.encode_stream(oracle.FrozenModel(oracle.Order0()), d)) == len(d) + 1
    with pytest.raises(AssertionError):
        oracle.decompress(oracle.Order0(), b"w31i" + bytes(20))


def test_stats_sink_matches_byte_stream(oracle):  # helpers.rs:60-90: csize = bits/8, flush adds nothing
    d = lcg_text(5000, seed=8)
    n = oracle.encode_stats(oracle.Order0(), d)
    s = oracle.encode_stream(oracle.Order0(), d)
    assert n <= len(s) <= n + 9


def test_block_mode(oracle):
    d = lcg_text(10000, seed=3)
    bs = 4096
    out, lens = oracle.encode_blocks(oracle.Order0(), d, bs, nthreads=3)
    assert len(lens) == 3 and lens.sum() == len(out)
    off = 0
    for b in range(3):
        blk = d[b * bs:(b + 1) * bs]
        assert out[off:off + lens[b]].tobytes() == oracle.encode_stream(oracle.Order0(), blk)
        off += lens[b]
    back = oracle.decode_blocks(oracle.Order0(), out, lens, bs, len(d), nthreads=2)
    assert back.tobytes() == d
    out0, lens0 = oracle.encode_blocks(oracle.Order0(), b"", bs)
    assert len(out0) == 0 and len(lens0) == 0
