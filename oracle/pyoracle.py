"""ctypes binding of the CPU ORACLE (oracle/libw3oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (weath3rb0i_amd) never
imports this module.  Names mirror the reference crate (src/models/*.rs etc.)
so the parity tests read like the reference's own.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libw3oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    so = os.environ.get("W3_ORACLE_SO") or _SO   # (bench.py points this at the native build before the first import)
    if so == _SO and not os.path.exists(_SO):
        build()
    lib = C.CDLL(so)
    vp, u8p, sz = C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t
    lib.w3o_order0.restype = vp
    lib.w3o_order1.restype = vp
    lib.w3o_ordern.restype = vp
    lib.w3o_ordern.argtypes = [C.c_uint8, C.c_uint8]
    lib.w3o_ordern_entropy.restype = vp
    lib.w3o_ordern_entropy.argtypes = [C.c_uint8, C.c_uint8, vp]
    lib.w3o_frozen.restype = vp
    lib.w3o_frozen.argtypes = [vp]
    lib.w3o_best_of_two.restype = vp
    lib.w3o_best_of_two.argtypes = [vp, vp]
    lib.w3o_model_clone_fresh.restype = vp
    lib.w3o_model_clone_fresh.argtypes = [vp]
    lib.w3o_model_reset.argtypes = [vp]
    lib.w3o_model_free.argtypes = [vp]
    lib.w3o_model_predict.restype = C.c_uint16
    lib.w3o_model_predict.argtypes = [vp]
    lib.w3o_model_update.argtypes = [vp, C.c_uint8]
    lib.w3o_opinion_mix.restype = C.c_uint16
    lib.w3o_opinion_mix.argtypes = [C.c_uint16, C.c_uint16]
    lib.w3o_encode_stream.restype = C.POINTER(C.c_uint8)
    lib.w3o_encode_stream.argtypes = [vp, vp, sz, C.POINTER(sz)]
    lib.w3o_encode_stats.restype = C.c_uint64
    lib.w3o_encode_stats.argtypes = [vp, vp, sz]
    lib.w3o_encode_stats_bits.restype = C.c_uint64
    lib.w3o_encode_stats_bits.argtypes = [vp, vp, sz]
    lib.w3o_decode_stream.argtypes = [vp, vp, sz, vp, sz]
    lib.w3o_predict_all.argtypes = [vp, vp, sz, vp]
    lib.w3o_compress_container.restype = C.POINTER(C.c_uint8)
    lib.w3o_compress_container.argtypes = [vp, vp, sz, C.POINTER(sz)]
    lib.w3o_decompress_container.argtypes = [vp, vp, sz, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz)]
    lib.w3o_encode_blocks.argtypes = [vp, vp, sz, sz, vp, sz, C.POINTER(sz), vp, C.c_int]
    lib.w3o_decode_blocks.argtypes = [vp, vp, vp, sz, sz, C.c_uint64, vp, C.c_int]
    lib.w3o_state_table.restype = vp
    lib.w3o_state_table_aux.restype = vp
    lib.w3o_st_next.restype = C.c_uint16
    lib.w3o_st_next.argtypes = [C.c_uint16, C.c_uint8]
    lib.w3o_st_p.restype = C.c_uint16
    lib.w3o_st_p.argtypes = [C.c_uint16]
    lib.w3o_hashmap_log_cell_count.restype = C.c_uint32
    lib.w3o_hashmap_log_cell_count.argtypes = [sz]
    lib.w3o_hashmap_cell_index.restype = C.c_uint64
    lib.w3o_hashmap_cell_index.argtypes = [C.c_uint64, C.c_uint32]
    lib.w3o_cell_get_slot.restype = C.c_uint8
    lib.w3o_cell_get_slot.argtypes = [vp, C.c_uint64]
    lib.w3o_slot_get_idx.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    lib.w3o_slot_get_state.restype = C.c_uint16
    lib.w3o_slot_get_state.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8]
    lib.w3o_slot_set_state.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16]
    lib.w3o_slot_get_nib.argtypes = [vp, C.c_uint8, C.c_uint8, vp]
    lib.w3o_slot_set_nib.argtypes = [vp, C.c_uint8, C.c_uint8, vp]
    lib.w3o_slot_model.restype = vp
    lib.w3o_slot_model.argtypes = [C.c_uint8, C.c_uint8]
    lib.w3o_apm.restype = vp
    lib.w3o_apm.argtypes = [vp, C.c_uint8, C.c_uint8]
    lib.w3o_squash.restype = C.c_uint16
    lib.w3o_squash.argtypes = [C.c_int]
    lib.w3o_stretch.restype = C.c_int
    lib.w3o_stretch.argtypes = [C.c_uint16]
    lib.w3o_slot_hash.restype = C.c_uint64
    lib.w3o_slot_hash.argtypes = [C.c_uint8, C.c_uint64, C.c_int, C.c_uint32]
    lib.w3o_st_conf.restype = C.c_uint8
    lib.w3o_st_conf.argtypes = [C.c_uint16]
    lib.free = C.CDLL(None).free
    lib.free.argtypes = [vp]
    return lib


lib = _load()


# ---- C structs --------------------------------------------------------------
class Sink(C.Structure):
    _fields_ = [("kind", C.c_int), ("buf", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t),
                ("owns", C.c_int), ("acc", C.c_uint8), ("idx", C.c_uint8), ("rev_bits", C.c_uint64),
                ("bit_count", C.c_uint64), ("state", C.c_uint32), ("max_bits", C.c_uint8), ("eidx", C.c_uint8),
                ("erev", C.c_uint16)]


class Reader(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("len", C.c_size_t), ("pos", C.c_size_t), ("cur", C.c_uint8), ("mask", C.c_uint8)]


class AC(C.Structure):
    _fields_ = [("x1", C.c_uint32), ("x2", C.c_uint32), ("x", C.c_uint32)]


class Stationary(C.Structure):
    _fields_ = [("table", C.c_uint16 * 8), ("alignment", C.c_uint8)]


class HuffTables(C.Structure):
    _fields_ = [("code", C.c_uint16 * 256), ("len", C.c_uint8 * 256), ("rem_code", C.c_uint16 * 256), ("rem_len", C.c_uint8 * 256)]


class History(C.Structure):
    _fields_ = [("kind", C.c_int), ("raw_bits", C.c_uint32), ("pos", C.c_uint64), ("bits", C.c_uint64),
                ("max_bits", C.c_uint8), ("model", Stationary), ("compressed_bits", C.c_uint32), ("huff", HuffTables)]


class Cell(C.Structure):
    _fields_ = [("hashes", C.c_uint8 * 6), ("slots", C.c_uint8 * 90)]


def _buf(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


# ---- entropy_coding ---------------------------------------------------------
class ACWriter:
    """ACWriter<Vec<u8>> (entropy_coding/io.rs:52-101)."""

    def __init__(self):
        self.s = Sink()
        lib.w3o_sink_init_bytes(C.byref(self.s))

    def write_bit(self, bit):
        lib.w3o_sink_write_bit(C.byref(self.s), C.c_uint8(bit))

    def inc_parity(self):
        lib.w3o_sink_inc_parity(C.byref(self.s))

    def flush(self, state):
        lib.w3o_sink_flush(C.byref(self.s), C.c_uint32(state))

    def bytes(self):
        return bytes(self.s.buf[: self.s.len])

    def __del__(self):
        lib.w3o_sink_free(C.byref(self.s))


class ACStats(ACWriter):
    """helpers.rs:60-90"""

    def __init__(self):
        self.s = Sink()
        lib.w3o_sink_init_stats(C.byref(self.s))

    def result(self):
        return lib.w3o_stats_result(C.byref(self.s))


lib.w3o_stats_result.restype = C.c_uint64


class ACReader:
    """entropy_coding/io.rs:7-49"""

    def __init__(self, data):
        self._a, p = _buf(data)
        self.r = Reader()
        lib.w3o_reader_init(C.byref(self.r), p, C.c_size_t(len(self._a)))

    def read_bit(self):
        return lib.w3o_reader_read_bit(C.byref(self.r))

    def read_u32(self):
        return lib.w3o_reader_read_u32(C.byref(self.r))


lib.w3o_reader_read_bit.restype = C.c_uint8
lib.w3o_reader_read_u32.restype = C.c_uint32
lib.w3o_ac_decode.restype = C.c_uint8


class ArithmeticCoder:
    """entropy_coding/arithmetic_coder.rs:13-106"""

    def __init__(self):
        self.ac = AC()

    @classmethod
    def new_coder(cls):
        o = cls()
        lib.w3o_ac_new_coder(C.byref(o.ac))
        return o

    @classmethod
    def new_decoder(cls, reader):
        o = cls()
        lib.w3o_ac_new_decoder(C.byref(o.ac), C.byref(reader.r))
        return o

    def encode(self, bit, prob, io):
        return lib.w3o_ac_encode(C.byref(self.ac), C.c_uint8(bit), C.c_uint16(prob), C.byref(io.s))

    def flush(self, io):
        return lib.w3o_ac_flush(C.byref(self.ac), C.byref(io.s))

    def decode(self, prob, reader):
        return lib.w3o_ac_decode(C.byref(self.ac), C.c_uint16(prob), C.byref(reader.r))


# ---- models -----------------------------------------------------------------
class StationaryModel:
    def __init__(self, buf=None, table=None):
        self.m = Stationary()
        if table is not None:
            lib.w3o_stationary_from_table(C.byref(self.m), (C.c_uint16 * 8)(*table))
        else:
            a, p = _buf(buf)
            lib.w3o_stationary_new(C.byref(self.m), p, C.c_size_t(len(a)))

    @classmethod
    def from_table(cls, table):
        return cls(table=table)

    @classmethod
    def for_book1(cls):
        o = cls(table=[0] * 8)
        lib.w3o_stationary_for_book1(C.byref(o.m))
        return o

    @classmethod
    def for_enwik7(cls):
        o = cls(table=[0] * 8)
        lib.w3o_stationary_for_enwik7(C.byref(o.m))
        return o

    @property
    def table(self):
        return list(self.m.table)


class RawHistory:
    def __init__(self):
        self.h = History()
        lib.w3o_history_raw(C.byref(self.h))


class ACHistory:
    def __init__(self, max_bits, model):
        self.h = History()
        lib.w3o_history_ac(C.byref(self.h), C.c_uint8(max_bits), C.byref(model.m))

    def update(self, bit):
        lib.w3o_history_update(C.byref(self.h), C.c_uint8(bit))

    def hash(self):
        return lib.w3o_history_hash(C.byref(self.h))


lib.w3o_history_hash.restype = C.c_uint32


class ACHistoryCached:
    """history/ac_history_cached.rs:9-76 restated with its memo (w3o_achc_*).  counts() -> (level-0 hits, level-1 hits, full runs,
    memo entries)."""

    def __init__(self, max_bits, model, cache_size):
        self.h = lib.w3o_achc_new(C.c_uint8(max_bits), C.byref(model.m), C.c_uint8(cache_size))

    def update(self, bit):
        lib.w3o_achc_update(self.h, C.c_uint8(bit))

    def hash(self):
        return lib.w3o_achc_hash(self.h)

    def counts(self):
        out = (C.c_uint64 * 4)()
        lib.w3o_achc_counts(self.h, out)
        return tuple(int(v) for v in out)

    def __del__(self):
        if getattr(self, "h", None):
            lib.w3o_achc_free(self.h)
            self.h = None


lib.w3o_achc_new.restype = C.c_void_p
lib.w3o_achc_new.argtypes = [C.c_uint8, C.c_void_p, C.c_uint8]
lib.w3o_achc_free.argtypes = [C.c_void_p]
lib.w3o_achc_update.argtypes = [C.c_void_p, C.c_uint8]
lib.w3o_achc_hash.restype = C.c_uint32
lib.w3o_achc_hash.argtypes = [C.c_void_p]
lib.w3o_achc_counts.argtypes = [C.c_void_p, C.c_void_p]


# ---- length-limited Huffman (entropy_coding/package_merge.rs) and HuffHistory (history/huff_history.rs) -------------
PM_ERRORS = {-1: "No symbols provided", -2: "Max length is too big", -3: "Max length is too small"}


def package_merge(counts, max_len):
    """package_merge.rs:1-29 (ties sorted stably; see w3_oracle.h).  Raises AssertionError with the reference's panic text."""
    c = (C.c_uint32 * max(len(counts), 1))(*counts)
    out = (C.c_uint8 * max(len(counts), 1))()
    rc = lib.w3o_package_merge(c, len(counts), C.c_uint8(max_len), out)
    if rc:
        raise AssertionError(PM_ERRORS[rc])
    return list(out[: len(counts)])


def canonical(code_lens):
    """package_merge.rs:87-117 -> [(code, len)]"""
    n = len(code_lens)
    cl = (C.c_uint8 * max(n, 1))(*code_lens)
    codes = (C.c_uint16 * max(n, 1))()
    lens = (C.c_uint8 * max(n, 1))()
    lib.w3o_canonical(cl, n, codes, lens)
    return [(codes[i], lens[i]) for i in range(n)]


def huff_tables(buf, huff_size, rem_huff_size):
    """HuffHistory::new's two tables (huff_history.rs:17-43) -> HuffTables"""
    a, p = _buf(buf)
    t = HuffTables()
    rc = lib.w3o_huff_tables_new(p, len(a), C.c_uint8(huff_size), C.c_uint8(rem_huff_size), C.byref(t))
    if rc:
        raise AssertionError(PM_ERRORS[rc])
    return t


class HuffHistory:
    """HuffHistory::new(buf, huff_size, rem_huff_size) (history/huff_history.rs:17-55), or from ready tables."""

    def __init__(self, buf=None, huff_size=12, rem_huff_size=12, tables=None):
        self.tables = tables if tables is not None else huff_tables(buf, huff_size, rem_huff_size)
        self.h = History()
        lib.w3o_history_huff(C.byref(self.h), C.byref(self.tables))

    def update(self, bit):
        lib.w3o_history_update(C.byref(self.h), C.c_uint8(bit))

    def hash(self):
        return lib.w3o_history_hash(C.byref(self.h))


class Model:
    """Owns a w3o_model tree; children handed to a composite are consumed."""

    def __init__(self, ptr):
        self.ptr = ptr
        self.owned = True

    def _take(self):
        assert self.owned, "model already moved into a composite"
        self.owned = False
        return self.ptr

    def predict(self):
        return lib.w3o_model_predict(self.ptr)

    def update(self, bit):
        lib.w3o_model_update(self.ptr, C.c_uint8(bit))

    def reset(self):
        lib.w3o_model_reset(self.ptr)

    def __del__(self):
        if getattr(self, "owned", False) and self.ptr:
            lib.w3o_model_free(self.ptr)


def Order0():
    return Model(lib.w3o_order0())


def Order1():
    return Model(lib.w3o_order1())


def OrderN(bits, align):
    return Model(lib.w3o_ordern(bits, align))


def OrderNEntropy(bits, align, history):
    return Model(lib.w3o_ordern_entropy(bits, align, C.byref(history.h)))


def FrozenModel(model):
    return Model(lib.w3o_frozen(model._take()))


def BestOfTwoModel(m1, m2):
    return Model(lib.w3o_best_of_two(m1._take(), m2._take()))


# ---- build-defined models (SURVEY §8 A19 ii-v) -----------------
APM_ORDER0, APM_ORDER1 = 0, 1


def SlotModel(order, log_cells):
    assert 0 <= order <= 7 and 1 <= log_cells <= 24
    return Model(lib.w3o_slot_model(order, log_cells))


def APM(model, ctx=APM_ORDER0, rate=7):
    assert ctx in (0, 1) and 1 <= rate <= 15
    return Model(lib.w3o_apm(model._take(), ctx, rate))


def squash(d):
    return lib.w3o_squash(d)


def stretch(p):
    return lib.w3o_stretch(p)


def slot_hash(order, hist_bytes, second, hi_nib):
    return lib.w3o_slot_hash(order, hist_bytes, int(second), hi_nib)


def st_conf(state):
    return lib.w3o_st_conf(state)


def opinion_mix(p1, p2):
    return lib.w3o_opinion_mix(p1, p2)


# ---- loops ------------------------------------------------------------------
def encode_stream(model, data):
    """main.rs:103-111 without the 12-byte header."""
    a, p = _buf(data)
    n = C.c_size_t()
    ptr = lib.w3o_encode_stream(model.ptr, p, len(a), C.byref(n))
    out = bytes(ptr[: n.value])
    lib.free(ptr)
    return out


def encode_stats(model, data):
    a, p = _buf(data)
    return lib.w3o_encode_stats(model.ptr, p, len(a))


def encode_stats_bits(model, data):
    """ACStats::bit_count (helpers.rs:62): csize = bits // 8."""
    a, p = _buf(data)
    return lib.w3o_encode_stats_bits(model.ptr, p, len(a))


def decode_stream(model, data, n):
    a, p = _buf(data)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    lib.w3o_decode_stream(model.ptr, p, len(a), out.ctypes.data_as(C.c_void_p), n)
    return out[:n].tobytes()


def predict_all(model, data):
    a, p = _buf(data)
    out = np.zeros(len(a) * 8, dtype=np.uint16)
    lib.w3o_predict_all(model.ptr, p, len(a), out.ctypes.data_as(C.c_void_p))
    return out


def compress(model, data):
    """Reference container: b'w30i' + u64 BE len + stream (main.rs:89-113)."""
    a, p = _buf(data)
    n = C.c_size_t()
    ptr = lib.w3o_compress_container(model.ptr, p, len(a), C.byref(n))
    out = bytes(ptr[: n.value])
    lib.free(ptr)
    return out


def decompress(model, data):
    a, p = _buf(data)
    optr = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    rc = lib.w3o_decompress_container(model.ptr, p, len(a), C.byref(optr), C.byref(n))
    if rc == -2:
        raise AssertionError("Magic numbers don't match up")
    if rc:
        raise IOError("unexpected EOF reading header")
    out = bytes(optr[: n.value])
    lib.free(optr)
    return out


def encode_blocks(model, data, block_size, nthreads=1):
    a, p = _buf(data)
    n = len(a)
    nb = (n + block_size - 1) // block_size
    cap = 2 * n + 64 * (nb + 1)
    while True:
        out = np.zeros(cap, dtype=np.uint8)
        lens = np.zeros(max(nb, 1), dtype=np.uint32)
        tot = C.c_size_t()
        rc = lib.w3o_encode_blocks(model.ptr, p, n, block_size, out.ctypes.data_as(C.c_void_p), cap, C.byref(tot),
                                   lens.ctypes.data_as(C.c_void_p), nthreads)
        if rc == 0:
            return out[: tot.value].copy(), lens[:nb].copy()
        cap = tot.value


def decode_blocks(model, comp, lens, block_size, orig_len, nthreads=1):
    a, p = _buf(comp)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    out = np.zeros(max(orig_len, 1), dtype=np.uint8)
    lib.w3o_decode_blocks(model.ptr, p, lens.ctypes.data_as(C.c_void_p), len(lens), block_size, orig_len,
                          out.ctypes.data_as(C.c_void_p), nthreads)
    return out[:orig_len]


# ---- state table / hashmap --------------------------------------------------
def state_table():
    t = np.ctypeslib.as_array(C.cast(lib.w3o_state_table(), C.POINTER(C.c_uint16)), shape=(3963, 3))
    return t.copy()  # columns: prob, next0, next1


def state_table_aux():
    t = np.ctypeslib.as_array(C.cast(lib.w3o_state_table_aux(), C.POINTER(C.c_uint16)), shape=(990, 3))
    return t.copy()


def state_table_csv(aux=False):
    t = state_table_aux() if aux else state_table()
    lines = ["state,tr0,tr1,prob"] + ["%d,%d,%d,%d" % (i, r[1], r[2], r[0]) for i, r in enumerate(t)]
    return "\n".join(lines) + "\n"
