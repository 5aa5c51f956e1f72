"""Host-side mirror of the reference's model surface (src/models, src/history,
src/mixers): same names and argument meaning, but each object only DESCRIBES a
model — it builds the w3_model_spec the HIP kernels execute.  There is no
per-bit Python arithmetic here; predict/update happen on the GPU."""
import ctypes as C
import threading

import numpy as np

from . import _lib as L


class W3Error(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__("w3hip error %d (%s) %s" % (code, L.load().w3_strerror(code).decode(), msg))


class StationaryModel:
    """models/ac_hash/stationary.rs:8-58 — per-bit-position static probabilities."""

    def __init__(self, buf):
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
        t = (C.c_uint16 * 8)()
        rc = L.load().w3_stationary_table(a.ctypes.data_as(C.c_void_p), len(a), t)
        if rc:
            raise W3Error(rc)
        self.table = list(t)

    @classmethod
    def new(cls, buf):
        return cls(buf)

    @classmethod
    def from_table(cls, table):
        o = cls.__new__(cls)
        o.table = [int(x) for x in table]
        assert len(o.table) == 8
        return o

    @classmethod
    def for_book1(cls):  # stationary.rs:40-42
        return cls.from_table([1, 50188, 62497, 15819, 22545, 31499, 22988, 29616])

    @classmethod
    def for_enwik7(cls):  # stationary.rs:44-46
        return cls.from_table([752, 50314, 58928, 21421, 24680, 30788, 24297, 32530])


class RawHistory:
    """history/raw_history.rs — last 32 bits."""
    kind = L.W3_HIST_RAW
    max_bits = 0
    table = [0] * 8

    @classmethod
    def new(cls):
        return cls()


class ACHistory:
    """history/ac_history.rs:9-47 — first max_bits arithmetic-coder output bits of the reversed 64-bit history."""
    kind = L.W3_HIST_AC

    def __init__(self, max_bits, model):
        self.max_bits = int(max_bits)
        self.table = list(model.table)

    @classmethod
    def new(cls, max_bits, model):
        return cls(max_bits, model)


class ACHistoryCached(ACHistory):
    """history/ac_history_cached.rs:31-76 — ACHistory with a std HashMap memo of coder states keyed by the low
    cache_size (and cache_size/2) history bits and the alignment.  The memo only skips re-encoding a prefix whose
    coder state is a pure function of its key, so hash() returns exactly ACHistory's value: cache_size changes the
    reference's CPU time, not its output (its own logs agree: bin/entropy-hashing-ac-cached/book1.mc.3.log:362 and
    bin/entropy-hashing-ac/book1.log:139 both give 262,871 bytes at (20, 3)).  Tested, not only argued: the CPU checker restates
    ac_history_cached.rs:37-76 with its memo, and tests/test_oracle_kats.py::test_ac_history_cached_equals_ac_history compares hash() with ACHistory's after every update for cache sizes 0..24, every
    alignment and three tables.  On the GPU every step's hash is computed in parallel (k_achash), so there is nothing to
    memoise: same spec as ACHistory."""

    def __init__(self, max_bits, model, cache_size=0):
        super().__init__(max_bits, model)
        self.cache_size = int(cache_size)

    @classmethod
    def new(cls, max_bits, model, cache_size):
        return cls(max_bits, model, cache_size)


class HuffHistory:
    """history/huff_history.rs:9-76 — the concatenated canonical Huffman codes of the completed bytes plus the code of the
    partial byte.  HuffHistory::new(buf, huff_size, rem_huff_size) builds the two code tables on the host (w3_huff_tables;
    among equal counts the reference's order is that of Rust's sort_unstable_by, here ascending symbol order);
    from_tables() takes ready (code, len) tables, e.g. ones dumped from the reference itself."""
    kind = L.W3_HIST_HUFF
    max_bits = 0
    table = [0] * 8

    def __init__(self, buf, huff_size, rem_huff_size):
        a = np.frombuffer(bytes(buf), dtype=np.uint8)
        self.tables = L.HuffTable()
        rc = L.load().w3_huff_tables(a.ctypes.data_as(C.c_void_p), len(a), huff_size, rem_huff_size, C.byref(self.tables))
        if rc:
            raise W3Error(rc, "HuffHistory::new: no symbols, or max length too big / too small for the alphabet")

    @classmethod
    def new(cls, buf, huff_size, rem_huff_size):
        return cls(buf, huff_size, rem_huff_size)

    @classmethod
    def from_tables(cls, code, length, rem_code, rem_length):
        o = cls.__new__(cls)
        o.tables = L.HuffTable()
        for i in range(256):
            o.tables.code[i], o.tables.len[i], o.tables.rem_code[i], o.tables.rem_len[i] = code[i], length[i], rem_code[i], rem_length[i]
        return o


class Model:
    """trait Model (models/mod.rs:12-15), as a spec tree."""
    _tls = threading.local()   # .huff_index: HuffHistory object -> slot in w3_model_spec::huff while spec() walks the tree (per thread:
                               # several contexts may build specs from several host threads at once)

    def _nodes(self):
        raise NotImplementedError

    def _huff_sets(self):
        """HuffHistory objects of the tree's leaves, in leaf order (each gets a slot in w3_model_spec::huff)."""
        return []

    def spec(self):
        huffs = []
        for h in self._huff_sets():
            if not any(h is x for x in huffs):
                huffs.append(h)
        if len(huffs) > L.W3_MAX_HUFF:
            raise W3Error(L.W3_E_UNSUPPORTED, "more than %d HuffHistory table sets" % L.W3_MAX_HUFF)
        Model._tls.huff_index = {id(h): i for i, h in enumerate(huffs)}   # read by AdaptiveModel._leaf while the tree is walked
        try:
            nodes = self._nodes()
        finally:
            Model._tls.huff_index = {}
        if len(nodes) > L.W3_MAX_NODES:
            raise W3Error(L.W3_E_UNSUPPORTED, "model tree too large")
        s = L.ModelSpec()
        s.n_nodes = len(nodes)
        for i, nd in enumerate(nodes):
            s.nodes[i] = nd
        if huffs:
            arr = (L.HuffTable * len(huffs))(*[h.tables for h in huffs])
            s._huff_keepalive = arr   # the spec points into this array
            s.n_huff = len(huffs)
            s.huff = C.cast(arr, C.POINTER(L.HuffTable))
        rc = L.load().w3_spec_validate(C.byref(s))
        if rc:
            raise W3Error(rc, "model spec rejected")
        return s


def _leaf(bits, align, history=L.W3_HIST_NONE, max_bits=0, table=None, frozen=0):
    nd = L.Node()
    nd.kind = L.W3_NODE_ORDERN
    if not (0 <= bits <= 255 and 0 <= align <= 255 and 0 <= max_bits <= 255):
        raise W3Error(L.W3_E_INVALID, "u8 parameter out of range")  # the reference takes u8s
    nd.bits, nd.align, nd.history, nd.max_bits, nd.frozen = bits, align, history, max_bits, frozen
    for i, v in enumerate(table or [0] * 8):
        nd.table[i] = v
    return nd


class AdaptiveModel(Model):
    """trait AdaptiveModel (models/mod.rs:17-21): leaves that FrozenModel can wrap."""

    def __init__(self, bits, align, history=None):
        self.bits, self.align, self.history = int(bits), int(align), history

    def _leaf(self, frozen=0):
        h = self.history
        if h is None:
            return _leaf(self.bits, self.align, frozen=frozen)
        nd = _leaf(self.bits, self.align, h.kind, h.max_bits, h.table, frozen=frozen)
        if h.kind == L.W3_HIST_HUFF:
            nd.reserved = getattr(Model._tls, "huff_index", {}).get(id(h), 0)
        return nd

    def _huff_sets(self):
        return [self.history] if self.history is not None and self.history.kind == L.W3_HIST_HUFF else []

    def _nodes(self):
        return [self._leaf()]


class OrderN(AdaptiveModel):
    """models/ordern.rs — OrderN::new(bits_in_context, alignment_bits)."""

    def __init__(self, bits_in_context, alignment_bits):
        super().__init__(bits_in_context, alignment_bits)

    @classmethod
    def new(cls, bits_in_context, alignment_bits):
        return cls(bits_in_context, alignment_bits)


class Order0(AdaptiveModel):
    """models/order0.rs — 2^11 counters, ctx = alignment<<8 | last 8 bits (== OrderN(11,3), bin/cmp/main.rs:14-24)."""

    def __init__(self):
        super().__init__(11, 3)

    @classmethod
    def new(cls):
        return cls()


class Order1(AdaptiveModel):
    """models/order1.rs — 2^19 counters (== OrderN(19,3))."""

    def __init__(self):
        super().__init__(19, 3)

    @classmethod
    def new(cls):
        return cls()


class OrderNEntropy(AdaptiveModel):
    """models/ordern_entropy.rs — OrderNEntropy::new(bits, align, history)."""

    def __init__(self, bits_in_context, alignment_bits, history):
        super().__init__(bits_in_context, alignment_bits, history)

    @classmethod
    def new(cls, bits_in_context, alignment_bits, history):
        return cls(bits_in_context, alignment_bits, history)


class FrozenModel(Model):
    """models/frozen.rs — update() advances the context but never adapts."""

    def __init__(self, model):
        if not isinstance(model, AdaptiveModel):
            raise TypeError("FrozenModel<T: AdaptiveModel> (models/frozen.rs:3)")
        self.model = model

    @classmethod
    def new(cls, model):
        return cls(model)

    def _huff_sets(self):
        return self.model._huff_sets()

    def _nodes(self):
        return [self.model._leaf(frozen=1)]


class BestOfTwoModel(Model):
    """models/mod.rs:42-75 with OpinionMixer2 (mixers/opinion_mixer2.rs)."""

    def __init__(self, m1, m2):
        self.m1, self.m2 = m1, m2

    @classmethod
    def new(cls, m1, m2):
        return cls(m1, m2)

    def _huff_sets(self):
        return self.m1._huff_sets() + self.m2._huff_sets()

    def _nodes(self):
        nd = L.Node()
        nd.kind = L.W3_NODE_BEST_OF_TWO
        return self.m1._nodes() + self.m2._nodes() + [nd]


# ---- build-defined models (SURVEY §8 A19 ii-v; include/w3hip.h) ----------------------
class StateTable:
    """trait StateTable (state_table/mod.rs:3-23) over NaiveStateTable (state_table/naive.rs): associated
    functions only, evaluated from the library's host-side copy of the table the kernels stage in LDS."""
    _rows = None

    @classmethod
    def rows(cls):
        if cls._rows is None:
            t = np.zeros(3963 * 3, dtype=np.uint16)
            rc = L.load().w3_state_table(t.ctypes.data_as(C.c_void_p))
            if rc:
                raise W3Error(rc)
            cls._rows = t.reshape(3963, 3)
        return cls._rows

    @classmethod
    def next(cls, state, bit):
        return int(cls.rows()[state, 1 + bit])

    @classmethod
    def p(cls, state):
        return int(cls.rows()[state, 0])

    @classmethod
    def next4(cls, states, nib):  # mod.rs:5-12: nibble applied MSB first
        return [cls.next(states[i], (nib >> (3 - i)) & 1) for i in range(4)]

    @classmethod
    def p4(cls, states):  # mod.rs:15-22
        return [cls.p(s) for s in states]


class HashMap:
    """hashmap.rs:7-22 — HashMap::new(size): only the sizing rule lives on the host; the table itself is
    per-block device state."""

    def __init__(self, size):
        import math
        self.log_cell_count = int(math.log2(float(size)) - math.log2(96.0))  # f64 arithmetic, truncated (hashmap.rs:9)

    @classmethod
    def new(cls, size):
        return cls(size)


class SlotModel(Model):
    """State-table CM leaf (docs/hashslots.md): context = previous `order` bytes, one Cell touch per nibble,
    probabilities and updates through the 12-bit NaiveStateTable.  BUILD-DEFINED wiring of hashmap.rs +
    state_table/naive.rs, which no reference model uses."""

    def __init__(self, order, hashmap):
        self.order = int(order)
        self.log_cells = hashmap.log_cell_count if isinstance(hashmap, HashMap) else int(hashmap)

    @classmethod
    def new(cls, order, hashmap):
        return cls(order, hashmap)

    def _nodes(self):
        nd = L.Node()
        nd.kind = L.W3_NODE_SLOT_STATE
        if not (0 <= self.order <= 255 and 0 <= self.log_cells <= 255):
            raise W3Error(L.W3_E_INVALID, "u8 parameter out of range")
        nd.bits, nd.log_cells = self.order, self.log_cells
        return [nd]


class Mixer:
    """The mixer surface (SURVEY §8 A19 v): OpinionMixer2 (mixers/opinion_mixer2.rs) is the stateless
    two-input instance used by BestOfTwoModel; APM is the adaptive one-input instance."""


class OpinionMixer2(Mixer):
    @staticmethod
    def mix(p1, p2):  # opinion_mixer2.rs:5-10 (host-side convenience; the kernels apply it per step)
        d1, d2 = abs(p1 - 32768), abs(p2 - 32768)
        return p1 if d1 >= d2 else p2


class APM(Model, Mixer):
    """Adaptive probability map ("APM mixers", README.md:10): BUILD-DEFINED, 33 interpolated buckets over
    stretch(p) per context row, output (p + 3 apm) / 4."""
    ORDER0, ORDER1 = L.W3_APM_ORDER0, L.W3_APM_ORDER1

    def __init__(self, model, ctx=L.W3_APM_ORDER0, rate=7):
        self.model, self.ctx, self.rate = model, int(ctx), int(rate)

    @classmethod
    def new(cls, model, ctx=L.W3_APM_ORDER0, rate=7):
        return cls(model, ctx, rate)

    def _huff_sets(self):
        return self.model._huff_sets()

    def _nodes(self):
        nd = L.Node()
        nd.kind = L.W3_NODE_APM
        if not (0 <= self.ctx <= 255 and 0 <= self.rate <= 255):
            raise W3Error(L.W3_E_INVALID, "u8 parameter out of range")
        nd.align, nd.max_bits = self.ctx, self.rate
        return self.model._nodes() + [nd]


def full_cm(log_cells=14):
    """BASELINE.json configs[2] — "full 12-bit state-table CM (all src/models + APM chain)": the three Counter
    orders, four slot-state orders, OpinionMixer2 selection, two APM stages."""
    m = BestOfTwoModel(BestOfTwoModel(Order0(), Order1()), OrderN(27, 3))
    for order in (1, 2, 3, 4):
        m = BestOfTwoModel(m, SlotModel(order, log_cells))
    return APM(APM(m, APM.ORDER0, 7), APM.ORDER1, 6)


def init_model():
    """main.rs:146-152 — the main binary's default model."""
    return OrderNEntropy.new(11, 3, ACHistory.new(8, StationaryModel.for_book1()))
