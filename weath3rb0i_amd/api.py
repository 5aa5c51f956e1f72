"""compress / decompress entry points over the C ABI (include/w3hip.h).

Context wraps one w3_ctx (one GPU).  Host-buffer calls take bytes / numpy
arrays; *_device calls take torch CUDA tensors (PyTorch is only the owner of
device memory and streams here)."""
import ctypes as C

import numpy as np

from . import _lib as L
from .models import Model, W3Error, init_model

MAGIC_STR = b"w30i"  # main.rs:14


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def _host_ptr(buf):
    """(address, bytes) of a host buffer: a numpy array, or anything with data_ptr() / numel() / element_size() (a torch CPU tensor —
    pinned ones make the copies asynchronous)."""
    if isinstance(buf, np.ndarray):
        assert buf.flags["C_CONTIGUOUS"]
        return buf.ctypes.data, buf.nbytes
    return buf.data_ptr(), buf.numel() * buf.element_size()


class Context:
    def __init__(self, device=0):
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.w3_ctx_create(device, C.byref(h))
        if rc:
            raise W3Error(rc, "w3_ctx_create(device=%d): no usable HIP device" % device)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.w3_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise W3Error(rc, self.lib.w3_last_error(self.h).decode())

    def set_path(self, path):
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_PATH, {"auto": 0, "generic": 1, "twophase": 2}[path]))

    def set_coder(self, mode):
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_CODER, {"x4": 0, "fast": 1, "robust": 2, "x2": 3, "x3": 4, "x5": 5}[mode]))

    def set_acc_limit(self, bits):
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_ACC_LIMIT, bits))

    def set_variant(self, *names):
        """Cross-check hook (W3_OPT_VARIANT): alternative bit-exact implementations, by name: no_lds_atomics, partition4,
        no_chained_partition, cm_unstaged, no_side_stream, half_cu (synchronous calls in the pipeline's kernel shapes), full_cu
        (submitted calls in the plain shapes), slot_table / slot_sorted (slot-state leaves on k_slot / on the sorted replay whatever
        the block count), decode_lane (decode on the lane-per-block kernels, not k_decode_spec).  No names = defaults."""
        bits = {"no_lds_atomics": L.W3_VAR_NO_LDS_ATOMICS, "partition4": L.W3_VAR_PARTITION4, "no_chained_partition": L.W3_VAR_NO_CHAINED_PARTITION,
                "cm_unstaged": L.W3_VAR_CM_UNSTAGED, "no_side_stream": L.W3_VAR_NO_SIDE_STREAM, "inject_lds_fault": L.W3_VAR_INJECT_LDS_FAULT,
                "half_cu": L.W3_VAR_HALF_CU, "full_cu": L.W3_VAR_FULL_CU, "slot_table": L.W3_VAR_SLOT_TABLE, "slot_sorted": L.W3_VAR_SLOT_SORTED, "decode_lane": L.W3_VAR_DECODE_LANE}
        v = 0
        for nm in names:
            v |= bits[nm]
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_VARIANT, v))

    def set_verify(self, on=True):
        """W3_OPT_VERIFY: sampled ballot-round re-prediction after every predict phase that used LDS-add rounds (default on)."""
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_VERIFY, int(on)))

    def set_tune(self, bits=0):
        """W3_OPT_TUNE: scheduling experiments of the submit / wait pipeline (set before the first encode_submit)."""
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_TUNE, int(bits)))

    def set_fault_block(self, block=-1):
        """Test hook (with set_variant("inject_lds_fault")): the one block the injected fault hits; -1 = every block."""
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_FAULT_BLOCK, int(block)))

    def set_slot_budget_mb(self, mb=0):
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_SLOT_BUDGET_MB, int(mb)))

    def set_host_chunk_blocks(self, blocks=0):
        """W3_OPT_HOST_CHUNK_BLOCKS: blocks per pipelined piece of a host-buffer call (0 = default)."""
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_HOST_CHUNK_BLOCKS, int(blocks)))

    def set_timing(self, on=True):
        self._chk(self.lib.w3_ctx_set_option(self.h, L.W3_OPT_TIMING, int(on)))

    def timing(self):
        t = L.Timing()
        self._chk(self.lib.w3_get_timing(self.h, C.byref(t)))
        return {k: (list(getattr(t, k)) if k in ("part_ms", "rank_ms") else getattr(t, k)) for k, _ in L.Timing._fields_}

    # ---- host buffers ----------------------------------------------------
    def encode_blocks(self, model, data, block_size, out_cap=None):
        """-> (concatenated streams: np.uint8[], block_lens: np.uint32[])"""
        spec = model.spec() if isinstance(model, Model) else model
        a = _u8(data)
        n = len(a)
        nb = (n + block_size - 1) // block_size if block_size else 0
        if out_cap is None:
            out_cap = 2 * n + 64 * nb + 64
        while True:
            out = np.empty(max(out_cap, 1), dtype=np.uint8)
            lens = np.zeros(max(nb, 1), dtype=np.uint32)
            olen = C.c_size_t()
            rc = self.lib.w3_encode_blocks(self.h, C.byref(spec), a.ctypes.data_as(C.c_void_p), n, block_size,
                                           out.ctypes.data_as(C.c_void_p), out_cap, C.byref(olen), lens.ctypes.data_as(C.c_void_p))
            if rc == L.W3_E_NOSPACE and olen.value > out_cap:
                out_cap = olen.value
                continue
            self._chk(rc)
            return out[: olen.value], lens[:nb]

    def encode_host_submit(self, model, data, block_size, out, lens):
        """w3_encode_host_submit: `data`, `out` (uint8) and `lens` (uint32[nb]) are host buffers — numpy arrays or torch CPU tensors,
        pinned or not — that must stay alive and untouched until encode_host_wait(job).  -> job handle"""
        spec = model.spec() if isinstance(model, Model) else model
        ip, n = _host_ptr(data)
        op, cap = _host_ptr(out)
        lp, _ = _host_ptr(lens)
        job = C.c_int(-1)
        self._chk(self.lib.w3_encode_host_submit(self.h, C.byref(spec), C.c_void_p(ip), n, block_size, C.c_void_p(op), cap, C.c_void_p(lp), C.byref(job)))
        return job.value

    def encode_host_wait(self, job):
        """w3_encode_host_wait -> compressed bytes now in the job's `out` (raises what w3_encode_blocks would have raised)."""
        olen = C.c_size_t()
        rc = self.lib.w3_encode_host_wait(self.h, int(job), C.byref(olen))
        self._chk(rc)
        return olen.value

    def host_max_in_flight(self, n, block_size, model=None):
        spec = None if model is None else (model.spec() if isinstance(model, Model) else model)
        return int(self.lib.w3_encode_host_max_in_flight(C.byref(spec) if spec is not None else None, int(n), int(block_size)))

    def decode_blocks(self, model, comp, block_lens, block_size, orig_len):
        spec = model.spec() if isinstance(model, Model) else model
        a = _u8(comp)
        lens = np.ascontiguousarray(block_lens, dtype=np.uint32)
        out = np.empty(max(orig_len, 1), dtype=np.uint8)
        rc = self.lib.w3_decode_blocks(self.h, C.byref(spec), a.ctypes.data_as(C.c_void_p), len(a), lens.ctypes.data_as(C.c_void_p),
                                       len(lens), block_size, orig_len, out.ctypes.data_as(C.c_void_p))
        self._chk(rc)
        return out[:orig_len]

    def encode_stats(self, model, data, block_size):
        """ACStats (helpers.rs:60-90) per block: the bit counts the reference's `csize = bits / 8` comes from.  -> np.uint32[nb]"""
        spec = model.spec() if isinstance(model, Model) else model
        a = _u8(data)
        nb = (len(a) + block_size - 1) // block_size if block_size else 0
        bits = np.zeros(max(nb, 1), dtype=np.uint32)
        self._chk(self.lib.w3_encode_stats(self.h, C.byref(spec), a.ctypes.data_as(C.c_void_p), len(a), block_size, bits.ctypes.data_as(C.c_void_p)))
        return bits[:nb]

    def sweep_ordern(self, data, block_size, configs):
        """OrderN(bits, align) for every (bits, align) in `configs`, all in one device launch (bin/ordern/main.rs:9-80).
        -> np.uint32[len(configs)][nb] ACStats bit counts."""
        a = _u8(data)
        nb = (len(a) + block_size - 1) // block_size if block_size else 0
        cb = np.array([c[0] for c in configs], dtype=np.uint8)
        ca = np.array([c[1] for c in configs], dtype=np.uint8)
        out = np.zeros((len(configs), max(nb, 1)), dtype=np.uint32)
        self._chk(self.lib.w3_sweep_ordern(self.h, a.ctypes.data_as(C.c_void_p), len(a), block_size, cb.ctypes.data_as(C.c_void_p),
                                           ca.ctypes.data_as(C.c_void_p), len(configs), out.ctypes.data_as(C.c_void_p)))
        return out[:, :nb]

    def export_counters(self, model, data):
        """Context statistics export (README.md:9): the model's Counter table after `data` as one stream.
        -> (n0: np.uint16[2^bits], n1: np.uint16[2^bits])"""
        spec = model.spec()
        a = _u8(data)
        bits = spec.nodes[0].bits
        t = np.zeros(1 << bits, dtype=np.uint32)
        self._chk(self.lib.w3_export_counters(self.h, C.byref(spec), a.ctypes.data_as(C.c_void_p), len(a), t.ctypes.data_as(C.c_void_p)))
        return (t & 0xFFFF).astype(np.uint16), (t >> 16).astype(np.uint16)

    def predict_blocks(self, model, data, block_size):
        spec = model.spec()
        a = _u8(data)
        p = np.empty(max(len(a) * 8, 1), dtype=np.uint16)
        self._chk(self.lib.w3_predict_blocks(self.h, C.byref(spec), a.ctypes.data_as(C.c_void_p), len(a), block_size,
                                             p.ctypes.data_as(C.c_void_p)))
        return p[: len(a) * 8]

    # ---- reference container (main.rs:89-144) ------------------------------
    def compress(self, data, model=None):
        """compress(): b"w30i" + u64 BE len + one stream."""
        model = model or init_model()
        spec = model.spec()
        a = _u8(data)
        cap = 2 * len(a) + 128
        while True:
            out = np.empty(cap, dtype=np.uint8)
            olen = C.c_size_t()
            rc = self.lib.w3_compress_stream(self.h, C.byref(spec), a.ctypes.data_as(C.c_void_p), len(a),
                                             out.ctypes.data_as(C.c_void_p), cap, C.byref(olen))
            if rc == L.W3_E_NOSPACE and olen.value > cap:
                cap = olen.value
                continue
            self._chk(rc)
            return out[: olen.value].tobytes()

    def decompress(self, data, model=None):
        model = model or init_model()
        spec = model.spec()
        a = _u8(data)
        if len(a) >= 12 and a[:4].tobytes() == MAGIC_STR:
            n = int.from_bytes(a[4:12].tobytes(), "big")
        else:
            n = 0
        out = np.empty(max(n, 1), dtype=np.uint8)
        olen = C.c_size_t()
        rc = self.lib.w3_decompress_stream(self.h, C.byref(spec), a.ctypes.data_as(C.c_void_p), len(a),
                                           out.ctypes.data_as(C.c_void_p), n, C.byref(olen))
        self._chk(rc)
        return out[: olen.value].tobytes()

    # ---- device-resident (torch tensors own the memory) ------------------------
    def encode_blocks_device(self, model, d_in, block_size, d_out, d_lens, d_total, stream=None):
        """d_in/d_out: torch.uint8 CUDA tensors, d_lens: int32/uint32[nb], d_total: int64[1].  Returns rc-checked None."""
        spec = model.spec() if isinstance(model, Model) else model
        # stream: a hipStream_t handle as an int.  None or 0 = the ctx's own (blocking) stream, which is ordered against the
        # legacy default stream — torch's default stream IS that stream (handle 0), so both spellings mean the same order.
        st = C.c_void_p(stream) if stream else None
        rc = self.lib.w3_encode_blocks_device(self.h, C.byref(spec), C.c_void_p(d_in.data_ptr()), d_in.numel(), block_size,
                                              C.c_void_p(d_out.data_ptr()), d_out.numel(), C.c_void_p(d_lens.data_ptr()),
                                              C.c_void_p(d_total.data_ptr()), st)
        self._chk(rc)

    def encode_submit(self, model, d_in, block_size, d_out, d_lens, d_total, stream=None):
        """w3_encode_submit: enqueue the encode and return a job handle at once (at most max_in_flight(n, block_size, model) jobs in
        flight: four up to 4,096 blocks, three up to 12,288, two beyond; the next call's predict phase runs beside this call's APM and
        coder kernels).  Keep every tensor alive and untouched until encode_wait(job)."""
        spec = model.spec() if isinstance(model, Model) else model
        st = C.c_void_p(stream) if stream else None
        job = C.c_int(-1)
        rc = self.lib.w3_encode_submit(self.h, C.byref(spec), C.c_void_p(d_in.data_ptr()), d_in.numel(), block_size,
                                       C.c_void_p(d_out.data_ptr()), d_out.numel(), C.c_void_p(d_lens.data_ptr()),
                                       C.c_void_p(d_total.data_ptr()), st, C.byref(job))
        self._chk(rc)
        return job.value

    def max_in_flight(self, n, block_size, model=None):
        """w3_encode_max_in_flight: submitted calls of this size one context keeps in flight (4 up to 4,096 blocks, 3 up to 12,288, else 2;
        models with slot-state leaves: 2)."""
        spec = None if model is None else (model.spec() if isinstance(model, Model) else model)
        return int(self.lib.w3_encode_max_in_flight(C.byref(spec) if spec is not None else None, int(n), int(block_size)))

    def encode_wait(self, job):
        """w3_encode_wait: block until the job's output is complete (raises what the synchronous call would have raised)."""
        self._chk(self.lib.w3_encode_wait(self.h, int(job)))

    def decode_blocks_device(self, model, d_comp, d_lens, block_size, orig_len, d_out, stream=None):
        spec = model.spec() if isinstance(model, Model) else model
        st = C.c_void_p(stream) if stream else None
        rc = self.lib.w3_decode_blocks_device(self.h, C.byref(spec), C.c_void_p(d_comp.data_ptr()), d_comp.numel(), C.c_void_p(d_lens.data_ptr()),
                                              d_lens.numel(), block_size, orig_len, C.c_void_p(d_out.data_ptr()), st)
        self._chk(rc)


def encode_blocks_sharded_device(ctxs, model, d_ins, block_size, d_out, d_lens, root=0, transport="auto"):
    """w3_encode_blocks_sharded_device: ONE process, one Context per device; d_ins[r] (torch.uint8 CUDA tensor on ctxs[r]'s device) is
    shard r of one stream (w3_shard_range); the packed streams and the length table are gathered on ctxs[root]'s device (d_out,
    d_lens) with RCCL (grouped send/recv over xGMI) or device copies.  -> per-shard compressed byte counts."""
    spec = model.spec() if isinstance(model, Model) else model
    k = len(ctxs)
    hs = (C.c_void_p * k)(*[c.h for c in ctxs])
    ins = (C.c_void_p * k)(*[C.c_void_p(t.data_ptr() if t.numel() else 0) for t in d_ins])
    ns = (C.c_size_t * k)(*[t.numel() for t in d_ins])
    totals = (C.c_uint64 * k)()
    tr = {"auto": L.W3_GATHER_AUTO, "rccl": L.W3_GATHER_RCCL, "peer_copy": L.W3_GATHER_PEER_COPY}[transport]
    rc = ctxs[0].lib.w3_encode_blocks_sharded_device(hs, k, C.byref(spec), ins, ns, block_size, root, C.c_void_p(d_out.data_ptr()), d_out.numel(),
                                                     C.c_void_p(d_lens.data_ptr()), totals, tr)
    if rc:
        raise W3Error(rc, ctxs[0].lib.w3_last_error(ctxs[0].h).decode())
    return [int(t) for t in totals]


def encode_sharded_submit(ctxs, model, d_ins, block_size):
    """w3_encode_sharded_submit: one STEP of a stream of sharded encodes — every shard's encode is enqueued on its context (d_ins[r]:
    torch.uint8 CUDA tensor on ctxs[r]'s device, kept alive until the step has been waited for).  -> step handle"""
    spec = model.spec() if isinstance(model, Model) else model
    k = len(ctxs)
    hs = (C.c_void_p * k)(*[c.h for c in ctxs])
    ins = (C.c_void_p * k)(*[C.c_void_p(t.data_ptr() if t.numel() else 0) for t in d_ins])
    ns = (C.c_size_t * k)(*[t.numel() for t in d_ins])
    sjob = C.c_int(-1)
    rc = ctxs[0].lib.w3_encode_sharded_submit(hs, k, C.byref(spec), ins, ns, block_size, C.byref(sjob))
    if rc:
        raise W3Error(rc, ctxs[0].lib.w3_last_error(ctxs[0].h).decode())
    return sjob.value


def encode_sharded_wait(ctxs, sjob, d_out, d_lens, root=0, transport="auto"):
    """w3_encode_sharded_wait: completes the step and gathers its packed streams and length table on ctxs[root]'s device.
    -> per-shard compressed byte counts"""
    k = len(ctxs)
    hs = (C.c_void_p * k)(*[c.h for c in ctxs])
    totals = (C.c_uint64 * k)()
    tr = {"auto": L.W3_GATHER_AUTO, "rccl": L.W3_GATHER_RCCL, "peer_copy": L.W3_GATHER_PEER_COPY}[transport]
    rc = ctxs[0].lib.w3_encode_sharded_wait(hs, k, int(sjob), root, C.c_void_p(d_out.data_ptr()), d_out.numel(), C.c_void_p(d_lens.data_ptr()), totals, tr)
    if rc:
        raise W3Error(rc, ctxs[0].lib.w3_last_error(ctxs[0].h).decode())
    return [int(t) for t in totals]


def sharded_max_in_flight(ctxs, model, sizes, block_size):
    spec = model.spec() if isinstance(model, Model) else model
    ns = (C.c_size_t * len(sizes))(*sizes)
    return int(ctxs[0].lib.w3_encode_sharded_max_in_flight(C.byref(spec), ns, len(sizes), block_size))
