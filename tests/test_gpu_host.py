"""GPU parity of the HOST-BUFFER path (compress() of main.rs:89-113 reads a file and writes a file): w3_encode_blocks cut into
pipelined pieces, and w3_encode_host_submit / w3_encode_host_wait with calls in flight — against the CPU oracle, byte for byte,
from pageable and from pinned memory."""
import numpy as np
import pytest

import weath3rb0i_amd as w3
from tests.synth import lcg_text, markov_text
from tests.test_gpu_parity import pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = w3.Context(0)
    yield c
    c.close()


def apm012(oracle):
    return (lambda: w3.APM(w3.BestOfTwoModel(w3.BestOfTwoModel(w3.Order0(), w3.Order1()), w3.OrderN(27, 3))),
            lambda: oracle.APM(oracle.BestOfTwoModel(oracle.BestOfTwoModel(oracle.Order0(), oracle.Order1()), oracle.OrderN(27, 3))))


def mk(oracle, name):
    return apm012(oracle) if name == "o012_apm" else pair(oracle, name)


@pytest.mark.parametrize("name", ["o012_apm", "best012", "order0", "main_default", "ordern_22_2", "huff_11_text"])
def test_encode_blocks_in_ragged_pieces(ctx, oracle, name):
    """w3_encode_blocks cut into pieces of whole blocks (W3_OPT_HOST_CHUNK_BLOCKS: 7 blocks, so 41 blocks + a ragged tail are six pieces,
    the last one short): streams and length table equal the oracle's and the one-piece call's, from pageable numpy buffers."""
    bs = 4096
    data = markov_text(41 * bs + 777, seed=71)
    dev, orc = mk(oracle, name)
    want, wlens = oracle.encode_blocks(orc(), data, bs, nthreads=8)
    one, olens = ctx.encode_blocks(dev(), data, bs)
    assert olens.tolist() == wlens.tolist() and one.tobytes() == want.tobytes(), name
    for pieces_of in (7, 1, 40, 64):
        ctx.set_host_chunk_blocks(pieces_of)
        try:
            out, lens = ctx.encode_blocks(dev(), data, bs)
            assert ctx.timing()["n_parts"] == (42 + pieces_of - 1) // pieces_of
            # w3_decode_blocks in runs of that many blocks (what an input above the per-call limit of 4 GiB goes through)
            assert ctx.decode_blocks(dev(), out, lens, bs, len(data)).tobytes() == data, (name, pieces_of)
        finally:
            ctx.set_host_chunk_blocks(0)
        assert lens.tolist() == wlens.tolist(), (name, pieces_of)
        assert out.tobytes() == want.tobytes(), (name, pieces_of)


def test_encode_blocks_default_piece_schedule(ctx, oracle):
    """More than 4,096 blocks: the default schedule (equal pieces of at most 4,096 blocks) gives the oracle's streams, like the
    one-piece call."""
    bs = 64
    data = markov_text(bs * 9001 + 17, seed=73)
    dev, orc = mk(oracle, "best012")
    want, wlens = oracle.encode_blocks(orc(), data, bs, nthreads=8)
    out, lens = ctx.encode_blocks(dev(), data, bs)
    assert ctx.timing()["n_parts"] == 3          # 9,002 blocks: pieces of 3,001 / 3,001 / 3,000
    assert lens.tolist() == wlens.tolist() and out.tobytes() == want.tobytes()


def test_encode_blocks_pieces_nospace_and_lane_per_block_specs(ctx, oracle):
    """A call in pieces that runs out of room reports the size needed and the whole length table (as the one-piece call does); specs
    that w3_encode_submit runs synchronously (lane-per-block path) are one piece whatever the option says."""
    import ctypes as C
    from weath3rb0i_amd import _lib as L
    bs = 2048
    data = np.frombuffer(markov_text(33 * bs + 5, seed=72), dtype=np.uint8)
    want, wlens = oracle.encode_blocks(oracle.Order1(), data.tobytes(), bs, nthreads=8)
    spec = w3.Order1().spec()
    ctx.set_host_chunk_blocks(5)
    try:
        for cap in (100, len(want) - 1, len(want) // 2):
            out = np.zeros(max(cap, 1), dtype=np.uint8)
            lens = np.zeros(34, dtype=np.uint32)
            olen = C.c_size_t()
            rc = ctx.lib.w3_encode_blocks(ctx.h, C.byref(spec), data.ctypes.data_as(C.c_void_p), len(data), bs,
                                          out.ctypes.data_as(C.c_void_p), cap, C.byref(olen), lens.ctypes.data_as(C.c_void_p))
            assert rc == L.W3_E_NOSPACE and olen.value == len(want) and lens.tolist() == wlens.tolist(), cap
        ctx.set_path("generic")
        try:
            out, lens = ctx.encode_blocks(w3.Order1(), data, bs)
            assert ctx.timing()["n_parts"] == 1 and ctx.timing()["path"] == 1
        finally:
            ctx.set_path("auto")
        assert out.tobytes() == want.tobytes()
    finally:
        ctx.set_host_chunk_blocks(0)


def _pinned(n, dtype):
    import torch
    return torch.empty(n, dtype=dtype).pin_memory()


@pytest.mark.parametrize("pinned", [False, True])
def test_host_submit_wait_calls_in_flight(ctx, oracle, pinned):
    """w3_encode_host_submit / w3_encode_host_wait: as many calls in flight as w3_encode_host_max_in_flight allows (one more than the device
    job slots), different inputs and specs together, waited for in and out of order, one more submission refused, synchronous entry
    points refused meanwhile; every output equals the oracle's.  Pageable numpy buffers and pinned torch tensors."""
    import torch
    from weath3rb0i_amd import _lib as L
    bs = 2048
    datas = [markov_text(200 * 1024 + 333, seed=81) + bytes(30 * 1024), lcg_text(150 * 1024 + 7, seed=29) + markov_text(90 * 1024, seed=82),
             markov_text(64 * 1024 + 1, seed=83)]
    names = ["o012_apm", "best012", "order0", "main_default"]
    want = {(k, nm): oracle.encode_blocks(mk(oracle, nm)[1](), datas[k], bs, nthreads=8) for k in range(3) for nm in names}
    depth = ctx.host_max_in_flight(len(datas[0]), bs)
    assert depth == 5 and ctx.host_max_in_flight(10**9, 65536) == 3
    if pinned:
        ins = []
        for d in datas:
            t = _pinned(len(d), torch.uint8)
            t.numpy()[:] = np.frombuffer(d, dtype=np.uint8)
            ins.append(t)
        outs = [(_pinned(2 * len(datas[0]) + 8192, torch.uint8), _pinned(256, torch.int32)) for _ in range(depth)]
        view = lambda o, m: o.numpy()[:m]
    else:
        ins = [np.frombuffer(d, dtype=np.uint8).copy() for d in datas]
        outs = [(np.zeros(2 * len(datas[0]) + 8192, dtype=np.uint8), np.zeros(256, dtype=np.uint32)) for _ in range(depth)]
        view = lambda o, m: o[:m]

    def finish(entry):
        job, bi, kk, nn = entry
        total = ctx.encode_host_wait(job)
        w_out, w_lens = want[(kk, nn)]
        o, ln = outs[bi]
        assert total == len(w_out), (kk, nn)
        assert view(ln, len(w_lens)).astype(np.uint32).tolist() == w_lens.tolist(), (kk, nn)
        assert view(o, total).tobytes() == w_out.tobytes(), (kk, nn)

    rng = np.random.default_rng(7)
    pending, free = [], list(range(depth))
    refused = False
    for i in range(19):
        if len(pending) == depth:
            if not refused:
                with pytest.raises(w3.W3Error) as e:
                    ctx.encode_host_submit(mk(oracle, "order0")[0](), ins[0], bs, *outs[0])
                assert e.value.code == L.W3_E_INVALID
                with pytest.raises(w3.W3Error) as e:   # no synchronous call while host jobs are in flight
                    ctx.encode_blocks(w3.Order0(), datas[2][:9000], bs)
                assert e.value.code == L.W3_E_INVALID
                refused = True
            entry = pending.pop(0 if i % 2 else int(rng.integers(0, len(pending))))
            finish(entry)
            free.append(entry[1])
        k, nm = i % 3, names[i % len(names)]
        bi = free.pop(0)
        job = ctx.encode_host_submit(mk(oracle, nm)[0](), ins[k], bs, *outs[bi])
        assert 0 <= job < 5 and job not in [p[0] for p in pending]
        pending.append((job, bi, k, nm))
    for entry in reversed(pending):
        finish(entry)
    with pytest.raises(w3.W3Error):
        ctx.encode_host_wait(0)   # nothing in flight
    # out_cap too small: W3_E_NOSPACE from the wait, the size needed in *out_len; the length table is still delivered
    import ctypes as C
    small = np.zeros(1000, dtype=np.uint8)
    lens = np.zeros(256, dtype=np.uint32)
    job = ctx.encode_host_submit(w3.Order0(), ins[2], bs, small, lens)
    olen = C.c_size_t()
    rc = ctx.lib.w3_encode_host_wait(ctx.h, job, C.byref(olen))
    w_out, w_lens = want[(2, "order0")]
    assert rc == L.W3_E_NOSPACE and olen.value == len(w_out) and lens[:len(w_lens)].tolist() == w_lens.tolist()
    # the synchronous call works again
    out, lens2 = ctx.encode_blocks(w3.Order0(), datas[2], bs)
    assert out.tobytes() == w_out.tobytes()


def test_entry_points_refused_while_a_job_is_in_flight(ctx, oracle):
    """include/w3hip.h: every entry point but submit / wait returns W3_E_INVALID while a submitted call is in flight (they all work on job
    0's workspace or on the options the job has taken)."""
    import torch
    from weath3rb0i_amd import _lib as L
    bs = 4096
    data = markov_text(64 * 1024 + 3, seed=91)
    a = np.frombuffer(data, dtype=np.uint8).copy()
    d_in = torch.from_numpy(a).cuda()
    nb = (len(a) + bs - 1) // bs
    d_out = torch.empty(2 * len(a) + 64 * nb + 64, dtype=torch.uint8, device="cuda")
    d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    want, wlens = oracle.encode_blocks(oracle.Order1(), data, bs, nthreads=8)
    comp, lens = ctx.encode_blocks(w3.Order1(), data, bs)
    d_comp = torch.from_numpy(comp.copy()).cuda()
    d_clens = torch.from_numpy(lens.astype(np.int32)).cuda()
    back = torch.empty(len(a), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def refused(fn):
        with pytest.raises(w3.W3Error) as e:
            fn()
        assert e.value.code == L.W3_E_INVALID, fn

    for level in ("device", "host"):
        if level == "device":
            job = ctx.encode_submit(w3.Order1(), d_in, bs, d_out, d_lens, d_total)
        else:
            h_out, h_lens = np.zeros(2 * len(a) + 4096, dtype=np.uint8), np.zeros(nb, dtype=np.uint32)
            job = ctx.encode_host_submit(w3.Order1(), a, bs, h_out, h_lens)
        refused(lambda: ctx.encode_blocks(w3.Order0(), data[:9000], bs))
        refused(lambda: ctx.encode_blocks_device(w3.Order0(), d_in, bs, d_out, d_lens, d_total))
        refused(lambda: ctx.decode_blocks(w3.Order1(), comp, lens, bs, len(a)))
        refused(lambda: ctx.decode_blocks_device(w3.Order1(), d_comp, d_clens, bs, len(a), back))
        refused(lambda: ctx.predict_blocks(w3.Order0(), data[:9000], bs))
        refused(lambda: ctx.encode_stats(w3.Order0(), data[:9000], bs))
        refused(lambda: ctx.export_counters(w3.Order0(), data[:9000]))
        refused(lambda: ctx.sweep_ordern(data[:9000], bs, [(11, 3)]))
        refused(lambda: ctx.compress(data[:9000]))
        refused(lambda: ctx.decompress(b"w30i" + (5).to_bytes(8, "big") + b"abcdefgh"))
        refused(lambda: ctx.set_variant("no_lds_atomics"))
        refused(lambda: ctx.set_coder("x3"))
        refused(lambda: ctx.set_path("generic"))
        ctx.set_timing(False)   # exempt: read when a call is submitted
        ctx.set_tune(0)         # exempt: scheduling only
        if level == "device":
            ctx.encode_wait(job)
            assert d_out[: int(d_total.item())].cpu().numpy().tobytes() == want.tobytes()
        else:
            total = ctx.encode_host_wait(job)
            assert h_out[:total].tobytes() == want.tobytes() and h_lens.tolist() == wlens.tolist()
    # and everything works again afterwards
    assert ctx.decode_blocks(w3.Order1(), comp, lens, bs, len(a)).tobytes() == data


def test_submit_of_a_synchronous_spec_reports_through_wait(ctx, oracle):
    """Specs that w3_encode_submit runs to completion inside the call (lane-per-block path): whatever the synchronous run returned —
    W3_E_NOSPACE with the need in d_total included — comes from w3_encode_wait, as for every other job."""
    import torch
    from weath3rb0i_amd import _lib as L
    bs = 1024
    data = markov_text(20 * 1024 + 9, seed=92)
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    nb = (len(data) + bs - 1) // bs
    want, wlens = oracle.encode_blocks(oracle.Order1(), data, bs, nthreads=8)
    small = torch.empty(500, dtype=torch.uint8, device="cuda")
    d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.set_path("generic")
    try:
        job = ctx.encode_submit(w3.Order1(), d_in, bs, small, d_lens, d_total)
        with pytest.raises(w3.W3Error) as e:
            ctx.encode_wait(job)
        assert e.value.code == L.W3_E_NOSPACE and int(d_total.item()) == len(want)
        big = torch.empty(2 * len(data) + 64 * nb + 64, dtype=torch.uint8, device="cuda")
        job = ctx.encode_submit(w3.Order1(), d_in, bs, big, d_lens, d_total)
        ctx.encode_wait(job)
        assert big[: int(d_total.item())].cpu().numpy().tobytes() == want.tobytes()
    finally:
        ctx.set_path("auto")


def test_variant_reset_reaches_every_job_slot(ctx, oracle):
    """W3_OPT_VARIANT resets the lane-order verdict of EVERY job slot: after set_variant("no_lds_atomics") a submit that lands on slot
    1..3 runs the ballot rounds too (its output is the oracle's either way; the timing says which path ran: the ballot rounds need no
    sampled verification, so n_lds_faults stays 0 and the streams match)."""
    import torch
    bs = 2048
    data = markov_text(100 * 1024 + 1, seed=93)
    d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    nb = (len(data) + bs - 1) // bs
    bufs = [(torch.empty(2 * len(data) + 64 * nb + 64, dtype=torch.uint8, device="cuda"), torch.zeros(nb, dtype=torch.int32, device="cuda"),
             torch.zeros(1, dtype=torch.int64, device="cuda")) for _ in range(4)]
    want, _ = oracle.encode_blocks(pair(oracle, "best012")[1](), data, bs, nthreads=8)
    # every slot sees the default path first
    jobs = [ctx.encode_submit(pair(oracle, "best012")[0](), d_in, bs, *bufs[k]) for k in range(4)]
    for j in jobs:
        ctx.encode_wait(j)
    ctx.set_variant("no_lds_atomics")
    try:
        jobs = [ctx.encode_submit(pair(oracle, "best012")[0](), d_in, bs, *bufs[k]) for k in range(4)]
        for k, j in enumerate(jobs):
            ctx.encode_wait(j)
            assert bufs[k][0][: int(bufs[k][2].item())].cpu().numpy().tobytes() == want.tobytes(), k
    finally:
        ctx.set_variant()
    jobs = [ctx.encode_submit(pair(oracle, "best012")[0](), d_in, bs, *bufs[k]) for k in range(4)]
    for k, j in enumerate(jobs):
        ctx.encode_wait(j)
        assert bufs[k][0][: int(bufs[k][2].item())].cpu().numpy().tobytes() == want.tobytes(), k
