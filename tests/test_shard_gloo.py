"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the sharding + exchange step.
The per-rank producer here is the CPU oracle (there is no GPU in this container); on
the GPU box bench.py feeds the same gather_streams() with the HIP path's output."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.synth import lcg_text


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, data, bs, q, async_op=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pyoracle as orc
    from weath3rb0i_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.byte_range(rank, world, len(data), bs)
    out, lens = orc.encode_blocks(orc.Order0(), data[lo:hi], bs)
    stream = torch.from_numpy(np.concatenate([out, np.zeros(7, dtype=np.uint8)]))  # slack past `total` must be ignored
    tl = torch.from_numpy(lens.astype(np.int32))
    if async_op:   # split-phase form used by bench.py to overlap the exchange with the next encode
        allb, alll, totals, reqs = shard.gather_streams(stream, len(out), tl, dst=0, async_op=True)
        shard.wait_all(reqs)
    else:
        allb, alll, totals = shard.gather_streams(stream, len(out), tl, dst=0)
    if rank == 0:
        q.put((allb.numpy().tobytes(), alll.numpy().astype(np.uint32).tolist(), totals))
    else:
        assert allb is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,bs,async_op", [(2, 70000, 8192, False), (3, 20000, 4096, False), (2, 4096, 8192, False), (3, 5, 4096, False),
                                                 (2, 70000, 8192, True), (3, 5, 4096, True)])
def test_gather_matches_single_process(oracle, world, n, bs, async_op):
    data = lcg_text(n, seed=77)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, data, bs, q, async_op)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want, wlens = oracle.encode_blocks(oracle.Order0(), data, bs)
    assert got[0] == want.tobytes()
    assert got[1] == wlens.tolist()
    assert sum(got[2]) == len(want)


def test_block_ranges_cover_exactly():
    from weath3rb0i_amd import shard
    for world in (1, 2, 3, 8):
        for nb in (0, 1, 7, 8, 9, 1526, 15259):
            r = [shard.block_range(k, world, nb) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == nb
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
