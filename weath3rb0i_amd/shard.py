"""Multi-GPU sharding of the block path: one process per GPU, no data-path collective
during encode (blocks are independent: fresh model + coder per block), then ONE
exchange step that concatenates the per-GPU packed streams on rank 0:

  1. all-gather of (total bytes, block count) per rank          (tiny)
  2. variable-length gather: every rank r>0 sends its packed stream and its block
     length table straight to rank 0 (grouped send/recv; RCCL has no gatherv).
     On xGMI every peer has its own link to the root, so the 7 transfers overlap.

Works with any torch.distributed backend: "nccl" (= RCCL) on GPU tensors, "gloo"
on CPU tensors (used by the CPU tests).  The reference has no counterpart (it is
single-process, Cargo.toml:14-15); the container this produces is the build-defined
block container of DESIGN.md.
"""
import torch
import torch.distributed as dist


def block_range(rank, world, nblocks):
    """Contiguous block range of `rank` (block b -> rank floor(b*world/nblocks)): rank order == stream order."""
    return rank * nblocks // world, (rank + 1) * nblocks // world


def byte_range(rank, world, n, block_size):
    nb = (n + block_size - 1) // block_size
    lo, hi = block_range(rank, world, nb)
    return min(lo * block_size, n), min(hi * block_size, n)


def gather_streams(stream, total, lens, dst=0, group=None, out=None, async_op=False):
    """Concatenate every rank's packed stream (stream[:total], uint8) and block lengths (lens, int32) on `dst`.

    Returns (all_streams, all_lens, totals) on dst — views into `out` when given — and (None, None, totals) elsewhere.
    async_op=True returns a 4th element, the outstanding point-to-point requests: the transfers then overlap whatever the
    caller enqueues next (RCCL runs them on its own stream); call wait_all() on them before touching `stream`, `lens` or
    the returned views again.
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = stream.device
    meta = torch.tensor([int(total), int(lens.numel())], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    totals = [int(m[0]) for m in metas]
    counts = [int(m[1]) for m in metas]
    if rank != dst:
        ops = []
        if totals[rank]:
            ops.append(dist.P2POp(dist.isend, stream[: totals[rank]], dst, group))
        if counts[rank]:
            ops.append(dist.P2POp(dist.isend, lens, dst, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if async_op:
            return None, None, totals, reqs
        wait_all(reqs)
        return None, None, totals
    need = sum(totals)
    if out is None or out.numel() < need:
        out = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    all_lens = torch.empty(max(sum(counts), 1), dtype=lens.dtype, device=dev)
    ops, so, lo = [], 0, 0
    for r in range(world):
        if r == dst:
            out[so: so + totals[r]].copy_(stream[: totals[r]])
            all_lens[lo: lo + counts[r]].copy_(lens)
        else:
            if totals[r]:
                ops.append(dist.P2POp(dist.irecv, out[so: so + totals[r]], r, group))
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, all_lens[lo: lo + counts[r]], r, group))
        so += totals[r]
        lo += counts[r]
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return out[:need], all_lens[: sum(counts)], totals, reqs
    wait_all(reqs)
    return out[:need], all_lens[: sum(counts)], totals


def wait_all(reqs):
    for req in reqs:
        req.wait()
