"""Full-size runs of the BASELINE.json configurations on the GPU, checked through size-independent properties
(the oracle cannot code 1e9 bytes in test time):

* every block is coded independently with a fresh model and coder (SURVEY §8 A19 i), so
  - sampled blocks must equal the CPU oracle's stream for that block's bytes alone,
  - encoding a sub-range of the blocks alone must reproduce exactly the slices of the full run,
  - decoding those slices must give the input back (encode -> decode round trip through the C ABI);
* the length table and the total must agree, and no block stream is empty (flush emits >= 1 byte, io.rs:91-100).

Inputs: tools/synth.c (seeded enwik-shaped text / Silesia-shaped mix; no corpora exist in this pipeline)."""
import numpy as np
import pytest

import weath3rb0i_amd as w3
from tools import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = w3.Context(0)
    yield c
    c.close()


def o012(m):
    return m.BestOfTwoModel(m.BestOfTwoModel(m.Order0(), m.Order1()), m.OrderN(27, 3))


def models(oracle, name):
    if name == "order012apm":      # BASELINE configs[1]
        return w3.APM(o012(w3)), oracle.APM(o012(oracle))
    if name == "fullcm":           # BASELINE configs[2]
        t = o012(oracle)
        for order in (1, 2, 3, 4):
            t = oracle.BestOfTwoModel(t, oracle.SlotModel(order, 14))
        return w3.full_cm(), oracle.APM(oracle.APM(t, 0, 7), 1, 6)
    if name == "ac20":             # bin/entropy-hashing-ac (book1.log:139): one wide leaf keyed by ACHistory hashes (k_achash32 -> k_predict_wave)
        book1 = [1, 50188, 62497, 15819, 22545, 31499, 22988, 29616]
        return (w3.OrderNEntropy(20, 3, w3.ACHistory(17, w3.StationaryModel.for_book1())),
                oracle.OrderNEntropy(20, 3, oracle.ACHistory(17, oracle.StationaryModel.from_table(book1))))
    if name == "huff19":           # the same with HuffHistory keys (k_huffkeys<true>)
        from tests.test_gpu_parity import huff_pair
        dev_h, orc_t = huff_pair(oracle, "text")
        return w3.OrderNEntropy(19, 3, dev_h), oracle.OrderNEntropy(19, 3, oracle.HuffHistory(tables=orc_t))
    return o012(w3), o012(oracle)


def encode_device(ctx, model, host, bs):
    import torch
    n = len(host)
    nb = (n + bs - 1) // bs
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda")
    d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
    d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ctx.timing()["path"] == 2
    lens = d_lens.cpu().numpy().astype(np.int64)
    total = int(d_total.item())
    return d_in, d_out, lens, total


def check_properties(ctx, oracle, name, host, bs, sample, sub):
    import torch
    dev, orc = models(oracle, name)
    n = len(host)
    nb = (n + bs - 1) // bs
    d_in, d_out, lens, total = encode_device(ctx, dev, host, bs)
    assert len(lens) == nb and int(lens.sum()) == total and int(lens.min()) >= 1
    offs = np.concatenate([[0], np.cumsum(lens)])
    # sampled blocks against the oracle run on that block's bytes alone
    for b in sample:
        blk = host[b * bs:min((b + 1) * bs, n)]
        want, wl = oracle.encode_blocks(orc, blk.tobytes(), bs)
        got = d_out[int(offs[b]):int(offs[b + 1])].cpu().numpy()
        assert len(got) == int(wl[0]) and got.tobytes() == want.tobytes(), (name, b)
    # a sub-range of blocks encoded alone reproduces the slices of the full run, and decodes back to the input
    b0, b1 = sub
    part = host[b0 * bs:min(b1 * bs, n)]
    out2, lens2 = ctx.encode_blocks(dev, part, bs)
    assert lens2.astype(np.int64).tolist() == lens[b0:b1].tolist(), name
    full_slice = d_out[int(offs[b0]):int(offs[b1])].cpu().numpy()
    assert out2.tobytes() == full_slice.tobytes(), name
    back = ctx.decode_blocks(dev, full_slice, lens[b0:b1].astype(np.uint32), bs, len(part))
    assert back.tobytes() == part.tobytes(), name
    del d_in, d_out
    torch.cuda.empty_cache()
    return total / n


@pytest.mark.parametrize("name", ["order012apm", "fullcm"])
def test_enwik8_size_64k_blocks(ctx, oracle, name):
    """BASELINE configs[1] / configs[2]: 1e8 bytes of enwik-shaped text, 64 KiB blocks (1,526 blocks, ragged last one)."""
    host = synth.text(100_000_000, seed=3)
    nb = (len(host) + 65535) // 65536
    ratio = check_properties(ctx, oracle, name, host, 65536, [0, 1, 700, nb - 2, nb - 1], (512, 512 + 24))
    assert 0.2 < ratio < 0.6


def test_enwik9_size_default_model(ctx, oracle):
    """BASELINE configs[3] per-GPU shard = the bench workload: 1e9 bytes, 64 KiB blocks (15,259 blocks), configs[1]'s model."""
    host = synth.text(1_000_000_000, seed=1)
    nb = (len(host) + 65535) // 65536
    ratio = check_properties(ctx, oracle, "order012apm", host, 65536, [0, 4097, 9999, nb - 1], (15000, 15000 + 48))
    assert 0.2 < ratio < 0.6


def test_silesia_size_256k_blocks_full_cm(ctx, oracle):
    """BASELINE configs[4] shape: 211,938,580 bytes of mixed text/binary, 256 KiB blocks (809 blocks), the full CM
    (hash-map stress: 2^19 nibble contexts per leaf and block, incompressible and all-zero stretches)."""
    host = synth.mixed(211_938_580, seed=2)
    nb = (len(host) + 262143) // 262144
    check_properties(ctx, oracle, "fullcm", host, 262144, [0, 403, nb - 1], (200, 204))


@pytest.mark.parametrize("name", ["ac20", "huff19"])
def test_keyed_leaf_beyond_2_pow_32_steps(ctx, oracle, name):
    """More than 2^32 bit-steps in one call (n > 2^29 bytes) with a leaf whose contexts are precomputed per step: a dispatch counts its
    work-items in 32 bits, and k_achash32 launched as one thread per step covered only 8 n mod 2^32 of them until round 4 (every step
    after those coded with stale keys: streams that still decoded to garbage only, found by a 1e9-byte decode run).  Sampled blocks on
    both sides of the 2^32nd step against the oracle, a sub-range re-encoded alone, and the round trip of that range."""
    n = (1 << 29) + 5 * 65536 + 123
    host = synth.text(n, seed=5)
    nb = (n + 65535) // 65536
    wrap = (8 * n - (1 << 32)) // 8 // 65536   # the block holding step 8 n mod 2^32: the first one the old launch left out
    ratio = check_properties(ctx, oracle, name, host, 65536, [0, wrap - 1, wrap, wrap + 1, 4099, nb - 2, nb - 1], (8000, 8000 + 24))
    assert 0.2 < ratio < 0.7
