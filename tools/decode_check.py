"""Round-trip diagnosis on device-resident data: encode [bytes] of synthetic text with [model], decode it with k_decode_spec and with the
lane-per-block decoders, and name the blocks that do not come back (count, first few indices, first differing offset inside the block).
python tools/decode_check.py [model] [bytes] [tune]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import weath3rb0i_amd as w3
from tools import synth
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "ac20"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
tune = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bs = 65536
nb = (n + bs - 1) // bs
model, mname = bench.make_model(w3, name)
host = synth.text(n, seed=1)
d_in = torch.from_numpy(host).cuda()
d_out = torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda")
d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
enc = w3.Context(0)
enc.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total)
torch.cuda.synchronize()
print({"model": mname, "bytes": n, "blocks": nb, "encoded": int(d_total.item())})


def report(tag, d_back):
    pad = nb * bs - n
    a = torch.nn.functional.pad(d_back, (0, pad)).view(nb, bs)
    b = torch.nn.functional.pad(d_in, (0, pad)).view(nb, bs)
    ne = a != b
    bad = ne.any(dim=1).nonzero().flatten()
    first = [int(ne[i].nonzero()[0]) for i in bad[:8].tolist()]
    print({"decoder": tag, "bad_blocks": int(bad.numel()), "first_bad": bad[:8].tolist(), "first_offset_in_block": first})


for tag, variant in (("k_decode_spec", None), ("lane", "decode_lane")):
    ctx = w3.Context(0)
    if variant:
        ctx.set_variant(variant)
    if tune:
        ctx.set_tune(tune)
    d_back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ctx.decode_blocks_device(model, d_out, d_lens, bs, n, d_back)
    torch.cuda.synchronize()
    report(tag, d_back)
    del ctx
