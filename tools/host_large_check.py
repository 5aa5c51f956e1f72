"""A host buffer above the per-device-call limit (4 GiB) through w3_encode_blocks / w3_decode_blocks: the calls go through in pieces.
Checks: round trip, and the first / last pieces' streams against separate calls on those block ranges.
python tools/host_large_check.py [model] [bytes]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import weath3rb0i_amd as w3
from tools import synth
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "order012apm"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_500_000_000
bs = 65536
model, mname = bench.make_model(w3, name)
part = synth.text(500_000_000, seed=9)
host = np.concatenate([part] * ((n + len(part) - 1) // len(part)))[:n]
host[::7919] ^= (np.arange(0, n, 7919, dtype=np.uint64) & 0xFF).astype(np.uint8)   # (the repeats differ)
ctx = w3.Context(0)
t0 = time.perf_counter()
out, lens = ctx.encode_blocks(model, host, bs)
t1 = time.perf_counter()
back = ctx.decode_blocks(model, out, lens, bs, n)
t2 = time.perf_counter()
ok = bool(np.array_equal(back, host))
offs = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
nb = len(lens)
same = True
for b0, b1 in ((0, 40), (nb - 40, nb), (32768 - 20, 32768 + 20)):   # (32,768 blocks = 2 GiB: a piece boundary)
    o2, l2 = ctx.encode_blocks(model, host[b0 * bs:min(n, b1 * bs)], bs)
    same &= l2.tolist() == lens[b0:b1].tolist() and o2.tobytes() == out[offs[b0]:offs[b1]].tobytes()
print({"model": mname, "bytes": n, "blocks": nb, "pieces": ctx.timing()["n_parts"], "encoded": int(offs[-1]), "encode_s": round(t1 - t0, 2), "decode_s": round(t2 - t1, 2),
       "round_trip_ok": ok, "ranges_equal_separate_calls": bool(same)})
