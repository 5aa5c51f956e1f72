"""Decode throughput (k_decode_spec, or the lane-per-block decoders with W3 variant decode_lane: argv[3] = "lane") on device-resident data:
python tools/decode_rate.py [model] [bytes]   (decode MiB/s figure; not the north-star metric)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import weath3rb0i_amd as w3
from tools import synth
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "order012apm"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
bs = 65536
nb = (n + bs - 1) // bs
model, mname = bench.make_model(w3, name)
ctx = w3.Context(0)
if len(sys.argv) > 3 and sys.argv[3] == "lane":
    ctx.set_variant("decode_lane")
if len(sys.argv) > 3 and sys.argv[3].isdigit():
    ctx.set_tune(int(sys.argv[3]))   # W3_OPT_TUNE: 131072 = exact maps probed slot by slot, 262144 = APM tables row-major (k_decode_spec's table formats)
host = synth.text(n, seed=1)
d_in = torch.from_numpy(host).cuda()
d_out = torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda")
d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda")
d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
ctx.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total)
d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
for it in range(2):   # (the first call also allocates the model tables: the second is the rate)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.decode_blocks_device(model, d_out, d_lens, bs, n, d_back)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
ok = bool(torch.equal(d_back, d_in))
print({"model": mname, "bytes": n, "decode_ms": round(dt * 1e3, 1), "decode_MiB_s": round(n / dt / 2**20, 1), "round_trip_ok": ok})
