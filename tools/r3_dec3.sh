#!/bin/bash
DST=$PWD/gpurun_out/r3_dec; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu -k "generic_path or twophase_path or edge or wave_per or random" > "$DST/pytest3.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 "$DST/pytest3.txt"
[ $rc -ne 0 ] && exit $rc
for t in 16384 32768; do
python3 - <<P
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import weath3rb0i_amd as w3
from tools import synth
import bench
for name, n in (("order012apm", 10**9), ("order012apm", 10**8), ("order012", 10**9), ("order0", 10**9), ("default", 10**9)):
    bs = 65536; nb = (n + bs - 1) // bs
    model, mname = bench.make_model(w3, name)
    ctx = w3.Context(0)
    host = synth.text(n, seed=1)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.empty(n + n // 4 + 64 * nb + 1024, dtype=torch.uint8, device="cuda")
    d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda"); d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.encode_blocks_device(model, d_in, bs, d_out, d_lens, d_total)
    d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.set_tune($t)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.decode_blocks_device(model, d_out, d_lens, bs, n, d_back)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("tune $t", name, n, "decode %.3f s = %.1f MiB/s" % (dt, n / dt / 2**20), "ok" if bool(torch.equal(d_back, d_in)) else "MISMATCH", flush=True)
    ctx.close(); del d_in, d_out, d_back; torch.cuda.empty_cache()
P
done
