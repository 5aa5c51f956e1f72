"""Timing experiment driver: runs the predict kernels only (W3_DEBUG_NOSTORE=1 makes encode stop before the coder)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import weath3rb0i_amd as w3
from tools import synth
n = 1_000_000_000
ctx = w3.Context(0)
d_in = torch.from_numpy(synth.text(n, seed=1)).cuda()
nb = (n + 65535) // 65536
d_out = torch.empty(n, dtype=torch.uint8, device="cuda"); d_lens = torch.zeros(nb, dtype=torch.int32, device="cuda"); d_total = torch.zeros(1, dtype=torch.int64, device="cuda")
m = w3.BestOfTwoModel(w3.BestOfTwoModel(w3.Order0(), w3.Order1()), w3.OrderN(27, 3))
for _ in range(3):
    try:
        ctx.encode_blocks_device(m, d_in, 65536, d_out, d_lens, d_total)
    except w3.W3Error as e:
        pass
torch.cuda.synchronize()
print("done")
