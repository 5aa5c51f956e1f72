import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import weath3rb0i_amd as w3
from oracle import pyoracle as orc
from tests.synth import markov_text, lcg_text
data = markov_text(70000, seed=12) + lcg_text(9000, seed=3) + bytes(3000) + markov_text(5000, seed=13)
ctx = w3.Context(0)
for bits in (3, 4, 5, 11):
    want, wl = orc.encode_blocks(orc.OrderN(bits, 3), data, 16384, nthreads=8)
    p = ctx.predict_blocks(w3.OrderN(bits, 3), data[:40000], 16384)
    wp = np.concatenate([orc.predict_all(orc.OrderN(bits, 3), data[o:min(o + 16384, 40000)]) for o in range(0, 40000, 16384)])
    bad = np.nonzero(p != wp)[0]
    print("bits", bits, "predict mismatches", len(bad), bad[:8], p[bad[:4]], wp[bad[:4]])
    ctx.set_path("twophase")
    for mode in ("x3", "x2", "fast", "robust"):
        ctx.set_coder(mode)
        out, lens = ctx.encode_blocks(w3.OrderN(bits, 3), data, 16384)
        print("   coder", mode, "ok" if (lens.tolist() == wl.tolist() and out.tobytes() == want.tobytes()) else "MISMATCH")
    ctx.set_coder("x3"); ctx.set_path("auto")
