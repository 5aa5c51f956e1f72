import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import weath3rb0i_amd as w3
from tests.synth import markov_text
ctx = w3.Context(0)
data = markov_text(30000, seed=41)
for name, mk in [("s2+o0", lambda: w3.BestOfTwoModel(w3.SlotModel(2, 12), w3.Order0())),
                 ("o0+s2", lambda: w3.BestOfTwoModel(w3.Order0(), w3.SlotModel(2, 12))),
                 ("s2+s1", lambda: w3.BestOfTwoModel(w3.SlotModel(2, 12), w3.SlotModel(1, 12))),
                 ("mix", lambda: w3.BestOfTwoModel(w3.SlotModel(2, 12), w3.BestOfTwoModel(w3.Order0(), w3.SlotModel(1, 12))))]:
    print(name, flush=True)
    out, lens = ctx.encode_blocks(mk(), data, 8192)
    print(name, len(out), flush=True)
