#!/usr/bin/env python3
"""Writes tests/golden/cm_streams.json: (length, sha256) of the oracle's stream for each build-defined CM
model on each seeded input.  The reference has no such models, so these vectors come from this
build's own CPU oracle — they pin the DEFINITION (a format break shows up as a diff), not reference parity."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as oracle  # noqa: E402
from tests.test_cm_cpu import GOLDEN, cm_model, golden_inputs  # noqa: E402

out = {}
for iname, data in golden_inputs().items():
    out[iname] = {}
    for mname in ("slot2", "apm0_order0", "o012_apm", "full_cm"):
        c = oracle.encode_stream(cm_model(oracle, mname), data)
        out[iname][mname] = [len(c), hashlib.sha256(bytes(c)).hexdigest()]
json.dump(out, open(GOLDEN, "w"), indent=1, sort_keys=True)
print("wrote", GOLDEN)
