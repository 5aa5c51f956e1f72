"""ctypes wrapper of tools/synth.c (seeded synthetic corpora for bench.py / full-size tests)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libw3synth.so")


def build():
    src = os.path.join(HERE, "synth.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", SO, src, "-lpthread"])
    return SO


def _lib():
    lib = C.CDLL(build())
    for f in (lib.w3s_text, lib.w3s_mixed):
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int]
        f.restype = None
    return lib


def text(n, seed=1, chunk0=0, nthreads=None):
    out = np.empty(n, dtype=np.uint8)
    _lib().w3s_text(out.ctypes.data_as(C.c_void_p), n, seed, chunk0, nthreads or os.cpu_count() or 1)
    return out


def mixed(n, seed=1, chunk0=0, nthreads=None):
    out = np.empty(n, dtype=np.uint8)
    _lib().w3s_mixed(out.ctypes.data_as(C.c_void_p), n, seed, chunk0, nthreads or os.cpu_count() or 1)
    return out
