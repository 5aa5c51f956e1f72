"""In-tree build of the HIP extension (libw3hip.so) for gfx950 with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libw3hip.so")
SRC = os.path.join(HERE, "csrc", "w3hip.hip")


def sources():
    d = os.path.join(HERE, "csrc")
    out = [os.path.join(d, f) for f in sorted(os.listdir(d))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "w3hip.h"))
    return out


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=False, out=None, extra=()):
    """out / extra: an experimental build under another name (loaded with W3HIP_SO=<path>), e.g. out="libw3hip_nt.so", extra=["-DW3_X"]"""
    if out is None and not force and not stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    target = os.path.join(HERE, out) if out else SO
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-o", target, SRC]
    cmd += list(extra)
    cmd += os.environ.get("W3_EXTRA_FLAGS", "").split()   # e.g. -DW3_TUNING: tuning hooks + the APM kernels' store guard (debug builds)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    import sys
    argv = sys.argv[1:]
    out = None
    if "--out" in argv:
        i = argv.index("--out")
        out = argv[i + 1]
        del argv[i:i + 2]
    build(force=True, verbose=True, out=out, extra=argv)
