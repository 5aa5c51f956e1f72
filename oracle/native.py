"""-O3 -march=native build of the CPU oracle for the host that TIMES it (bench.py's cpu_baseline leg; BASELINE.md section 2 /
SURVEY section 8(d): "same algorithm, -O3 -march=native").  TEST INFRASTRUCTURE, like everything under oracle/.
Kept apart from pyoracle.py, which loads a library at import: bench.py calls build_native() first and points W3_ORACLE_SO at
the result.  The checker of tests/ and smoke() stays the portable build (it is made in the build container and travels)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_native():
    """-> path of oracle/_native/<cpu hash>/libw3oracle.so, or None when it cannot be built here.
    The output directory is keyed by the host CPU (a build made on another machine may travel with the repo snapshot)."""
    import hashlib
    try:
        cpu = "".join(ln for ln in open("/proc/cpuinfo") if ln.startswith(("model name", "flags")))[:20000]
    except OSError:
        cpu = ""
    d = os.path.join(_HERE, "_native", hashlib.sha1(cpu.encode()).hexdigest()[:12])
    so = os.path.join(d, "libw3oracle.so")
    if os.path.exists(so):
        return so
    try:
        os.makedirs(d, exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-D_GNU_SOURCE", "-shared", "-o", so,
                               os.path.join(_HERE, "w3_oracle.c"), "-lpthread", "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    return so if os.path.exists(so) else None


