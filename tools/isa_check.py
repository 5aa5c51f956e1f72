#!/usr/bin/env python3
"""Instruction mix of libw3hip.so's kernels from the device assembly (hipcc -S): flat_/scratch_ accesses (a pointer that lost
its address space, register spills) are what to look for after touching a kernel.  Usage: tools/isa_check.py [substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pats = sys.argv[1:]
    out = "/tmp/_w3.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                           os.path.join(ROOT, "weath3rb0i_amd", "csrc", "w3hip.hip")] + os.environ.get("W3_EXTRA_FLAGS", "").split(), cwd="/tmp")
    txt = open(out).read()
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", txt)
    names, bodies = parts[1::2], parts[2::2]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for body, d in zip(bodies, dem):
        d = re.sub(r"^void ", "", d).split("(")[0]
        if "::k_" not in d or (pats and not any(p in d for p in pats)):
            continue
        body = body.split("s_endpgm")[0]
        ins = [ln.split()[0] for ln in body.splitlines() if ln.startswith("\t") and ln.strip() and not ln.startswith("\t.") and not ln.strip().startswith(";")]
        cnt = lambda pre: sum(1 for i in ins if i.startswith(pre))
        print("%-46s instr %6d  valu %6d  salu %5d  ds %5d  global %4d  flat %3d  scratch %3d  waitcnt %4d" % (
            d[:46], len(ins), cnt("v_"), cnt("s_") - cnt("s_waitcnt"), cnt("ds_"), cnt("global_"), cnt("flat_"), cnt("scratch_"), cnt("s_waitcnt")))


if __name__ == "__main__":
    main()
