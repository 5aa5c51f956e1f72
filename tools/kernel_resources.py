#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of libw3hip.so's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python3 tools/kernel_resources.py [substring ...]  ->  one line per kernel whose demangled name holds a substring."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pats = sys.argv[1:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_w3res.so", os.path.join(ROOT, "weath3rb0i_amd", "csrc", "w3hip.hip")]
    cmd += os.environ.get("W3_EXTRA_FLAGS", "").split()
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = {}, None
    keys = {"SGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
            "LDS Size [bytes/block]": "lds"}
    for line in txt.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        for k, short in keys.items():
            m = re.search(r"remark: .*\s" + re.escape(k) + r": (\d+)", line)
            if m and cur:
                rows[cur][short] = int(m.group(1))
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    for (k, v), name in zip(rows.items(), names):
        name = re.sub(r"^void ", "", name).split("(")[0]
        if not pats or any(p in name for p in pats):
            print("%-44s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d occ %d" % (name[:44], v.get("vgpr", -1), v.get("agpr", -1), v.get("sgpr", -1),
                                                                            v.get("lds", -1), v.get("scratch", -1), v.get("occ", -1)))


if __name__ == "__main__":
    main()
