"""Turns the FETCH_SIZE / WRITE_SIZE summaries of `tools/gpu.sh <tag> traffic <model>` into entries of profiles/r*_traffic.json:
HBM traffic per STEP of every w3 kernel = 2 x FETCH_SIZE + WRITE_SIZE (KB = 1024 B; FETCH doubled per the gfx950 note of
MI355X_MICROARCH.md), summed over the kernel's launches and divided by the steps the profiled bench run made (warm-up and priming
included: every launch is counted, so is every step).
usage: tools/traffic_json.py <model> <dir> [bench args of the profiled run]"""
import json
import os
import re
import sys


def main():
    model, dst = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    size = 1_000_000_000
    bs = 65536
    for i, a in enumerate(rest):
        if a == "--size":
            size = int(float(rest[i + 1]))
        if a == "--block-size":
            bs = int(rest[i + 1])
    f = json.load(open(os.path.join(dst, "%s_pmc_FETCH_SIZE.json" % model)))
    w = json.load(open(os.path.join(dst, "%s_pmc_WRITE_SIZE.json" % model)))
    # steps the run made = launches of the coder kernel (one per step)
    steps = max([v["launches"] for k, v in f.items() if "k_coder" in k] or [1])
    entries = []
    for k in sorted(f):
        short = re.sub(r"^.*w3::", "", k)
        short = re.sub(r"void ", "", short)
        fk = f[k]["per_launch"].get("FETCH_SIZE", 0.0) * f[k]["launches"] / steps
        wk = w.get(k, {"per_launch": {}, "launches": 0})
        wv = wk["per_launch"].get("WRITE_SIZE", 0.0) * wk["launches"] / steps
        tb = int((2 * fk + wv) * 1024)
        if tb < (1 << 20):
            continue
        entries.append({"config": {"model": model, "bytes_per_gpu": size, "block_size": bs}, "kernel": short,
                        "launches_per_step": round(f[k]["launches"] / steps, 2),
                        "fetch_size_kb_per_step": fk, "write_size_kb_per_step": wv, "traffic_bytes_per_step": tb,
                        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --quick --steps 2 --warmup 1 --model %s %s` (%d steps counted; --pmc serialises the kernels)" % (model, " ".join(rest), steps)})
        print("%-46s %8.2f GB per step (fetch %.2f x 2, write %.2f)" % (short[:46], tb / 1e9, fk * 1024 / 1e9, wv * 1024 / 1e9))
    json.dump({"entries": entries}, open(os.path.join(dst, "traffic_%s.json" % model), "w"), indent=1)


if __name__ == "__main__":
    main()
