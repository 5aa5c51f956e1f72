#!/bin/bash
# one rocprofv3 PMC pass with an arbitrary counter list: tools/pmc.sh <tag> "<counters>" [bench args...]
TAG=$1; CNT=$2; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/p" -o p -- python3 bench.py --steps 1 --warmup 1 --quick "$@" > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "w3::" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == list(acc[k].keys())[0]: n[k] += 1
for k, d in acc.items():
    print(k[-45:], "launches", n[k])
    for c, v in d.items(): print("    %-28s %.4g per launch" % (c, v / max(n[k], 1)))
PY
rm -rf "$OUT/p"
