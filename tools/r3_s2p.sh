#!/bin/bash
# kernel stats of the sorted slot replay: tools/r3_s2p.sh <tag> [bench args]
TAG=$1; shift
OUT=$PWD/gpurun_out/s2p_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py --steps 3 --warmup 1 --quick "$@" > "$OUT/bench.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/kt"
python3 - <<P
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
for r in rows[:14]:
    print("%-60s calls %4s avg %10.3f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"]))
P
