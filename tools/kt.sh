#!/bin/bash
# kernel-trace only: tools/kt.sh <tag> [bench args...]  ->  gpurun_out/kt_<tag>/kernel_stats.csv
TAG=$1; shift
OUT=$PWD/gpurun_out/kt_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name "*kernel_trace.csv" -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/kt"
python3 - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_trace.csv")))
for r in rows:
    n = r["Kernel_Name"]
    if "w3::" in n:
        print("%-40s grid %-10s %9.3f ms" % (n.split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
