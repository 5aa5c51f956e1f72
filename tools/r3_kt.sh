#!/bin/bash
# kernel trace + timeline of a pipelined bench run: tools/r3_kt.sh <tag> [bench args...]
TAG=$1; shift
OUT=$PWD/gpurun_out/kt_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py --steps 6 --warmup 2 --quick "$@" > "$OUT/bench.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name "*kernel_trace.csv" -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/kt"
python3 tools/pipeline_timeline.py "$OUT/kernel_trace.csv" 260 > "$OUT/timeline.txt"
grep '^{' "$OUT/bench.log" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
rm -f "$OUT/kernel_trace.csv"
cat "$OUT/timeline.txt"
