#!/bin/bash
# rocprofv3 evidence for bench.py's default workload (run on the GPU box through gpurun):
#   tools/profile.sh <tag> [bench args...]
# 1. kernel trace + stats  2. PMC FETCH_SIZE  3. PMC WRITE_SIZE   (separate passes, as MI355X_MICROARCH.md prescribes)
# Summaries land in gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --quick $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 bench.py $ARGS > "$OUT/bench_kt.log" 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o fetch -- python3 bench.py $ARGS > "$OUT/bench_fetch.log" 2>&1
echo "FETCH_SIZE done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o write -- python3 bench.py $ARGS > "$OUT/bench_write.log" 2>&1
echo "WRITE_SIZE done"
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for name in ("fetch", "write"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(out + "/" + name + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    res[name] = {k: {"sum": v[0], "launches": v[1], "per_launch": v[0] / max(v[1], 1)} for k, v in acc.items() if "w3::" in k}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf "$OUT/kt" ; find "$OUT/fetch" "$OUT/write" -name "*.csv" ! -name "*counter_collection.csv" -delete
