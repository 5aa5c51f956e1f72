#!/bin/bash
# k_decode_spec under rocprofv3: kernel stats, then FETCH_SIZE and WRITE_SIZE in separate passes (tools/decode_rate.py: one encode + one decode of 1e9 B)
OUT=$PWD/gpurun_out/r3_decprof; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$OUT/.." && cd ..
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 tools/decode_rate.py order012apm 1e9 > "$OUT/run_stats.log" 2>&1
find "$OUT/kt" -name "*kernel_stats.csv" -exec cp {} "$OUT/decode_order012apm_kernel_stats.csv" \; ; rm -rf "$OUT/kt"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p" -o p -- python3 tools/decode_rate.py order012apm 1e9 > "$OUT/run_$C.log" 2>&1
  python3 - "$OUT" $C <<'PY'
import csv, glob, sys, collections
out, cnt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_decode_spec" not in k and "k_cm_nl" not in k: continue
        acc[k] += float(r["Counter_Value"]); n[k] += 1
with open(out + "/decode_order012apm_pmc_%s.txt" % cnt, "w") as f:
    for k in acc:
        line = "%s  %s = %.6g per launch (%d launch(es); gfx950: FETCH_SIZE in 64-byte... units as reported, see MI355X_MICROARCH.md)" % (k[-60:], cnt, acc[k] / max(n[k], 1), n[k])
        print(line); f.write(line + "\n")
PY
  rm -rf "$OUT/p"
done
tail -2 "$OUT/run_stats.log"; head -6 "$OUT/decode_order012apm_kernel_stats.csv"
