#!/bin/bash
# kernel timeline of two contexts encoding at the same time (tools/inflight_test.py under rocprofv3 --kernel-trace)
OUT=$PWD/gpurun_out/infl_tl; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python3 tools/inflight_test.py order012apm > "$OUT/log.txt" 2>&1
f=$(find "$OUT/kt" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$OUT/timeline.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "w3::" in r["Kernel_Name"] and any(k in r["Kernel_Name"] for k in ("k_apm0","k_coder_x4","k_rank_sorted","k_partition8","k_predict_small"))]
rows = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 1000000]   # the main launches (> 1 ms)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-60:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-34s %8.2f -> %8.2f (%6.2f ms) queue %s stream %s" % (r["Kernel_Name"].split("(")[0][-34:], s, e, e - s, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
PY
rm -rf "$OUT/kt"; tail -3 "$OUT/log.txt"; tail -40 "$OUT/timeline.txt"
