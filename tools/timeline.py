#!/usr/bin/env python3
"""Start/end of every w3 kernel of the LAST bench step in a rocprofv3 kernel trace, relative to the step's first kernel.
usage: tools/timeline.py gpurun_out/kt_<tag>/kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "w3::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last first-kernel-of-a-step on: steps begin with the earliest kernel after a k_pack
last_pack = max((i for i, r in enumerate(rows[:-1]) if "k_pack" in r["Kernel_Name"]), default=-1)
step = rows[last_pack + 1:] if last_pack + 1 < len(rows) else rows
prev_packs = [i for i, r in enumerate(rows) if "k_pack" in r["Kernel_Name"]]
if len(prev_packs) >= 2:
    step = rows[prev_packs[-2] + 1:prev_packs[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-44s %8.2f -> %8.2f  (%6.2f ms)  stream %s" % (r["Kernel_Name"].split("(")[0][-44:], s, e, e - s, r.get("Stream_Id", r.get("Queue_Id", "?"))))
