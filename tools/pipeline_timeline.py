#!/usr/bin/env python3
"""Kernel timeline of a pipelined bench run (rocprofv3 --kernel-trace csv): every w3 kernel longer than 0.2 ms in a window of
the steady state, with its queue, so that one sees which kernels of step k+1 run beside which of step k.
usage: tools/pipeline_timeline.py <kernel_trace.csv> [window_ms=200]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "w3::" in r["Kernel_Name"]]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tend = int(rows[-1]["End_Timestamp"])
rows = [r for r in rows if int(r["Start_Timestamp"]) > tend - win * 1e6]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%-40s %8.2f -> %8.2f  (%6.2f ms)  queue %s" % (r["Kernel_Name"].split("(")[0][-40:], s, e, e - s, r.get("Queue_Id", "?")))
