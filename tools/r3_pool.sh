#!/bin/bash
DST=$PWD/gpurun_out/r3_pool; mkdir -p $DST
python3 - <<P > $DST/out.txt 2>&1
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import weath3rb0i_amd as w3
import bench
env = {"rank": 0, "world": 1, "local_rank": 0, "exchange": False, "host_staged": False}
def one(tag, model="order012apm", size=100_000_000, kind="text", bs=65536, steps=12):
    r = bench.short_run(w3, model, kind, size, bs, 1, steps, env)
    print(tag, r["value"], r["ms_per_step"], r["pipeline"], flush=True)
one("first ctx e8")
one("second ctx e8")
one("fullcm cfg4", "fullcm", 211938580, "mixed", 262144, 4)
one("after fullcm e8")
one("fullcm e8", "fullcm", 100_000_000, "text", 65536, 6)
one("again e8")
P
cat $DST/out.txt | tail -8
