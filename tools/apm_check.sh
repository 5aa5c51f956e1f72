#!/bin/bash
# CM parity tests + a bench line of the default model (run through gpurun): tools/apm_check.sh <tag>
TAG=${1:-x}
DST=$PWD/gpurun_out/apm_$TAG
mkdir -p "$DST"
timeout -k 10 600 python3 -m pytest tests/test_gpu_cm.py -x -q > "$DST/pytest_cm.txt" 2>&1 || { tail -30 "$DST/pytest_cm.txt"; exit 1; }
tail -2 "$DST/pytest_cm.txt"
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --quick > "$DST/bench.json" 2> "$DST/bench.err" || { tail -5 "$DST/bench.err"; exit 1; }
python3 - "$DST/bench.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
