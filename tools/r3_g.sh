#!/bin/bash
# deeper submit / wait pipeline: tests, then small-input bench lines at 1, 2 and 4 encodes in flight
DST=$PWD/gpurun_out/r3_g; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cm.py tests/test_gpu_bench.py -x -q -m gpu -k "submit or bench" > "$DST/pytest.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 "$DST/pytest.txt"
[ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 16 --warmup 4 --quick "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['config']['encodes_in_flight'], d['kernel_ms_per_step'])"; }
for p in 1 2 4; do run e8_p$p --size 100000000 --pipeline $p; done
for p in 1 2 4; do run s8_p$p --size 125000000 --pipeline $p; done
for p in 2 4; do run e7_p$p --size 10000000 --pipeline $p; done
run o0_e8_p4 --size 100000000 --model order0
run o012_e8_p4 --size 100000000 --model order012
run full_default --steps 10 --warmup 3
