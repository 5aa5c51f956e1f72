#!/bin/bash
DST=$PWD/gpurun_out/r3_fr; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cm.py tests/test_gpu_bench.py -x -q -m gpu -k "submit or bench" > "$DST/pytest.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 "$DST/pytest.txt"
[ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 12 --warmup 3 --no-other-configs --no-cpu-baseline "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['config']['encodes_in_flight'], d['kernel_ms_per_step'], d['decode']['value'], d['decode']['roundtrip_all_blocks'])"; }
run final_order0 --model order0
run final_default --model default
run final_order012 --model order012
