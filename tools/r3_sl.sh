#!/bin/bash
DST=$PWD/gpurun_out/r3_sl; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_cm.py -x -q -m gpu > "$DST/pytest.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 "$DST/pytest.txt"
[ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --quick "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['config']['encodes_in_flight'], d['kernel_ms_per_step'])"; }
run cfg4_p2 --model fullcm --data mixed --block-size 262144 --size 211938580
run cfg4_p1 --model fullcm --data mixed --block-size 262144 --size 211938580 --pipeline 1
run e8_p2 --model fullcm --size 100000000
run e8_p1 --model fullcm --size 100000000 --pipeline 1
