#!/bin/bash
# sorted slot replay (w3_slot2.h): the CM tests, then the shapes it is for against k_slot (W3_OPT_VARIANT 256 via --variant)
DST=$PWD/gpurun_out/r3_s2; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_cm.py -x -q -m gpu > "$DST/pytest.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 "$DST/pytest.txt"
[ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --quick "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"; }
run cfg4_sorted --model fullcm --data mixed --block-size 262144 --size 211938580
run cfg4_table --model fullcm --data mixed --block-size 262144 --size 211938580 --variant slot_table
run e8_sorted --model fullcm --size 100000000
run e8_table --model fullcm --size 100000000 --variant slot_table
run e8x4_sorted --model fullcm --size 400000000
run e8x4_table --model fullcm --size 400000000 --variant slot_table
