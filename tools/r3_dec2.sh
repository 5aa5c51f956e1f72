#!/bin/bash
DST=$PWD/gpurun_out/r3_dec; mkdir -p $DST
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "generic_path or twophase_path or edge or wave_per" > "$DST/pytest2.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 "$DST/pytest2.txt"
[ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift
  timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-model --no-other-configs "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['decode'])"; }
run dec2_default
run dec2_e8 --size 100000000
run dec2_o012 --model order012
