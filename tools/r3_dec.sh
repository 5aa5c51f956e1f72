#!/bin/bash
# k_decode_spec: the decode tests, then decode rates (bench.py's decode key) of a few models / sizes
DST=$PWD/gpurun_out/r3_dec; mkdir -p $DST
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cm.py -x -q -m gpu > "$DST/pytest.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 "$DST/pytest.txt"
[ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift
  timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ref-model --no-other-configs "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['decode'])"; }
run dec_default
run dec_default_lane --variant decode_lane
run dec_e8 --size 100000000
run dec_o012 --model order012
run dec_o0 --model order0
run dec_main --model default
