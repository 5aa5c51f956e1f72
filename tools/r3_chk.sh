#!/bin/bash
DST=$PWD/gpurun_out/r3_chk; mkdir -p $DST
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['config']['encodes_in_flight'], d['kernel_ms_per_step'])"; }
run e8_a --size 100000000 --steps 16 --warmup 4 --quick
run e8_b --size 100000000 --steps 12 --warmup 1 --quick
run e8_c --size 100000000 --steps 12 --warmup 1 --quick --no-verify
