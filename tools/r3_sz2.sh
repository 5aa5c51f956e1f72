#!/bin/bash
DST=$PWD/gpurun_out/r3_sz; mkdir -p $DST
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 9 --warmup 3 --quick "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['config']['encodes_in_flight'], d['kernel_ms_per_step'])"; }
run s750_ord --size 750000000
run s750_f3 --size 750000000 --pipeline 3 --tune 8192
run s1000_f3 --pipeline 3 --tune 8192
run s1000_f2 --pipeline 2 --tune 8192
