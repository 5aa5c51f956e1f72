#!/bin/bash
# free-running jobs at growing input sizes (is a large input better coded as several smaller calls?)
DST=$PWD/gpurun_out/r3_sz; mkdir -p $DST
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --quick "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['config']['encodes_in_flight'], d['kernel_ms_per_step'])"; }
run s250_p4 --size 250000000
run s250_p3 --size 250000000 --pipeline 3
run s500_p4 --size 500000000
run s500_p2o --size 500000000 --pipeline 2 --tune 4096
run s500_p3 --size 500000000 --pipeline 3
