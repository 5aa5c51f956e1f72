#!/bin/bash
DST=$PWD/gpurun_out/r3_nb; mkdir -p $DST
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 12 --warmup 3 --quick "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
k={r['kernel'].split(' ')[0]: r['avg_launch_ms'] for r in d['roofline_kernels']}
print('$tag', d['value'], d['ms_per_step'], k)"; }
run nb_default
run nb_default_b
