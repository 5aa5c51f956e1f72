#!/bin/bash
# free-running jobs at enwik9 size for the models whose coder finds no rank kernels beside it
DST=$PWD/gpurun_out/r3_fr; mkdir -p $DST
run() { tag=$1; shift
  timeout -k 10 300 python3 bench.py --steps 9 --warmup 3 --quick "$@" > $DST/$tag.json 2> $DST/$tag.err || tail -3 $DST/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$DST/$tag.json') if l.startswith('{')][0])
print('$tag', d['value'], d['ms_per_step'], d['config']['encodes_in_flight'], d['kernel_ms_per_step'])"; }
for m in default order0 order012; do
  run ${m}_ord --model $m
  run ${m}_f2 --model $m --pipeline 2 --tune 8192
  run ${m}_f3 --model $m --pipeline 3 --tune 8192
done
