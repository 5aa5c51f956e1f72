#!/bin/bash
DST=$PWD/gpurun_out/r3_step6; mkdir -p $DST
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --quick > $DST/bench_quick.json 2> $DST/bench_quick.err || tail -3 $DST/bench_quick.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --quick --no-verify > $DST/bench_quick_noverify.json 2> $DST/bench_quick_nv.err || tail -3 $DST/bench_quick_nv.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --quick --model order012 > $DST/bench_quick_o012.json 2> $DST/bench_quick_o012.err || tail -3 $DST/bench_quick_o012.err
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --quick --model order012 --pipeline 1 > $DST/bench_quick_o012_p1.json 2> $DST/bench_quick_o012_p1.err || tail -3 $DST/bench_quick_o012_p1.err
for f in bench_quick bench_quick_noverify bench_quick_o012 bench_quick_o012_p1; do python3 -c "
import json
d=json.loads([l for l in open('$DST/$f.json') if l.startswith('{')][0])
print('$f', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['kernel'][:30], d['roofline']['frac'])"; done
( time timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 ) > $DST/bench_full.json 2> $DST/bench_full.err; tail -4 $DST/bench_full.err
python3 -c "
import json
d=json.loads([l for l in open('$DST/bench_full.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step']); print(d.get('decode')); print(d.get('other_configs')); print(d.get('cpu_baseline',{}).get('bit_exact_vs_gpu'), d.get('reference_stream_model',{}).get('value'))"
