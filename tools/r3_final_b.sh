#!/bin/bash
DST=$PWD/gpurun_out/r3_fin; mkdir -p $DST
( time timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $DST/bench_driver_style.json 2> $DST/bench_driver_style.err; tail -4 $DST/bench_driver_style.err
python3 -c "
import json
d=json.loads([l for l in open('$DST/bench_driver_style.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['frac'])
print('sync', d['one_call_at_a_time']['value']); print('decode', d['decode']['value']); print('cpu', d['cpu_baseline']['bit_exact_vs_gpu'], d['cpu_baseline']['buffers_checked'], d['cpu_baseline']['value'])
print('other', [ (o.get('context_model','')[:12], o.get('bytes'), o.get('value'), o.get('ms_per_step'), o.get('pipeline'), o.get('error')) for o in d['other_configs']])
print('small', [ (o.get('what','')[:30], o.get('value'), o.get('ms_per_step'), o.get('pipeline'), o.get('projected_8gpu_strong_MiBps'), o.get('error')) for o in d['small_inputs']])
print('ref', d['reference_stream_model']['value'])"
