#!/bin/bash
DST=$PWD/gpurun_out/r3_t; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -5 "$DST/pytest.txt"
( time timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $DST/bench_driver_style.json 2> $DST/bench_driver_style.err; tail -4 $DST/bench_driver_style.err
python3 -c "
import json
d=json.loads([l for l in open('$DST/bench_driver_style.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:20], d['roofline']['frac'], d['roofline']['traffic'])
print(d['one_call_at_a_time']); print(d['floors']); print(d['decode']); print(d['cpu_baseline']['bit_exact_vs_gpu'], d['cpu_baseline']['buffers_checked'])
print([ (o.get('value'), o.get('ms_per_step')) for o in d['other_configs']])"
