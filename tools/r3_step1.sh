#!/bin/bash
# round 3, first GPU pass: changed tests, then bench sync vs pipelined
DST=$PWD/gpurun_out/r3_step1
mkdir -p "$DST"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cm.py -x -q -m gpu > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -5 "$DST/pytest.txt"
for mode in "--pipeline 1" "--pipeline 1 --variant half_cu" "--pipeline 2 --variant full_cu" "--pipeline 2"; do
  tag=$(echo $mode | tr -d ' -' )
  timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --quick $mode > "$DST/bench_$tag.json" 2> "$DST/bench_$tag.err" || { echo "bench $mode failed"; tail -3 "$DST/bench_$tag.err"; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('$DST/bench_$tag.json') if l.startswith('{')][0])
print('$mode', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
for r in d['roofline_kernels']: print('   ', r['kernel'][:40], r['avg_launch_ms'], r['frac'])
" 2>&1 | tail -12
done
