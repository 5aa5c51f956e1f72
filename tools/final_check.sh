#!/bin/bash
# Round-end validation on the GPU box: GPU test suite, smoke(), bench lines of every model (gpurun_out/final/).
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/final/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
for m in order012apm order012 default order0 fullcm; do
  timeout -k 10 300 python bench.py --model $m > gpurun_out/final/bench_$m.json.log 2>&1
  tail -1 gpurun_out/final/bench_$m.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['model'][:40], d['value'], d['ms_per_step'], d['roofline']['kernel'][:12], d['roofline']['frac'], d['kernel_ms_per_step'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('bit_exact_vs_gpu'))"
done
