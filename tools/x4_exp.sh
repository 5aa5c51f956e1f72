#!/bin/bash
# timing experiments for k_coder_x4 (run on the GPU box): which of the three waves sets the pace?
# usage: tools/x4_exp.sh "0 1 2 3"
set -e
mkdir -p gpurun_out/r2_x4exp
for e in ${1:-0 1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DW3_X4_EXP=$e -o weath3rb0i_amd/libw3hip.so weath3rb0i_amd/csrc/w3hip.hip
  echo "EXP=$e" >> gpurun_out/r2_x4exp/out.txt
  timeout -k 10 200 python bench.py --model order0 --size 100000000 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['kernel_ms_per_step'])
" >> gpurun_out/r2_x4exp/out.txt
done
cat gpurun_out/r2_x4exp/out.txt
