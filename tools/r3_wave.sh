#!/bin/bash
DST=$PWD/gpurun_out/r3_wave; mkdir -p $DST
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave" > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -3 "$DST/pytest.txt"
for m in ac26 ordern32_1 ordern22_2 ac20; do
  timeout -k 10 400 python3 bench.py --model $m --steps 3 --warmup 1 --quick --pipeline 1 > $DST/bench_$m.json 2> $DST/bench_$m.err || { echo "$m failed"; tail -3 $DST/bench_$m.err; }
  python3 -c "
import json
d=json.loads([l for l in open('$DST/bench_$m.json') if l.startswith('{')][0])
print('$m', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['config']['compressed_ratio'], d['config']['path'])"
done
