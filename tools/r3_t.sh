#!/bin/bash
DST=$PWD/gpurun_out/r3_t; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "twophase_path or saturation or order2 or ballot or large_blocks or edge or mixed or submit" > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -5 "$DST/pytest.txt"
for mode in "--pipeline 1" "--pipeline 2"; do
timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --quick $mode > $DST/b.json 2> $DST/b.err || tail -3 $DST/b.err
python3 -c "
import json
d=json.loads([l for l in open('$DST/b.json') if l.startswith('{')][0])
print('$mode', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
for r in d['roofline_kernels']: print('   ', r['kernel'][:40], r['avg_launch_ms'], r['frac'])"
done
