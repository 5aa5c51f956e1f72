#!/bin/bash
DST=$PWD/gpurun_out/r3_t; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_cm.py -x -q -m gpu > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -5 "$DST/pytest.txt"
for t in 0 256 1024; do
timeout -k 10 300 python3 bench.py --model fullcm --steps 2 --warmup 1 --quick --pipeline 1 --no-verify --tune $t > /tmp/b.json 2> /tmp/b.err || tail -3 /tmp/b.err
python3 -c "
import json
d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][0])
print('tune $t', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['slot_ms'])"
done
