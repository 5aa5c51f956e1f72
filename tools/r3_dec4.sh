#!/bin/bash
DST=$PWD/gpurun_out/r3_dec; mkdir -p $DST
timeout -k 10 1000 python -m pytest tests/test_gpu_cm.py tests/test_gpu_random.py -x -q -m gpu > "$DST/pytest4.txt" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 "$DST/pytest4.txt"
[ $rc -ne 0 ] && exit $rc
python3 tools/decode_rate.py fullcm 1e8 | tail -1
python3 tools/decode_rate.py fullcm 4e8 | tail -1
