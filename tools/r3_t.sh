#!/bin/bash
DST=$PWD/gpurun_out/r3_t; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_sweep.py -x -q -m gpu > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -25 "$DST/pytest.txt"
