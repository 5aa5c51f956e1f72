#!/bin/bash
DST=$PWD/gpurun_out/r3_t; mkdir -p $DST
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave or rejects or huff or twophase_path" > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -30 "$DST/pytest.txt"
