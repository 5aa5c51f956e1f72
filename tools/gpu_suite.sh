#!/bin/bash
# the whole GPU test suite in one process (what the driver runs at round end), output kept under gpurun_out/
DST=$PWD/gpurun_out/suite_${1:-x}; mkdir -p $DST
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -6 "$DST/pytest.txt"
