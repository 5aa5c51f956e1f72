#!/bin/bash
DST=$PWD/gpurun_out/r3_step3; mkdir -p $DST
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cm.py -x -q -m gpu -k "submit or half or verification or huff_keys or coder" > "$DST/pytest.txt" 2>&1
echo "pytest rc=$?"; tail -4 "$DST/pytest.txt"
for t in "--pipeline 2" "--pipeline 2 --tune 4"; do
  tag=$(echo $t | tr -d ' -_')
  echo "=== $t"
  bash tools/r3_kt.sh $tag $t 2>&1 | tail -34
done
