#!/bin/bash
for t in "--pipeline 2 --no-verify" "--pipeline 2 --no-verify --tune 36" "--pipeline 2 --no-verify --tune 37" "--pipeline 2 --no-verify --tune 38"; do
  tag=$(echo $t | tr -d ' -_')
  echo "=== $t"
  bash tools/r3_kt.sh $tag $t 2>&1 | tail -22
done
