#!/bin/bash
for t in "--pipeline 2 --no-verify" "--pipeline 2 --no-verify --tune 8" "--pipeline 2 --no-verify --tune 16" "--pipeline 2 --no-verify --tune 24"; do
  tag=$(echo $t | tr -d ' -_')
  echo "=== $t"
  bash tools/r3_kt.sh $tag $t 2>&1 | tail -16
done
