#!/bin/bash
for t in "--pipeline 1 --variant half_cu --tune 2" "--pipeline 2 --tune 1" "--pipeline 2 --tune 3" "--pipeline 2 --tune 2"; do
  tag=$(echo $t | tr -d ' -_')
  echo "=== $t"
  bash tools/r3_kt.sh $tag $t 2>&1 | tail -42
done
