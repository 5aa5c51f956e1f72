#!/bin/bash
# Evidence set of a round (run through gpurun): tools/profile.sh (kernel stats + FETCH_SIZE + WRITE_SIZE passes) for the bench
# default model and the others named; results gathered under gpurun_out/round_<tag>/ in the layout of profiles/r1_s*/ .
#   tools/profile_round.sh <tag> [models...]
TAG=$1; shift
MODELS=${*:-order012apm}
DST=$PWD/gpurun_out/round_$TAG
mkdir -p "$DST"
for m in $MODELS; do
  bash tools/profile.sh ${TAG}_$m --model $m > "$DST/${m}_profile.log" 2>&1 || { echo "profile.sh failed for $m"; tail -5 "$DST/${m}_profile.log"; exit 1; }
  P=$PWD/gpurun_out/prof_${TAG}_$m
  cp "$P/kernel_stats.csv" "$DST/${m}_kernel_stats.csv"
  grep "^{\"metric\"" "$P/bench_kt.log" | tail -1 > "$DST/${m}_bench_line_under_rocprof.json"
  cp "$P/pmc_summary.json" "$DST/${m}_pmc_summary.json"
  for c in fetch write; do
    f=$(find "$P/$c" -name "*counter_collection.csv" | head -1)
    up=$(echo $c | tr a-z A-Z)
    [ -n "$f" ] && (head -1 "$f"; grep "w3::" "$f") > "$DST/${m}_pmc_${up}_SIZE.csv"
  done
  echo "$m done"
done
