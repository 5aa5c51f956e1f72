#!/bin/bash
# end-of-round evidence (run through gpurun): bench lines of every model (with the CPU legs), rocprofv3 stats + PMC passes of the
# default model and of order012.  Results under gpurun_out/final_<tag>/ ; copy into profiles/<round>/.
TAG=${1:-r2}
DST=$PWD/gpurun_out/final_$TAG
mkdir -p "$DST"
export TMPDIR=/tmp
for m in order012apm order012 default order0; do
  timeout -k 10 300 python3 bench.py --model $m --steps 10 --warmup 2 > "$DST/bench_$m.json" 2> "$DST/bench_$m.err" || echo "bench $m failed"
  echo "bench $m done"
done
timeout -k 10 300 python3 bench.py --model fullcm --steps 3 --warmup 1 --pipeline 1 --no-ref-model --no-other-configs > "$DST/bench_fullcm.json" 2> "$DST/bench_fullcm.err" || echo "bench fullcm failed"
timeout -k 10 300 python3 bench.py --data mixed --block-size 262144 --size 211938580 --steps 10 --warmup 2 > "$DST/bench_mixed_256k.json" 2> "$DST/bench_mixed.err" || echo "bench mixed failed"
timeout -k 10 300 python3 bench.py --scaling strong --force-exchange --steps 5 --quick > "$DST/bench_strong_1gpu_exchange_rehearsal.json" 2> "$DST/bench_strong.err" || echo "bench strong failed"
timeout -k 10 600 bash tools/profile_round.sh $TAG order012apm order012 2>&1 | tail -3
cp -r gpurun_out/round_$TAG/* "$DST/" 2>/dev/null
timeout -k 10 200 python3 tools/sweep_bench.py > "$DST/sweep_115_configs_20MB.txt" 2>&1
timeout -k 10 300 python3 tools/host_api_rate.py 2>&1 | grep -v amdgpu.ids > "$DST/host_api_rate.txt"
ls "$DST"
