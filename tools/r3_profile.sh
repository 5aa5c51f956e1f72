#!/bin/bash
# round-3 evidence: kernel stats + FETCH_SIZE / WRITE_SIZE passes of the default bench (two encodes in flight) and of the synchronous form,
# a pipelined kernel timeline, and the bench lines of the other models
export TMPDIR=/tmp
bash tools/profile.sh r3_pipe > gpurun_out/r3_profile_pipe.log 2>&1; echo "pipe profile rc=$?"
bash tools/profile.sh r3_sync --pipeline 1 > gpurun_out/r3_profile_sync.log 2>&1; echo "sync profile rc=$?"
bash tools/r3_kt.sh r3_timeline > gpurun_out/r3_timeline.log 2>&1; echo "timeline rc=$?"
DST=$PWD/gpurun_out/r3_lines; mkdir -p $DST
for m in order012 default order0; do
  timeout -k 10 300 python3 bench.py --model $m --steps 10 --warmup 2 --no-other-configs > "$DST/bench_$m.json" 2> "$DST/bench_$m.err" || echo "bench $m failed"
done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --pipeline 1 --no-other-configs > "$DST/bench_order012apm_sync.json" 2> "$DST/bench_sync.err" || echo "bench sync failed"
timeout -k 10 300 python3 bench.py --scaling strong --force-exchange --steps 5 --quick > "$DST/bench_1gpu_exchange_rehearsal.json" 2> "$DST/bench_exch.err" || echo "bench exchange failed"
timeout -k 10 300 python3 bench.py --size 100000000 --steps 10 --warmup 2 --quick > "$DST/bench_order012apm_enwik8_size.json" 2> "$DST/bench_e8.err" || echo "bench enwik8 failed"
ls $DST
