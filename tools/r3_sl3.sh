#!/bin/bash
for t in 0 512; do
  bash tools/r3_s2p.sh e9t$t --model fullcm --pipeline 1 --variant slot_sorted --tune $t | grep -E "k_slot|k_apm|k_coder" | head -6 | sed "s/^/1e9 tune $t: /"
  grep '^{' gpurun_out/s2p_e9t$t/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('1e9 tune $t', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
