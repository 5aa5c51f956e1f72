#!/bin/bash
# (the no-store knock-outs, tune 256 / 768 / 1280, need a -DW3_TUNING build: W3_EXTRA_FLAGS=-DW3_TUNING python3 -m weath3rb0i_amd.build)
for t in 0 256 512 768 1024 1280; do
  bash tools/r3_s2p.sh e8t$t --model fullcm --size 100000000 --pipeline 1 --tune $t | grep -E "k_slot_replay|k_slot_sort|k_slot_events" | head -4 | sed "s/^/tune $t: /"
done
bash tools/r3_s2p.sh cfg4t512 --model fullcm --data mixed --block-size 262144 --size 211938580 --pipeline 1 --tune 512 | grep -E "k_slot_replay" | sed "s/^/cfg4 tune 512: /"
bash tools/r3_s2p.sh cfg4t1024 --model fullcm --data mixed --block-size 262144 --size 211938580 --pipeline 1 --tune 1024 | grep -E "k_slot_replay" | sed "s/^/cfg4 tune 1024: /"
