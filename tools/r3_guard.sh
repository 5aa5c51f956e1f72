#!/bin/bash
# the APM kernels' debug store guard (-DW3_TUNING): rebuild on the box, run the CM / CLI tests and one bench pass through it
DST=$PWD/gpurun_out/r3_guard; mkdir -p $DST
cp weath3rb0i_amd/libw3hip.so /tmp/libw3hip_release.so
W3_EXTRA_FLAGS=-DW3_TUNING python3 -c "from weath3rb0i_amd import build; build.build(force=True, verbose=True)" > $DST/build.txt 2>&1 || { tail -5 $DST/build.txt; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_cm.py tests/test_gpu_cli.py -x -q -m gpu > $DST/pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 $DST/pytest.txt
timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --quick > $DST/bench.json 2> $DST/bench.err; echo "bench rc=$?"; head -c 300 $DST/bench.json; echo
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --quick --size 270000 > $DST/bench_270000.json 2> $DST/bench_270000.err; echo "bench 270000 rc=$?"
cp /tmp/libw3hip_release.so weath3rb0i_amd/libw3hip.so
