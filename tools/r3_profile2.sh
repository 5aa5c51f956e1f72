#!/bin/bash
# end-of-round evidence of the final code: kernel stats + FETCH/WRITE passes of the default bench (two encodes in flight), a pipelined kernel
# timeline, kernel stats of the enwik8-size run with four jobs in flight (+ its timeline)
export TMPDIR=/tmp
bash tools/profile.sh r3f_pipe > gpurun_out/r3f_profile_pipe.log 2>&1; echo "pipe profile rc=$?"
bash tools/r3_kt.sh r3f_timeline > gpurun_out/r3f_timeline.log 2>&1; echo "timeline rc=$?"
bash tools/r3_kt.sh r3f_e8_timeline --size 100000000 > gpurun_out/r3f_e8_timeline.log 2>&1; echo "e8 timeline rc=$?"
ls gpurun_out/prof_r3f_pipe gpurun_out/kt_r3f_timeline gpurun_out/kt_r3f_e8_timeline
