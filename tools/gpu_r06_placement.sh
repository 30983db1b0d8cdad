#!/bin/bash
# round 6, VERDICT item 1: the arena tests, six back-to-back default processes (no pause) with the slab search's log,
# then the final profile.  $1 = tag
tag=${1:-r06a}
mkdir -p gpurun_out/$tag
python -m pytest tests/test_gpu_accel.py tests/test_gpu_mapmaker_e2e.py -q -s 2>&1 | grep -v "^\[toast_hip\]" | tail -40 > gpurun_out/$tag/accel_tests.log
TRACE=1 tools/bench_repeat.sh 6 --no-fft --no-operator-level > gpurun_out/$tag/repeat6.log 2>&1
tools/bench_repeat.sh 3 > gpurun_out/$tag/repeat3_default.log 2>&1
tools/gpu_final_profile.sh $tag > gpurun_out/$tag/final.log 2>&1
cat gpurun_out/$tag/accel_tests.log gpurun_out/$tag/repeat6.log gpurun_out/$tag/repeat3_default.log | cut -c1-400
tail -5 gpurun_out/$tag/final.log
