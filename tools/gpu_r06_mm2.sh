#!/bin/bash
# round 6: one-pass pack, wide flag count: tests, workflow wall times and gap listing, the default bench line.  $1 = tag
tag=${1:-r06j}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests/test_gpu_packed.py tests/test_gpu_mapmaker_e2e.py tests/test_gpu_ops.py tests/test_gpu_configs.py tests/test_gpu_bench.py -q -x 2>&1 | tail -8 > $out/tests.log
cat $out/tests.log
tools/gpu_wf_gaps.sh gpurun_out/$tag/wf > $out/wf.log 2>&1
head -16 $out/wf.log; grep -v "k_offset_.*_pr<\|short kernels" $out/wf/gaps_mapmaker.txt | head -40
python bench.py > $out/bench.json 2> $out/bench.err
python tools/bench_line.py -v $out/bench.json | head -8
