#!/bin/bash
# round 6: MapMaker set-up / final phases: tests of the changed routes, then the workflow's wall times and gap listing.  $1 = tag
tag=${1:-r06i}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests/test_gpu_mapmaker_e2e.py tests/test_gpu_rccl.py tests/test_gpu_packed.py -q -s 2>&1 | grep -v "^\[toast_hip\]" | tail -25 > $out/tests.log
python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_rccl_mock.py tests/test_gpu_deterministic.py -q -x 2>&1 | tail -6 >> $out/tests.log
cat $out/tests.log
tools/gpu_wf_gaps.sh gpurun_out/$tag/wf > $out/wf.log 2>&1
head -16 $out/wf.log; grep -v "k_offset_.*_pr<\|short kernels" $out/wf/gaps_mapmaker.txt | head -60
