#!/bin/bash
# Kernel trace of the operator-level workflow (NoiseFilter + MapMaker at cfg-3) plus its un-profiled wall times and the
# TOAST_HIP_TRACE=2 call time line (run on the GPU box from the repo root): $1 = tag
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_wf
mkdir -p $out
python workflows/mapmaker_pcg.py > $out/plain.log 2>&1
TOAST_HIP_TRACE=2 python workflows/mapmaker_pcg.py > $out/trace2.log 2>&1
python tools/trace_timeline.py $out/trace2.log > $out/timeline.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/prof -o wf -- python3 $GRAFT_REPO_ROOT/workflows/mapmaker_pcg.py > $out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(ls $out/prof/wf_results.db 2>/dev/null || ls $out/prof/*/wf_results.db | head -1)
python tools/rocpd_summary.py $db > $out/kernels.txt
tail -12 $out/plain.log
