#!/bin/bash
# all GPU tests (stop at the first failure), smoke, the workflow's gap listing, the default bench line.  $1 = tag
tag=${1:-r06k}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $out/gputests.log
cat $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/gpu_wf_gaps.sh gpurun_out/$tag/wf > $out/wf.log 2>&1
head -16 $out/wf.log; grep -v "k_offset_.*_pr<\|short kernels" $out/wf/gaps_mapmaker.txt | head -30
python bench.py > $out/bench.json 2> $out/bench.err
python tools/bench_line.py -v $out/bench.json | head -8
