#!/bin/bash
# round 6 mid-round check: all GPU tests, the default bench line, per-kernel times of the packed sweeps.  $1 = tag
tag=${1:-r06f}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $out/gputests.log
python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fft --no-operator-level > $out/trace.json 2> $out/trace.err
db=$(ls $out/trace/bench_results.db 2>/dev/null || ls $out/trace/*/bench_results.db | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | grep "k_offset_.*_pr\|k_scan_map_v2\|k_build_noise" > $out/kernels.txt
find $out/trace -name '*.db' -delete
cd $GRAFT_REPO_ROOT
cat $out/gputests.log | tail -6; cat $out/kernels.txt; python tools/bench_line.py -v $out/bench.json 2>/dev/null | head -20
