#!/bin/bash
# Host side of ops.MapMaker at cfg-3: cProfile of the call (own time per function, then cumulative time of this package's
# functions) and the TOAST_HIP_TRACE=2 call time line (every library call synchronised and stamped; tools/trace_timeline.py
# names the host-only stretches).  $1 = out dir, $2 = extra pytest selection to run first (optional)
out=${1:-gpurun_out/wf_host}
mkdir -p $out
cd /root/repo
if [ -n "$2" ]; then python -m pytest $2 -x -q 2>&1 | tail -8 | tee $out/tests.txt; fi
for i in 1 2 3; do python workflows/mapmaker_pcg.py > $out/plain$i.log 2>&1; grep "MapMaker (cov\|NoiseFilter  " $out/plain$i.log; done
python workflows/mapmaker_pcg.py --profile > $out/cprofile.log 2>&1
TOAST_HIP_TRACE=2 python workflows/mapmaker_pcg.py > $out/trace_stdout.log 2> $out/trace.log
python tools/trace_timeline.py $out/trace.log > $out/timeline.txt 2>&1
tail -14 $out/plain1.log
grep -A 25 "tottime" $out/cprofile.log | head -40
