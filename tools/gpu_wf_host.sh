#!/bin/bash
# Host side of ops.MapMaker at cfg-3: cProfile of the call (own time per function) and the TOAST_HIP_TRACE=2 call time line
# (every library call synchronised and stamped; tools/trace_timeline.py names the host-only stretches).  $1 = out dir
out=${1:-gpurun_out/wf_host}
mkdir -p $out
cd /root/repo
python workflows/mapmaker_pcg.py > $out/plain.log 2>&1
python workflows/mapmaker_pcg.py --profile > $out/cprofile.log 2>&1
TOAST_HIP_TRACE=2 python workflows/mapmaker_pcg.py > $out/trace_stdout.log 2> $out/trace.log
python tools/trace_timeline.py $out/trace.log > $out/timeline.txt 2>&1
tail -14 $out/plain.log
grep -A 40 "tottime" $out/cprofile.log | head -60
