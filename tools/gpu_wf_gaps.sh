#!/bin/bash
# Where the GPU idles during the operator-level workflow (NoiseFilter + MapMaker at cfg-3): kernel trace -> idle-gap
# listing of the MapMaker part, per-kernel summary, and the un-profiled phase times.  $1 = out dir (under gpurun_out/)
out=${1:-gpurun_out/wf_gaps}
mkdir -p $out
cd /root/repo
python workflows/mapmaker_pcg.py > $out/plain.log 2>&1
export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/prof -o wf -- python3 workflows/mapmaker_pcg.py > $out/prof.log 2>&1
db=$(ls $out/prof/wf_results.db 2>/dev/null || ls $out/prof/*/wf_results.db | head -1)
python tools/rocpd_summary.py $db > $out/kernels.txt
python tools/rocpd_gaps.py $db --min 0.25 > $out/gaps_all.txt
python tools/rocpd_gaps.py $db --min 0.25 --from k_build_cov > $out/gaps_mapmaker.txt 2>/dev/null
find $out -name '*.db' -delete
tail -14 $out/plain.log
head -60 $out/gaps_mapmaker.txt
