#!/bin/bash
# Placement policy: lower bound of the probed size class 1 GB (default) against 256 MB (the 0.74 GB detector flags of
# cfg-3 are then probed as well), alternating fresh bench.py processes on one box.  Run on the GPU box (profiles/r03_b).
out=${1:-gpurun_out/r03k/alloc_min.txt}
n=${2:-3}
mkdir -p $(dirname $out); : > $out
for i in $(seq 1 $n); do
  for min in 1024 256; do
    export TOAST_HIP_ALLOC_PROBE_MIN_MB=$min
    python bench.py --no-cpu-baseline --no-fft --no-operator-level --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['allocator_stats']
print('min %-4s run $i  value %.2f G/s  step %.3f ms  bnw %.3f  scan %.3f  setup %.2f s  probed %d blocks (%d fast) %d candidates, probe %.1f ms' % ('$min', d['value']/1e9, d['ms_per_step'], d['kernel_ms']['bnw'], d['kernel_ms']['scan'], d['setup_s'], a['probed_blocks'], a['fast_blocks'], a['candidates'], a['probe_ms']))
" >> $out
  done
done
cat $out
