#!/bin/bash
# Placement policy: acceptance level 5.65 TB/s (default: the middle level 5.7-5.77 counts as fast) against 5.9 (only the
# top level ends a search early), alternating fresh bench.py processes on one box.  Run on the GPU box (profiles/r03_b).
out=${1:-gpurun_out/r03k/alloc_accept.txt}
n=${2:-5}
mkdir -p $(dirname $out); : > $out
for i in $(seq 1 $n); do
  for acc in 5.65 5.9; do
    export TOAST_HIP_ALLOC_ACCEPT_TBS=$acc
    python bench.py --no-cpu-baseline --no-fft --no-operator-level --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['allocator_stats']
print('accept %-4s run $i  value %.2f G/s  step %.3f ms  bnw %.3f  scan %.3f  setup %.2f s  probed %d blocks (%d fast) %d candidates, probe %.1f ms' % ('$acc', d['value']/1e9, d['ms_per_step'], d['kernel_ms']['bnw'], d['kernel_ms']['scan'], d['setup_s'], a['probed_blocks'], a['fast_blocks'], a['candidates'], a['probe_ms']))
" >> $out
  done
done
cat $out
