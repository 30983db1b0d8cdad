#!/bin/bash
# Placement policy of the memory manager at the bench level: alternating fresh processes with the policy
# (default, TOAST_HIP_ALLOC unset = probe:8) and without (TOAST_HIP_ALLOC=plain) on one box; prints value, step time,
# the two kernels, set-up time and what the policy did (profiles/r03_b).  Run on the GPU box.
out=${1:-gpurun_out/r03b/alloc_policy.txt}
n=${2:-3}
mkdir -p $(dirname $out); : > $out
for i in $(seq 1 $n); do
  for mode in policy plain; do
    if [ $mode = plain ]; then export TOAST_HIP_ALLOC=plain; else unset TOAST_HIP_ALLOC; fi
    python bench.py --no-cpu-baseline --no-fft --no-operator-level --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['allocator_stats']
print('%-6s run $i  value %.2f G/s  step %.3f ms  bnw %.3f  scan %.3f  rw-stream %.2f TB/s  setup %.2f s  probed %d blocks (%d fast) %d candidates %.1f ms' % ('$mode', d['value']/1e9, d['ms_per_step'], d['kernel_ms']['bnw'], d['kernel_ms']['scan'], d['roofline']['stream_ceiling']['read_write_GBs']/1e3, d['setup_s'], a['probed_blocks'], a['fast_blocks'], a['candidates'], a['probe_ms']))
" >> $out
  done
done
cat $out
