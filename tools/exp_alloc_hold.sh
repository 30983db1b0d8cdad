#!/bin/bash
# Placement policy: slow candidates held between searches (default, TOAST_HIP_ALLOC_HOLD_GB=24) against released at once
# (=0), alternating fresh bench.py processes on one box, each after an allocate / touch / free cycle of 150 GB by another
# process (what the test suite leaves behind when the driver runs bench.py).  Run on the GPU box (profiles/r03_b).
out=${1:-gpurun_out/r03k/alloc_hold.txt}
n=${2:-3}
mkdir -p $(dirname $out); : > $out
for i in $(seq 1 $n); do
  for hold in 24 0; do
    tools/ubench/placement_survey 1 150 > /dev/null 2>&1
    export TOAST_HIP_ALLOC_HOLD_GB=$hold
    python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['allocator_stats']
print('hold %-2s run $i  value %.2f G/s  step %.3f ms  bnw %.3f  scan %.3f  setup %.2f s  probed %d blocks (%d fast) %d candidates, probe %.1f ms, hipMalloc %.0f ms (max %.0f), budget stops %d, held reused %d' % ('$hold', d['value']/1e9, d['ms_per_step'], d['kernel_ms']['bnw'], d['kernel_ms']['scan'], d['setup_s'], a['probed_blocks'], a['fast_blocks'], a['candidates'], a['probe_ms'], a['malloc_ms'], a['max_malloc_ms'], a['budget_stops'], a['held_reused']))
" >> $out
  done
done
cat $out
