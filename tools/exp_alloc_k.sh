#!/bin/bash
# Placement policy on a box in the state the test suite leaves behind (the full-size tests run first): K = 8 candidates
# per block (default) against K = 16, and the acceptance level 5.65 against 5.9 TB/s; alternating fresh bench.py
# processes.  Run on the GPU box (profiles/r03_b).
out=${1:-gpurun_out/r03k/alloc_k.txt}
n=${2:-4}
mkdir -p $(dirname $out); : > $out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -m gpu > /dev/null 2>&1
for i in $(seq 1 $n); do
  for mode in "probe:8 5.65" "probe:16 5.65" "probe:16 5.9"; do
    set -- $mode
    export TOAST_HIP_ALLOC=$1 TOAST_HIP_ALLOC_ACCEPT_TBS=$2
    python bench.py --no-cpu-baseline --no-fft --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['allocator_stats']
print('%-9s accept %-4s run $i  value %.2f G/s  step %.3f ms  bnw %.3f  scan %.3f  setup %.2f s  probed %d (%d fast) %d candidates, probe %.1f ms, hipMalloc %.0f ms (max %.0f), stops %d, held %.1f GB reused %d' % ('$1', '$2', d['value']/1e9, d['ms_per_step'], d['kernel_ms']['bnw'], d['kernel_ms']['scan'], d['setup_s'], a['probed_blocks'], a['fast_blocks'], a['candidates'], a['probe_ms'], a['malloc_ms'], a['max_malloc_ms'], a['budget_stops'], a['held_GB'], a['held_reused']))
" >> $out
  done
done
cat $out
