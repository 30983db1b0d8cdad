#!/bin/bash
# headline kernels of bench.py in N fresh processes (the placement of the allocations moves every kernel of a process by
# a few per cent: compare ratios and several runs): $1 = N (default 4), rest = extra bench.py arguments.
# TRACE=1: also the zone-interleaved slab's log lines of every process (rates of the candidate chunks, what was built).
# PAUSE=seconds: wait between the processes (the driver clears a finished process' memory in the background).
n=${1:-4}; shift
for i in $(seq $n); do
  [ -n "$PAUSE" ] && [ $i -gt 1 ] && sleep $PAUSE
  ${TRACE:+env TOAST_HIP_TRACE=1} python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/tmp/bench_repeat.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms']
print('value %.2f G/s  step %.3f ms  scan %.3f  bnw %.3f  scan/bnw %.3f  frac %.3f  setup %.2f s' % (d['value']/1e9, d['ms_per_step'], k['scan'], k['bnw'], k['scan']/k['bnw'], d['roofline']['frac'], d.get('setup_s', 0)))
"
  [ -n "$TRACE" ] && grep "vmm \|scatter class\|zone" /tmp/bench_repeat.err | cut -c1-700
done
