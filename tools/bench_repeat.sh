#!/bin/bash
# headline kernels of bench.py in N fresh processes (the placement of the allocations moves every kernel of a process by
# a few per cent: compare ratios and several runs): $1 = N (default 4), rest = extra bench.py arguments
n=${1:-4}; shift
for i in $(seq $n); do
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms']
print('value %.2f G/s  step %.3f ms  scan %.3f  bnw %.3f  scan/bnw %.3f  frac %.3f' % (d['value']/1e9, d['ms_per_step'], k['scan'], k['bnw'], k['scan']/k['bnw'], d['roofline']['frac']))
"
done
