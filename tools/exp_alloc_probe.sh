#!/bin/bash
# EXPERIMENT: operator-level PCG iteration with and without the manager's placement probe (TOAST_HIP_ALLOC=probe:K),
# alternating fresh processes on one box.
for r in 1 2 3; do
  for mode in none probe:4; do
    if [ $mode = none ]; then unset TOAST_HIP_ALLOC; else export TOAST_HIP_ALLOC=$mode; fi
    TOAST_HIP_TRACE=1 python workflows/mapmaker_pcg.py --no-filter > /tmp/probe_$$.log 2>&1
    echo "$mode: $(grep 'PCG iteration' /tmp/probe_$$.log | cut -c1-80)  $(grep 'MapMaker total' /tmp/probe_$$.log)  probes: $(grep -c '\] probe' /tmp/probe_$$.log)"
    grep '\] probe' /tmp/probe_$$.log | cut -c1-110 | head -4
  done
done
