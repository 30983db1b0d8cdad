#!/bin/bash
# round 6, VERDICT item 5: the packed solver sweeps with the wave-uniform amplitude look-up against the per-lane one
# (TOAST_HIP_PACKED_UNIFORM_AMPS=0): parity tests, then per-kernel times from a rocprofv3 kernel trace of bench.py.  $1 = tag
tag=${1:-r06c}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests/test_gpu_packed.py tests/test_gpu_mapmaker_e2e.py -q -s 2>&1 | grep -v "^\[toast_hip\]" | tail -25 > $out/tests.log
cd /tmp && export TMPDIR=/tmp
for u in 1 0 1 0; do
  export TOAST_HIP_PACKED_UNIFORM_AMPS=$u
  timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/trace_$u -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fft > $out/bench_$u.json 2> $out/bench_$u.err
  db=$(ls $out/trace_$u/bench_results.db 2>/dev/null || ls $out/trace_$u/*/bench_results.db | head -1)
  echo "== TOAST_HIP_PACKED_UNIFORM_AMPS=$u" >> $out/kernels.txt
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | grep "k_offset_\|k_scan_map_v2\|k_build_noise" >> $out/kernels.txt
  python3 -c "
import json,sys
d=json.loads(open('$out/bench_$u.json').read().strip().splitlines()[-1])
p=d['pcg_lhs_offset_templates']
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in p.items() if k.startswith('packed') and not isinstance(v,dict)})
" >> $out/kernels.txt
  find $out/trace_$u -name '*.db' -delete
done
cd $GRAFT_REPO_ROOT
cat $out/tests.log | tail -8; cat $out/kernels.txt
