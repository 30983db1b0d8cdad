#!/bin/bash
# round 6, VERDICT item 5: packed sweeps, wave-uniform amplitude look-up on / off: kernel times + SQ counters.  $1 = tag
tag=${1:-r06e}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests/test_gpu_packed.py -q 2>&1 | tail -3 > $out/tests.log
cd /tmp && export TMPDIR=/tmp
for u in 1 0 1 0; do
  export TOAST_HIP_PACKED_UNIFORM_AMPS=$u
  timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/trace_$u -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fft --no-operator-level > $out/bench_$u.json 2> $out/bench_$u.err
  db=$(ls $out/trace_$u/bench_results.db 2>/dev/null || ls $out/trace_$u/*/bench_results.db | head -1)
  echo "== TOAST_HIP_PACKED_UNIFORM_AMPS=$u" >> $out/kernels.txt
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $db | grep "k_offset_.*_pr" >> $out/kernels.txt
  find $out/trace_$u -name '*.db' -delete
done
for u in 1 0; do
  export TOAST_HIP_PACKED_UNIFORM_AMPS=$u
  timeout -k 5 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d $out/pmc_$u -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-level --no-fft --steps 3 --warmup 1 > $out/pmc_$u.log 2>&1
  echo "== TOAST_HIP_PACKED_UNIFORM_AMPS=$u" >> $out/pmc.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $out/pmc_$u "_pr<true" >> $out/pmc.txt
  find $out/pmc_$u -name '*.db' -delete
done
cat $out/tests.log $out/kernels.txt $out/pmc.txt
