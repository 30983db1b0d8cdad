#!/bin/bash
# round 6: (1) the fused FFT call with small detector batches -- does a work buffer that stays in the 256 MB Infinity Cache
# pay now that the passes run at half the HBM rate (round 2: it did not, the passes were latency bound then)?
# (2) SQ_INSTS_VALU of the packed sweeps with and without the wave-uniform amplitude look-up.   $1 = tag
tag=${1:-r06d}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for b in 4 8 12 16 24 32 64 128 512; do
  EXP_FFT_BATCH=$b python tools/exp_fft_long.py 1024 720000 2>/dev/null | tail -1 >> $out/fft_batch.txt
done
for b in 1 2 3 4 8 128; do
  EXP_FFT_BATCH=$b python tools/exp_fft_long.py 256 2880000 2>/dev/null | tail -1 >> $out/fft_batch.txt
done
cat $out/fft_batch.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for u in 1 0; do
  export TOAST_HIP_PACKED_UNIFORM_AMPS=$u
  timeout -k 5 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $out/pmc_$u -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-operator-level --no-fft --steps 3 --warmup 1 > $out/pmc_$u.log 2>&1
  echo "== TOAST_HIP_PACKED_UNIFORM_AMPS=$u" >> $out/pmc.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_table.py $out/pmc_$u "k_offset" | grep -A9 "_pr<true" >> $out/pmc.txt
  find $out/pmc_$u -name '*.db' -delete
done
cat $out/pmc.txt
