#!/bin/bash
# Round 4: non-temporal hints in the fused FFT passes (TOAST_HIP_FFT_STREAM_HINT, bits: 1 pass-1 stores, 2 pass-1
# timestream loads, 4 pass-3 loads, 8 pass-3 stores, 16 row-pass loads, 32 row-pass stores): ms per call, alternating
# processes on one box, then the exact read counter per kernel for the interesting ones.   $1 = tag
tag=${1:-r04g}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for rep in 1 2; do
  for h in 0 1 2 3 12 15 48 63; do
    TOAST_HIP_FFT_STREAM_HINT=$h python tools/exp_fft_prefetch.py 2>/dev/null | grep "ms per call" >> $out/timing.txt
  done
done
cat $out/timing.txt
cd /tmp; export TMPDIR=/tmp
for h in 0 3 63; do
  export TOAST_HIP_FFT_STREAM_HINT=$h
  for c in TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B; do
    rocprofv3 --pmc $c -d /tmp/g${h}_$c -o fft -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_prefetch.py > /dev/null 2>&1
  done
  rocprofv3 --kernel-trace --stats -d /tmp/gt$h -o fft -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_prefetch.py > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_bytes.py /tmp/g${h}_ "stream_hint $h" | grep "==\|k_fft\|kernel" >> $out/bytes.txt
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls /tmp/gt$h/*/fft_results.db /tmp/gt$h/fft_results.db 2>/dev/null | head -1) | grep "k_fft" >> $out/bytes.txt
done
cat $out/bytes.txt
