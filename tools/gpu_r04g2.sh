#!/bin/bash
# second matrix of TOAST_HIP_FFT_STREAM_HINT (tools/gpu_r04g.sh): single bits and combinations around bit 0
tag=${1:-r04g2}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for rep in 1 2; do
  for h in 0 1 4 8 16 32 9 33 41 5; do
    TOAST_HIP_FFT_STREAM_HINT=$h python tools/exp_fft_prefetch.py 2>/dev/null | grep "ms per call" >> $out/timing.txt
  done
done
cat $out/timing.txt
cd /tmp; export TMPDIR=/tmp
for h in 1 41; do
  export TOAST_HIP_FFT_STREAM_HINT=$h
  for c in TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B; do
    rocprofv3 --pmc $c -d /tmp/g${h}_$c -o fft -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_prefetch.py > /dev/null 2>&1
  done
  rocprofv3 --kernel-trace --stats -d /tmp/gt$h -o fft -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_prefetch.py > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_bytes.py /tmp/g${h}_ "stream_hint $h" | grep "==\|k_fft\|kernel" >> $out/bytes.txt
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls /tmp/gt$h/*/fft_results.db /tmp/gt$h/fft_results.db 2>/dev/null | head -1) | grep "k_fft" >> $out/bytes.txt
done
cat $out/bytes.txt
