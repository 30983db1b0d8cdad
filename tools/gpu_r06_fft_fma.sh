#!/bin/bash
# round 6: the fused FFT passes with floating-point contraction allowed (v_fma_f64 in the butterflies and complex products)
# against the library's -ffp-contract=off build.  EXPERIMENT: the second library is built on the box with the flag for EVERY
# file (the pixel path then loses its bit-exactness: timing of the FFT only).  $1 = tag
tag=${1:-r06h}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
run() {
  python tools/exp_fft_long.py 1024 720000 2>/dev/null | tail -1 | cut -c1-160 >> $out/fft.txt
  python tools/exp_fft_long.py 256 2880000 2>/dev/null | tail -1 | cut -c1-160 >> $out/fft.txt
}
echo "== -ffp-contract=off (the shipped library)" >> $out/fft.txt
run; run
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $out/t0 -o f -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_long.py 1024 720000 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls $out/t0/f_results.db $out/t0/*/f_results.db 2>/dev/null | head -1) | grep "k_fft" >> $out/fft.txt
cd $GRAFT_REPO_ROOT
TOAST_HIP_EXTRA_FLAGS="-ffp-contract=fast" python -m toast_amd.build --force > $out/build.log 2>&1
echo "== -ffp-contract=fast (every file: experiment)" >> $out/fft.txt
run; run
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $out/t1 -o f -- python3 $GRAFT_REPO_ROOT/tools/exp_fft_long.py 1024 720000 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(ls $out/t1/f_results.db $out/t1/*/f_results.db 2>/dev/null | head -1) | grep "k_fft" >> $out/fft.txt
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fft.py -q 2>&1 | tail -4 >> $out/fft.txt
find $out -name '*.db' -delete
cat $out/fft.txt
