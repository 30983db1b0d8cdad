#!/bin/bash
# A/B of the fused FFT call: tile in LDS / in registers for the row pass (TOAST_HIP_FFT_ROWS) and the column passes
# (TOAST_HIP_FFT_COLS) at cfg-3, the 2^22 shape and the configs[3] shard shape, then per-kernel times from rocprofv3.
# Usage (GPU box): tools/gpu_fft_ab.sh [out_dir]      FFT_AB_MODES="rows:cols ..."  FFT_AB_NOPROF=1
out=${1:-gpurun_out/fft_ab}
mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
modes=${FFT_AB_MODES:-lds:lds reg:lds lds:reg reg:reg}
for m in $modes; do
  rows=${m%%:*}; cols=${m##*:}
  for shape in "1024 720000" "1024 1440000" "512 2880000"; do
    echo "== rows=$rows cols=$cols shape=$shape" | tee -a $out/ab.txt
    TOAST_HIP_FFT_ROWS=$rows TOAST_HIP_FFT_COLS=$cols python3 tools/exp_fft_long.py $shape 2>&1 | tail -1 | tee -a $out/ab.txt
  done
done
if [ -z "$FFT_AB_NOPROF" ]; then
for m in ${FFT_AB_PROF:-lds:lds reg:reg}; do
  rows=${m%%:*}; cols=${m##*:}
  export TOAST_HIP_FFT_ROWS=$rows TOAST_HIP_FFT_COLS=$cols
  for shape in "1024 720000" "512 2880000"; do
    tag=${rows}_${cols}_${shape##* }
    timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o p -- python3 tools/exp_fft_long.py $shape > $out/prof_$tag.log 2>&1
    echo "== rocprofv3 rows=$rows cols=$cols shape=$shape" | tee -a $out/ab.txt
    python3 tools/rocpd_summary.py $(ls $out/prof_$tag/p_results.db $out/prof_$tag/*/p_results.db 2>/dev/null | head -1) | grep "k_fft\|kernel " | tee -a $out/ab.txt
  done
done
fi
