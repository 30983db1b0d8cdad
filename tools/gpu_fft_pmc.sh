#!/bin/bash
# Counters of the FFT passes (one rocprofv3 --pmc run per counter group; no tracing in the same run): SQ groups and the
# exact DRAM byte counters (32-byte units).   Usage (GPU box): tools/gpu_fft_pmc.sh <out_dir> "<n_det> <n_samp>" [rows:cols ...]
out=${1:-gpurun_out/fft_pmc}
shape=${2:-512 720000}
shift; shift
modes=${@:-reg:reg lds:lds}
mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
 "SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"
 "TCC_EA0_RDREQ_DRAM_32B"
 "TCC_EA0_WRREQ_WRITE_DRAM_32B"
 "TCC_HIT_sum TCC_MISS_sum"
)
for m in $modes; do
  rows=${m%%:*}; cols=${m##*:}
  export TOAST_HIP_FFT_ROWS=$rows TOAST_HIP_FFT_COLS=$cols
  i=0
  for g in "${groups[@]}"; do
    [ -n "$FFT_PMC_GROUPS" ] && ! echo " $FFT_PMC_GROUPS " | grep -q " $i " && { i=$((i+1)); continue; }
    timeout -k 5 900 rocprofv3 --pmc $g -d $out/${rows}_$cols/g$i -o p -- python3 tools/exp_fft_long.py $shape > $out/${rows}_${cols}_g$i.log 2>&1
    i=$((i+1))
  done
  echo "#### rows=$rows cols=$cols shape=$shape" >> $out/pmc.txt
  python3 tools/pmc_table.py $out/${rows}_$cols k_fft >> $out/pmc.txt
done
cat $out/pmc.txt
