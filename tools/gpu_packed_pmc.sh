#!/bin/bash
# Counters of the packed left-hand-side sweeps (k_offset_accumulate_pr / k_offset_scan_project_pr) as bench.py runs them at
# cfg-3: one rocprofv3 --pmc run per counter group (no tracing in the same run), then a kernel trace for the durations.
# Usage (GPU box): tools/gpu_packed_pmc.sh <out_dir>
out=${1:-gpurun_out/packed_pmc}
mkdir -p $out
cd /root/repo
export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"
 "TCC_EA0_RDREQ_DRAM_32B"
 "TCC_EA0_WRREQ_WRITE_DRAM_32B"
 "TCC_EA0_WRREQ_ATOMIC_DRAM_32B"
 "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
)
i=0
for g in "${groups[@]}"; do
  timeout -k 5 900 rocprofv3 --pmc $g -d $out/g$i -o p -- python3 bench.py --no-cpu-baseline --no-operator-level --no-fft --steps 3 --warmup 1 > $out/g$i.log 2>&1
  i=$((i+1))
done
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/trace -o p -- python3 bench.py --no-cpu-baseline --no-operator-level --no-fft --steps 3 --warmup 1 > $out/trace.log 2>&1
python3 tools/pmc_table.py $out "k_offset" > $out/pmc.txt
python3 tools/rocpd_summary.py $(ls $out/trace/p_results.db $out/trace/*/p_results.db 2>/dev/null | head -1) | grep "k_offset\|kernel " >> $out/pmc.txt
find $out -name '*.db' -delete      # (gpurun brings back at most 64 MiB)
cat $out/pmc.txt
