#!/bin/bash
# PMC passes over tools/prof_hot_kernels.py (run on the GPU box from the repo root): $1 = tag
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/trace -o k -- python3 $GRAFT_REPO_ROOT/tools/prof_hot_kernels.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p1 -o k -- python3 $GRAFT_REPO_ROOT/tools/prof_hot_kernels.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/p2 -o k -- python3 $GRAFT_REPO_ROOT/tools/prof_hot_kernels.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for d in trace p1 p2; do python tools/rocpd_summary.py gpurun_out/pmc_$tag/$d/k_results.db > gpurun_out/pmc_$tag/$d.txt 2>&1; done
ls gpurun_out/pmc_$tag
