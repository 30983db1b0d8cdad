#!/bin/bash
# HBM bytes per launch from the L2's memory-side request counters that count in FIXED units:
#   TCC_EA0_RDREQ_DRAM_32B        "32-byte read requests due to DRAM traffic, a 64-byte request counts 2, a 128-byte one 4"
#   TCC_EA0_WRREQ_WRITE_DRAM_32B  the same for writes, TCC_EA0_WRREQ_WRITE_ATOMIC_32B for atomics
# (rocprofv3 --list-avail on gfx950) -- unlike FETCH_SIZE, whose gfx950 value depends on the request width (profiles/r03_d
# section 1: x2 for streams, x1.93 for the FFT column passes' 128-byte pieces).  First against known byte counts
# (tools/ubench/strided_copy, stream_ceiling), then the benchmark's kernels.  $1 = tag
tag=${1:-pmcb}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for c in TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_ATOMIC_32B TCC_BUBBLE; do
  timeout -k 5 900 rocprofv3 --pmc $c -d /tmp/pb_sc_$c -o sc -- $GRAFT_REPO_ROOT/tools/ubench/strided_copy > /dev/null 2>&1
  timeout -k 5 900 rocprofv3 --pmc $c -d /tmp/pb_st_$c -o st -- $GRAFT_REPO_ROOT/tools/ubench/stream_ceiling > /dev/null 2>&1
  timeout -k 5 900 rocprofv3 --pmc $c -d /tmp/pb_bench_$c -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/bench_$c.err
done
python3 $GRAFT_REPO_ROOT/tools/pmc_bytes.py /tmp/pb_sc_ "strided_copy: 4 294 967 296 B read and written per launch" \
    /tmp/pb_st_ "stream_ceiling: arrays of 5 898 240 000 B (x3 for the weights)" \
    /tmp/pb_bench_ "bench.py cfg3 (1024 x 720 000 samples)" > $out/pmc_bytes.txt
cat $out/pmc_bytes.txt
