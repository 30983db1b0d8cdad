#!/bin/bash
# Final profile of a round (run on the GPU box from the repo root): kernel trace + FETCH_SIZE / WRITE_SIZE passes of
# bench.py, the un-profiled bench line, and the condensed text / traffic JSON.  $1 = tag (e.g. r02i)
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/trace.json 2> $out/trace.err
timeout -k 5 900 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --unfused > $out/fetch.json 2> /dev/null
timeout -k 5 900 rocprofv3 --pmc WRITE_SIZE -d $out/write -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --unfused > $out/write.json 2> /dev/null
# the same kernels through the 32-byte-unit DRAM request counters (exact on known byte counts, profiles/r04_e)
for c in TCC_EA0_RDREQ_DRAM_32B TCC_EA0_WRREQ_WRITE_DRAM_32B TCC_EA0_WRREQ_WRITE_ATOMIC_32B; do
  timeout -k 5 900 rocprofv3 --pmc $c -d /tmp/fp_$c -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_bytes.py --json $out/traffic_exact_cfg3.json cfg3 /tmp/fp_ "bench.py cfg3 (1024 x 720 000 samples)" > $out/exact_bytes.txt
cd $GRAFT_REPO_ROOT
f() { ls $out/$1/bench_results.db 2>/dev/null || ls $out/$1/*/bench_results.db | head -1; }
python tools/rocpd_summary.py $(f trace) > $out/trace.txt
python tools/rocpd_summary.py $(f fetch) | grep "FETCH_SIZE" > $out/fetch.txt
python tools/rocpd_summary.py $(f write) | grep "WRITE_SIZE" > $out/write.txt
python tools/rocpd_summary.py --traffic $(f fetch) $(f write) $out/traffic_cfg3.json cfg3 "$2"
find $out -name '*.db' -delete
tail -c 600 $out/bench.json
