#!/bin/bash
# Final profile of a round (run on the GPU box from the repo root): kernel trace + FETCH_SIZE / WRITE_SIZE passes of
# bench.py, the un-profiled bench line, and the condensed text / traffic JSON.  $1 = tag (e.g. r02i)
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
# the traced process with the library's own log of the slab search (TOAST_HIP_TRACE=1 prints; it does not synchronise)
export TOAST_HIP_TRACE=1
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/trace.json 2> $out/trace.err
unset TOAST_HIP_TRACE
# A trace of a process whose zone placement did not work out is not a profile of the line (round 5: 1 chunk of 8 in the
# other zone under the profiler, scan_map 12 % slower than in the line): refuse to turn it into profiles/ text.
python3 - $out/trace.json <<'PY' || { echo "gpu_final_profile: the traced process' placement failed (see $out/trace.json, trace.err): no profile text" >&2; touch $out/PLACEMENT_FAILED; }
import json, sys
a = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["allocator_stats"]
print("traced process: chunks_other_zone %d of %d wanted, placement_ok %s, search_exhausted %s, %.1f ms per hipMemCreate" % (
    a["chunks_other_zone"], a["chunks_other_wanted"], a["placement_ok"], a["search_exhausted"], a["create_ms_per_chunk"]), file=sys.stderr)
sys.exit(0 if 2 * a["chunks_other_zone"] >= a["chunks_other_wanted"] and a["chunks_other_zone"] > 0 else 1)
PY
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
