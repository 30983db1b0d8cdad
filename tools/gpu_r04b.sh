#!/bin/bash
# round 4: rank-interleaved slabs -- the region scan on an interleaved slab, then the default bench line twice.  $1 = tag
tag=${1:-r04b}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
TOAST_HIP_TRACE=1 python tools/exp_arena_regions.py --gb 64 --plain 2 > $out/regions.txt 2>&1
python bench.py > $out/bench1.json 2> $out/bench1.err
python bench.py > $out/bench2.json 2> $out/bench2.err
grep -v "arena alloc\|create \|delete \|update" $out/regions.txt; tail -5 $out/bench1.err
python - <<PY
import json
for i in (1, 2):
    d = json.load(open("$out/bench%d.json" % i))
    o = d.get("operator_level", {})
    print("bench", i, "value %.2f G/s step %.3f ms bnw %.3f scan %.3f setup %.2f s" % (d["value"] / 1e9, d["ms_per_step"], d["kernel_ms"]["bnw"], d["kernel_ms"]["scan"], d["setup_s"]))
    print("  alloc", d["allocator_stats"])
    print("  oplevel", {k: o.get(k) for k in ("noise_filter_s", "mapmaker_s", "pcg_iteration_ms", "phases_s", "error")})
    print("  fft", d["fft_noise_weight"]["ms"], "lhs", d.get("pcg_lhs_offset_templates", {}))
    print("  expansion", d["expansion"])
    print("  stream", d["roofline"]["stream_ceiling"])
PY
