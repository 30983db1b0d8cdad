#!/bin/bash
# round 4, first look at the arena: the region experiment, the default bench line, the accel tests.  $1 = tag
tag=${1:-r04a}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python tools/exp_arena_regions.py --gb 96 > $out/regions.txt 2>&1
python bench.py > $out/bench1.json 2> $out/bench1.err
python -m pytest tests/test_gpu_accel.py -x -q 2>&1 | tail -15 > $out/accel_tests.txt
python bench.py > $out/bench2.json 2> $out/bench2.err
cat $out/regions.txt; tail -5 $out/bench1.err; cat $out/accel_tests.txt
python - <<PY
import json
for i in (1, 2):
    d = json.load(open("$out/bench%d.json" % i))
    o = d.get("operator_level", {})
    print("bench", i, "value %.2f G/s step %.3f ms bnw %.3f scan %.3f setup %.2f s" % (d["value"] / 1e9, d["ms_per_step"], d["kernel_ms"]["bnw"], d["kernel_ms"]["scan"], d["setup_s"]))
    print("  alloc", d["allocator_stats"])
    print("  oplevel", {k: o.get(k) for k in ("noise_filter_s", "mapmaker_s", "pcg_iteration_ms", "phases_s", "error")})
    print("  fft", d["fft_noise_weight"]["ms"], "lhs", d.get("pcg_lhs_offset_templates", {}).get("packed_ms"))
PY
