#!/bin/bash
# round 4: the tree in the driver's order -- GPU tests, smoke, then the default bench line three times.  $1 = tag
tag=${1:-r04f}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $out/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $out/smoke.txt
for i in 1 2 3; do
  python bench.py > $out/bench$i.json 2> $out/bench$i.err
  python - <<PY
import json
d = json.load(open("$out/bench$i.json"))
a = d["allocator_stats"]
o = d.get("operator_level", {})
l = d.get("pcg_lhs_offset_templates", {})
print("bench $i: %.2f G/s step %.3f bnw %.3f scan %.3f frac %.3f setup %.2f s | rw %.0f GB/s | fft %.2f | lhs seq %.2f fused %.2f packed %.2f | op level: NoiseFilter %.3f s MapMaker %.3f s PCG %.2f ms %s"
      % (d["value"] / 1e9, d["ms_per_step"], d["kernel_ms"]["bnw"], d["kernel_ms"]["scan"], d["roofline"]["frac"], d["setup_s"],
         d["roofline"]["stream_ceiling"]["read_write_GBs"], d["fft_noise_weight"]["ms"],
         l.get("operator_sequence_ms", 0), l.get("fused_ms", 0), l.get("packed_ms", 0),
         o.get("noise_filter_s", 0), o.get("mapmaker_s", 0), o.get("pcg_iteration_ms", 0), o.get("error", "")))
print("   alloc: slabs %d (%d interleaved) %.0f GB, peak used %.1f GB, hipMalloc calls %d, %.0f ms in them (max %.0f), other-zone %d/%d created %d"
      % (a["slabs"], a["interleaved_slabs"], a["slab_GB"], a["peak_used_GB"], a["slab_mallocs"], a["malloc_ms"], a["max_malloc_ms"],
         a["chunks_other_zone"], a["chunks"], a["chunks_created"]))
print("   phases", o.get("phases_s"))
PY
done | tee $out/lines.txt
cat $out/tests.txt $out/smoke.txt
