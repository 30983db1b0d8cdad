#!/bin/bash
# The default bench.py line as the FIRST bench process on a box after the complete GPU test suite and smoke() -- the order
# in which the driver runs things at the end of a round -- followed by two more default lines on the same box.
# $1 = tag.  Run on the GPU box.
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -1 > $out/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $out/smoke.txt
for i in 1 2 3; do
  python bench.py > $out/bench$i.json 2> $out/bench$i.err
  python - <<PY
import json
d = json.load(open("$out/bench$i.json"))
a = d["allocator_stats"]
o = d.get("operator_level", {})
print("bench %d  value %.2f G/s  step %.3f ms  bnw %.3f  scan %.3f  frac %.3f  setup %.2f s  probed %d (%d fast) %d candidates, max hipMalloc %.0f ms, held %.1f GB | fft %.2f ms | operator level: NoiseFilter %.2f s MapMaker %.2f s PCG %.2f ms"
      % ($i, d["value"] / 1e9, d["ms_per_step"], d["kernel_ms"]["bnw"], d["kernel_ms"]["scan"], d["roofline"]["frac"], d["setup_s"],
         a["probed_blocks"], a["fast_blocks"], a["candidates"], a["max_malloc_ms"], a["held_GB"], d["fft_noise_weight"]["ms"],
         o.get("noise_filter_s", 0), o.get("mapmaker_s", 0), o.get("pcg_iteration_ms", 0))
      + " | LHS on the bench buffers: sequence %.2f fused %.2f packed %.2f ms" % tuple(
          d.get("pcg_lhs_offset_templates", {}).get(k, 0.0) for k in ("operator_sequence_ms", "fused_ms", "packed_ms")))
PY
done | tee $out/lines.txt
cat $out/tests.txt $out/smoke.txt
