#!/bin/bash
# round 4: rank-interleaved slabs on and off, alternating processes on one box.  $1 = tag
tag=${1:-r04c}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for rep in 1 2; do
  for il in 1 0; do
    TOAST_HIP_TRACE=1 TOAST_HIP_ARENA_INTERLEAVE=$il python bench.py --no-operator-level --no-cpu-baseline > $out/bench_il${il}_$rep.json 2> $out/bench_il${il}_$rep.err
    grep "vmm slab" $out/bench_il${il}_$rep.err
    python - <<PY
import json
d = json.load(open("$out/bench_il${il}_$rep.json"))
a = d["allocator_stats"]
l = d.get("pcg_lhs_offset_templates", {})
print("interleave $il rep $rep: %.2f G/s step %.3f bnw %.3f scan %.3f setup %.2f s | rw %.0f GB/s | pix %.1f sw %.1f G/s otf %.2f %.2f ms | fft %.2f | lhs seq %.2f fused %.2f packed %.2f | slabs %d (%d il) %.0f GB, other-zone %d/%d created %d, malloc %.0f ms"
      % (d["value"] / 1e9, d["ms_per_step"], d["kernel_ms"]["bnw"], d["kernel_ms"]["scan"], d["setup_s"],
         d["roofline"]["stream_ceiling"]["read_write_GBs"], d["expansion"]["pixels_healpix_Gsamp_s"], d["expansion"]["stokes_weights_IQU_Gsamp_s"],
         d["expansion"]["from_boresight_pixels_ms"], d["expansion"]["from_boresight_weights_ms"], d["fft_noise_weight"]["ms"],
         l.get("operator_sequence_ms", 0), l.get("fused_ms", 0), l.get("packed_ms", 0),
         a["slabs"], a["interleaved_slabs"], a["slab_GB"], a["chunks_other_zone"], a["chunks"], a["chunks_created"], a["malloc_ms"]))
PY
  done
done
