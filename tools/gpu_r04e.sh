#!/bin/bash
# round 4: chunk size of the interleaved slabs against build_noise_weighted (read-only) and scan_map (read + write).  $1 = tag
tag=${1:-r04e}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for cfg in "1 1024" "1 2048" "1 4096" "0 1024" "1 512" "1 1024"; do
  set -- $cfg
  TOAST_HIP_TRACE=1 TOAST_HIP_ARENA_INTERLEAVE=$1 TOAST_HIP_ARENA_CHUNK_MB=$2 python bench.py --no-operator-level --no-cpu-baseline --no-fft > $out/bench_$1_$2.json 2> $out/bench_$1_$2.err
  grep "vmm slab" $out/bench_$1_$2.err
  python - <<PY
import json
d = json.load(open("$out/bench_$1_$2.json"))
a = d["allocator_stats"]
l = d.get("pcg_lhs_offset_templates", {})
print("interleave $1 chunk $2 MB: %.2f G/s step %.3f bnw %.3f scan %.3f setup %.2f s | rw %.0f GB/s | pix %.1f sw %.1f G/s | lhs seq %.2f fused %.2f packed %.2f | other-zone %d/%d created %d, malloc %.0f ms"
      % (d["value"] / 1e9, d["ms_per_step"], d["kernel_ms"]["bnw"], d["kernel_ms"]["scan"], d["setup_s"],
         d["roofline"]["stream_ceiling"]["read_write_GBs"], d["expansion"]["pixels_healpix_Gsamp_s"], d["expansion"]["stokes_weights_IQU_Gsamp_s"],
         l.get("operator_sequence_ms", 0), l.get("fused_ms", 0), l.get("packed_ms", 0),
         a["chunks_other_zone"], a["chunks"], a["chunks_created"], a["malloc_ms"]))
PY
done
