#!/usr/bin/env python3
"""Assemble profiles/rNN_x_final_rocprofv3.txt from the files tools/gpu_final_profile.sh leaves in gpurun_out/<tag>/:
    final_profile_text.py gpurun_out/<tag> "<header line(s)>" > profiles/rNN_x_final_rocprofv3.txt
and copy <tag>/traffic_cfg3.json, <tag>/traffic_exact_cfg3.json to profiles/ by hand."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_line  # noqa: E402

d, header = sys.argv[1], sys.argv[2]
if os.path.exists(os.path.join(d, "PLACEMENT_FAILED")):
    sys.exit("final_profile_text: %s/PLACEMENT_FAILED exists -- the traced process' zone placement did not work out, "
             "its kernel durations are not those of the bench line" % d)
print("# " + header.replace("\n", "\n# "))
print("# tools/gpu_final_profile.sh: the default bench.py line, then the same command under rocprofv3 (kernel trace; FETCH_SIZE / WRITE_SIZE")
print("# with --unfused; the exact 32-byte-unit DRAM counters, one pass each).  profiles/traffic_cfg3.json and traffic_exact_cfg3.json are")
print("# computed from these passes.\n")
b = json.load(open(os.path.join(d, "bench.json")))
print("== 1  python bench.py (un-profiled) ==")
buf = io.StringIO()
with redirect_stdout(buf):
    bench_line.line("  line", b, True)
print(buf.getvalue().rstrip())
for k in ("roofline", "cpu_baseline", "fft_noise_weight"):
    print("  %s: %s" % (k, json.dumps(b.get(k))))
print("  pcg_lhs_offset_templates: %s" % json.dumps({k: v for k, v in (b.get("pcg_lhs_offset_templates") or {}).items() if not isinstance(v, dict)}))
print("\n== 2  rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline ==")
t = json.loads(open(os.path.join(d, "trace.json")).read().strip().splitlines()[-1])
ta, tk = t["allocator_stats"], t["kernel_ms"]
print("  traced process: placement_ok %s, search_exhausted %s, chunks_other_zone %d of %d wanted (%d chunks, %d created, %.1f ms per "
      "hipMemCreate, %d probes of which %d by the device clock)" % (
          ta["placement_ok"], ta["search_exhausted"], ta["chunks_other_zone"], ta["chunks_other_wanted"], ta["chunks"], ta["chunks_created"],
          ta["create_ms_per_chunk"], ta["probes"], ta["probes_by_clock"]))
print("  its own HIP-event kernel_ms: bnw %.3f  scan %.3f   (un-profiled line above: bnw %.3f  scan %.3f)" % (
    tk["bnw"], tk["scan"], b["kernel_ms"]["bnw"], b["kernel_ms"]["scan"]))
print(open(os.path.join(d, "trace.txt")).read().rstrip())
print("\n== 3  FETCH_SIZE / WRITE_SIZE passes (--unfused) ==")
for f in ("fetch.txt", "write.txt"):
    print(open(os.path.join(d, f)).read().rstrip())
print("\n== 4  exact DRAM counters (32 B x TCC_EA0_RDREQ_DRAM_32B / WRREQ_WRITE_DRAM_32B / WRREQ_WRITE_ATOMIC_32B), bytes per launch ==")
print(open(os.path.join(d, "exact_bytes.txt")).read().rstrip())
