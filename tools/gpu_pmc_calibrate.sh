#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against KNOWN byte counts per access pattern (profiles/r03_d): tools/ubench/strided_copy
# (128-byte pieces at a 32 KB stride, the FFT column passes' pattern; every launch reads and writes 4 294 967 296 B)
# and tools/ubench/stream_ceiling (contiguous 8 / 16 B per lane streams over 5 898 240 000 B arrays).
out=$GRAFT_REPO_ROOT/gpurun_out/r03d
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 900 rocprofv3 --pmc $c -d /tmp/cal_$c -o sc -- $GRAFT_REPO_ROOT/tools/ubench/strided_copy > /dev/null 2>&1
  timeout -k 5 900 rocprofv3 --pmc $c -d /tmp/cal2_$c -o st -- $GRAFT_REPO_ROOT/tools/ubench/stream_ceiling > /dev/null 2>&1
done
python3 - <<'PY' > $out/pmc_calibration.txt
import sqlite3, glob
def counters(pattern):
    out = {}
    for db in glob.glob(pattern, recursive=True):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        view = [t for t in tabs if t == 'counters_collection'] or [t for t in tabs if 'counters_collection' in t]
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % view[0])]
        q = "select kernel_name, counter_name, value, grid_size_x from %s" % view[0] if 'grid_size_x' in cols else "select kernel_name, counter_name, value, 0 from %s" % view[0]
        for name, cname, val, gx in con.execute(q):
            out.setdefault((name.split('(')[0][:60], gx, cname), []).append(val)
    return out
for tag, known in (("cal_", "strided_copy: 4 294 967 296 B read and written per launch"), ("cal2_", "stream_ceiling: arrays of 5 898 240 000 B (x3 for the weights)")):
    print("==", known)
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for (name, gx, cname), vals in counters("/tmp/%s%s/**/*.db" % (tag, c)).items():
            res.setdefault((name, gx), {})[cname] = (sum(vals) / len(vals), len(vals))
    for (name, gx), d in sorted(res.items()):
        f = d.get("FETCH_SIZE", (0, 0)); w = d.get("WRITE_SIZE", (0, 0))
        print("  %-60s grid %-9s launches %3d  FETCH_SIZE %14.0f KiB = %8.3f GB   WRITE_SIZE %14.0f KiB = %8.3f GB" % (name, gx, f[1], f[0], f[0] * 1024 / 1e9, w[0], w[0] * 1024 / 1e9))
PY
cat $out/pmc_calibration.txt
