#!/bin/bash
# Kernel trace of the configs[1]-size MapMaker (30 PCG iterations) with the PCG scalars on the device: how much of an
# iteration is kernel time (profiles/r03_c).  Run on the GPU box.
# usage: gpu_pcg_trace.sh [out-dir-name]   (TOAST_HIP_PCG_FUSE / TOAST_HIP_PCG_SCALARS from the environment)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r03c}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pcgprof
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d /tmp/pcgprof -o pcg -- python3 $GRAFT_REPO_ROOT/workflows/mapmaker_pcg.py --ndet 64 --minutes 60 --rate 100 --nside 512 --iter 30 --no-filter > $out/pcg_trace_stdout.txt 2>/tmp/pcgprof.err
python3 - <<'PY' > $out/pcg_trace.txt
import sqlite3, glob, re
db = glob.glob('/tmp/pcgprof/*.db')[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name, start, end from kernels order by start"))
# the PCG loop = from the first k_pcg_stage to the last
idx = [i for i, r in enumerate(rows) if 'k_pcg_dot_stage' in r[0] or 'k_pcg_stage' in r[0]]
lo, hi = idx[0], idx[-1]
loop = rows[lo:hi + 1]
busy = sum(e - s for _, s, e in loop)
wall = loop[-1][2] - loop[0][1]
n_it = len(idx) // 3
print("PCG loop: %d dot+stage kernels = %d iterations, %d kernels, wall %.3f ms, kernel time %.3f ms (%.1f %%), per iteration wall %.3f ms / kernels %.3f ms / %d launches"
      % (len(idx), n_it, len(loop), wall / 1e6, busy / 1e6, 100.0 * busy / wall, wall / 1e6 / n_it, busy / 1e6 / n_it, len(loop) // n_it))
agg = {}
for n, s, e in loop:
    m = re.search(r"(k_[a-z_0-9]+)", n)
    k = m.group(1) if m else n[:40]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-36s %5d calls  %9.1f us total  %8.2f us avg" % (k, c, t / 1e3, t / 1e3 / c))
gaps = sorted((loop[i + 1][1] - loop[i][2]) / 1e3 for i in range(len(loop) - 1))
print("gaps between consecutive kernels: median %.2f us, mean %.2f us, max %.1f us" % (gaps[len(gaps) // 2], sum(gaps) / len(gaps), gaps[-1]))
PY
cat $out/pcg_trace.txt; grep "PCG iteration" $out/pcg_trace_stdout.txt
