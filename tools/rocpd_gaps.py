#!/usr/bin/env python3
"""Where does the GPU idle?  From a rocprofv3 --kernel-trace database: the kernels in start order with the idle time
before each of them; runs of short kernels are folded.  Gaps above `--min` ms are the places where the host (or a
transfer) keeps the device waiting.

    rocpd_gaps.py <results.db> [--min 0.3] [--from <kernel name part>] [--to <kernel name part>]
"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("toast_hip::", "").replace("fused_fft::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:48]


def main():
    args = sys.argv[1:]
    db = args[0]
    opt = dict(zip(args[1::2], args[2::2]))
    gap_min = float(opt.get("--min", 0.3))
    con = sqlite3.connect(db)
    rows = [(s, e, short(n)) for s, e, n in con.execute("select start, end, name from kernels order by start")]
    if "--from" in opt:
        i0 = next(i for i, r in enumerate(rows) if opt["--from"] in r[2])
        rows = rows[i0:]
    if "--to" in opt:
        i1 = max(i for i, r in enumerate(rows) if opt["--to"] in r[2])
        rows = rows[:i1 + 1]
    t0 = rows[0][0]
    busy = sum(e - s for s, e, _ in rows) / 1e6
    span = (rows[-1][1] - t0) / 1e6
    print("kernels %d   span %.2f ms   busy %.2f ms (%.0f %%)" % (len(rows), span, busy, 100 * busy / span))
    prev_end = rows[0][0]
    folded, folded_ms = 0, 0.0
    total_gap = 0.0
    for s, e, n in rows:
        gap = max(0.0, (s - prev_end) / 1e6)
        dur = (e - s) / 1e6
        if gap < gap_min and dur < 0.5:
            folded += 1
            folded_ms += dur
        else:
            if folded:
                print("            ... %d short kernels, %.2f ms" % (folded, folded_ms))
                folded, folded_ms = 0, 0.0
            mark = "  <-- idle %.2f ms" % gap if gap >= gap_min else ""
            print("%9.2f  %8.3f ms  %-48s%s" % ((s - t0) / 1e6, dur, n, mark))
        if gap >= gap_min:
            total_gap += gap
        prev_end = max(prev_end, e)
    if folded:
        print("            ... %d short kernels, %.2f ms" % (folded, folded_ms))
    print("idle in gaps >= %.2f ms: %.2f ms" % (gap_min, total_gap))


if __name__ == "__main__":
    main()
