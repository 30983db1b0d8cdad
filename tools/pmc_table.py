#!/usr/bin/env python3
"""Per kernel and counter: average value per launch (summed over all instances / dimensions of the counter), from the
rocprofv3 --pmc databases under a directory.   pmc_table.py <dir> [kernel-name filter]"""
import glob
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("toast_hip::", "").replace("fused_fft::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:56]


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    res = {}
    for db in glob.glob(root + "/**/*.db", recursive=True):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        view = [t for t in tabs if t == "counters_collection"] or [t for t in tabs if "counters_collection" in t]
        if not view:
            continue
        cols = [r[1] for r in con.execute("pragma table_info(%s)" % view[0])]
        disp = "dispatch_id" if "dispatch_id" in cols else "id"
        q = "select kernel_name, counter_name, %s, sum(value) from %s group by kernel_name, counter_name, %s" % (disp, view[0], disp)
        for name, counter, _, val in con.execute(q):
            res.setdefault(short(name), {}).setdefault(counter, []).append(val)
    for name in sorted(res):
        if flt and flt not in name:
            continue
        print("== %s" % name)
        for c in sorted(res[name]):
            v = res[name][c]
            print("   %-32s %6d launches  avg %18.1f" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
