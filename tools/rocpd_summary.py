#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) as text:
per-kernel call count, average / min / max duration, register and LDS use, and the average of
every PMC counter collected.  Usage: rocpd_summary.py <results.db> [<results.db> ...]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    name = re.sub(r"\(.*", "", name)
    return name[-60:]


def main(paths):
    for p in paths:
        con = sqlite3.connect(p)
        cur = con.cursor()
        print("==", p)
        rows = list(cur.execute(
            "select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
            "max(lds_size), max(grid_x), max(grid_y) from kernels group by name order by sum(duration) desc"))
        if rows:
            print("%-44s %6s %12s %12s %12s %5s %5s %6s %s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "vgpr", "sgpr", "lds", "grid"))
            for r in rows:
                print("%-44s %6d %12.1f %12.1f %12.1f %5s %5s %6s %sx%s" % (short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], r[6], r[7], r[8], r[9]))
        try:
            rows = list(cur.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                "group by kernel_name, counter_name order by kernel_name"))
        except sqlite3.OperationalError:
            rows = []
        if rows:
            print("%-44s %-14s %6s %16s %16s %16s" % ("kernel", "counter", "n", "avg", "min", "max"))
            for r in rows:
                print("%-44s %-14s %6d %16.1f %16.1f %16.1f" % (short(r[0]), r[1], r[2], r[3], r[4], r[5]))


def traffic_json(fetch_db, write_db, out_path, workload, note=""):
    """HBM bytes per launch for our kernels = 2 x FETCH_SIZE + WRITE_SIZE (KiB), with the gfx950
    FETCH_SIZE correction calibrated on k_noise_weight in the same run (profiles/README.md)."""
    import json

    def avg(db, counter):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, avg(value) from counters_collection where counter_name=? "
                           "group by kernel_name", (counter,))
        return {short(r[0]): r[1] for r in rows if "k_" in r[0]}

    fetch, write = avg(fetch_db, "FETCH_SIZE"), avg(write_db, "WRITE_SIZE")
    out = {"workload": workload, "unit": "bytes per launch", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "note": note, "kernels": {}}
    for k in sorted(set(fetch) & set(write)):
        out["kernels"][k] = {"fetch_KiB_raw": fetch[k], "write_KiB": write[k],
                             "hbm_bytes": (2.0 * fetch[k] + write[k]) * 1024.0}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        traffic_json(*sys.argv[2:])
    else:
        main(sys.argv[1:])
