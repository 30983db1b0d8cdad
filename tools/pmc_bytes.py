#!/usr/bin/env python3
"""Per kernel: bytes per launch = 32 x (sum over all TCC instances of a *_32B counter), from rocprofv3 --pmc databases.

    pmc_bytes.py [--json OUT.json WORKLOAD] <directory prefix> <title> [<prefix> <title> ...]
                                                                       (directories <prefix><COUNTER>/**/*.db)
    --json: the LAST prefix' kernels also as profiles/traffic_exact_<workload>.json (what bench.py's roofline.traffic_exact reads)
"""
import glob
import sqlite3
import sys

COUNTERS = ("TCC_EA0_RDREQ_DRAM_32B", "TCC_EA0_WRREQ_WRITE_DRAM_32B", "TCC_EA0_WRREQ_WRITE_ATOMIC_32B", "TCC_BUBBLE")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("toast_hip::", "").replace("fused_fft::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:64]


def per_launch(db, counter):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    view = [t for t in tabs if t == "counters_collection"] or [t for t in tabs if "counters_collection" in t]
    if not view:
        return {}
    cols = [r[1] for r in con.execute("pragma table_info(%s)" % view[0])]
    disp = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else None)
    out = {}
    if disp is None:
        for name, val in con.execute("select kernel_name, value from %s where counter_name = ?" % view[0], (counter,)):
            out.setdefault(short(name), []).append(val)
        return out
    q = ("select kernel_name, %s, sum(value), count(*) from %s where counter_name = ? group by kernel_name, %s"
         % (disp, view[0], disp))
    for name, _, val, rows in con.execute(q, (counter,)):
        out.setdefault(short(name), []).append(val)
    return out


def main():
    args = sys.argv[1:]
    json_out = workload = None
    if args and args[0] == "--json":
        json_out, workload, args = args[1], args[2], args[3:]
    res = {}
    for prefix, title in zip(args[0::2], args[1::2]):
        print("==", title)
        res = {}
        for c in COUNTERS:
            for db in glob.glob(prefix + c + "/**/*.db", recursive=True):
                for name, vals in per_launch(db, c).items():
                    res.setdefault(name, {})[c] = (sum(vals) / len(vals), len(vals))
        print("  %-44s %8s %12s %12s %12s %14s" % ("kernel", "launches", "read GB", "written GB", "atomics GB", "128-B reads GB"))
        for name, d in sorted(res.items()):
            g = lambda c, unit: d.get(c, (0.0, 0))[0] * unit / 1e9
            n = max(v[1] for v in d.values())
            print("  %-44s %8d %12.3f %12.3f %12.3f %14.3f" % (name[:44], n, g(COUNTERS[0], 32), g(COUNTERS[1], 32),
                                                             g(COUNTERS[2], 32), g(COUNTERS[3], 128)))
    if json_out:
        import json

        kernels = {}
        for name, d in sorted(res.items()):
            b = lambda c: float(d.get(c, (0.0, 0))[0] * 32)
            kernels[name] = {"launches": max(v[1] for v in d.values()), "read_bytes": b(COUNTERS[0]), "written_bytes": b(COUNTERS[1]),
                             "atomic_bytes": b(COUNTERS[2]), "hbm_bytes": b(COUNTERS[0]) + b(COUNTERS[1]) + b(COUNTERS[2])}
        with open(json_out, "w") as f:
            json.dump({"workload": workload, "unit": "bytes per launch",
                       "counters": "32 B x TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B / TCC_EA0_WRREQ_WRITE_ATOMIC_32B summed over "
                                   "all TCC instances, one rocprofv3 --pmc pass per counter (tools/gpu_final_profile.sh); exact on known "
                                   "byte counts (profiles/r04_e)",
                       "kernels": kernels}, f, indent=1)


if __name__ == "__main__":
    main()
