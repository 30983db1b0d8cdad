#!/usr/bin/env python3
"""Condense a TOAST_HIP_TRACE=2 log (serialised timeline of the library calls, stderr) into the calls that matter:
every call longer than --min ms or preceded by a host gap longer than --gap ms, the transfers and the phase marks.

    TOAST_HIP_TRACE=2 python workflows/mapmaker_pcg.py 2> trace.log;  python tools/trace_timeline.py trace.log
"""
import argparse
import re


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log")
    ap.add_argument("--min", type=float, default=0.5)
    ap.add_argument("--gap", type=float, default=1.0)
    args = ap.parse_args()
    prev_end, n, acc = None, 0, 0.0

    def flush():
        nonlocal n, acc
        if n:
            print(f"      ... {n} short calls {acc:.2f} ms")
        n, acc = 0, 0.0

    for line in open(args.log):
        m = re.match(r"\[toast_hip\] call\s+([\d.]+) ms\s+\+\s*([\d.]+) ms\s+(\S+)", line)
        if m:
            t, d, fn = float(m.group(1)), float(m.group(2)), m.group(3)
            gap = t - prev_end if prev_end is not None else 0.0
            prev_end = max(prev_end or 0.0, t + d)
            if d > args.min or gap > args.gap:
                flush()
                print(f"{t:9.2f} gap {gap:7.2f} dur {d:8.2f}  {fn}")
            else:
                n += 1
                acc += d
        elif "] phase" in line or "] update_" in line:
            flush()
            print("   ", line.rstrip()[:118])
    flush()


if __name__ == "__main__":
    main()
