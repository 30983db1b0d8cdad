#!/usr/bin/env python3
"""EXPERIMENT (profiles/r05_d): the full pairwise map of a process' memory.

A plain allocation of --plain GB (the "read-mostly slab") and --chunks separately created 1 GB chunks (creation order): every
pair of 1 GB ranges gets one read + write pass with its rows dealt alternately to the two (toast_hip_probe_stream_split);
the rate says whether the two ranges slow each other down.  Prints the matrix as digits (rate level) and the levels found.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd import capi  # noqa: E402

GB = 1 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plain", type=int, default=64)
    ap.add_argument("--chunks", type=int, default=48)
    ap.add_argument("--step", type=int, default=4, help="every step-th GB of the plain allocation takes part")
    args = ap.parse_args()
    capi.accel_assign_device(1, 0, 0.0, False)
    os.environ.setdefault("TOAST_HIP_ARENA_STREAM_GB", "0")
    base = capi.device_malloc(args.plain * GB, 0)
    capi.probe_stream(base, args.plain * GB)
    v = capi.device_malloc_vmm(args.chunks * GB, 1024, False)
    capi.probe_stream(v, args.chunks * GB)
    names, ptrs = [], []
    for x in range(0, args.plain, args.step):
        names.append("p%02d" % x)
        ptrs.append(base + x * GB)
    for c in range(args.chunks):
        names.append("c%02d" % c)
        ptrs.append(v + c * GB)
    n = len(ptrs)
    rate = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            ms = capi.probe_stream_split([ptrs[i], ptrs[j]], GB)
            rate[i, j] = rate[j, i] = 4.0 * GB / ms / 1e9 if ms > 0 else 0.0
    vals = rate[np.triu_indices(n, 1)]
    hist, edges = np.histogram(vals, bins=24)
    print("rates TB/s (x 1e-3 of GB/s): histogram over all pairs")
    for h, e0, e1 in zip(hist, edges[:-1], edges[1:]):
        print("  %.3f - %.3f  %5d %s" % (e0 / 1e3, e1 / 1e3, h, "#" * int(60 * h / max(hist))))
    lo, hi = np.percentile(vals, 2), np.percentile(vals, 98)
    print("matrix: digit = 9 x (rate - p2) / (p98 - p2), p2 = %.3f, p98 = %.3f TB/s; rows / columns: %s .. %s" % (lo / 1e3, hi / 1e3, names[0], names[-1]))
    for i in range(n):
        row = ""
        for j in range(n):
            if i == j:
                row += "."
            else:
                row += str(int(np.clip(9.0 * (rate[i, j] - lo) / (hi - lo), 0, 9)))
        print("  %s %s" % (names[i], row))


if __name__ == "__main__":
    main()
