#!/usr/bin/env python3
"""EXPERIMENT (profiles/r04_a): where inside one large slab does the time-major access pattern stream fast?

Takes one slab of --gb from the driver (timed), touches it, then measures a read + write pass with 1024 rows in flight
(toast_hip_probe_stream = k_probe_stream, the pattern of every TOD-domain kernel) over
  * windows of 5.9 GB (one cfg-3 timestream) at every 2 GB offset,
  * every 1 GB region on its own,
and, for comparison, over separately hipMalloc'ed blocks of 5.9 GB.  Prints rates in TB/s (read + write bytes / time).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=64.0)
    ap.add_argument("--plain", type=int, default=6, help="separately allocated 5.9 GB blocks to compare with")
    args = ap.parse_args()
    capi.accel_assign_device(1, 0, 0.0, False)
    GB = 1 << 30
    n = int(args.gb * GB)
    t0 = time.time()
    capi.arena_reserve(n)
    t1 = time.time()
    base = capi.device_malloc(n - (64 << 20))
    capi.synchronize()
    t2 = time.time()
    st = capi.alloc_stats()
    print(f"slab {args.gb:.0f} GB: reserve {1e3 * (t1 - t0):.1f} ms (hipMalloc {st['malloc_ms']:.1f} ms), first touch done after "
          f"{1e3 * (t2 - t1):.1f} ms; slabs {st['slabs']}")
    win = 1024 * 720000 * 8
    rate = lambda nbytes, ms: 2.0 * nbytes / ms / 1e9
    print("5.9 GB windows at 2 GB steps (TB/s):")
    row = []
    off = 0
    while off + win <= n - (64 << 20):
        row.append(rate(win, capi.probe_stream(base + off, win)))
        off += 2 * GB
    print("  " + " ".join(f"{r:.2f}" for r in row))
    print("1 GB regions (TB/s):")
    row = []
    for k in range(int((n - (64 << 20)) // GB)):
        row.append(rate(GB, capi.probe_stream(base + k * GB, GB)))
    print("  " + " ".join(f"{r:.2f}" for r in row))
    print("again, 5.9 GB windows (stability):")
    row = []
    off = 0
    while off + win <= n - (64 << 20):
        row.append(rate(win, capi.probe_stream(base + off, win)))
        off += 2 * GB
    print("  " + " ".join(f"{r:.2f}" for r in row))
    blocks = []
    row = []
    tm = []
    for _ in range(args.plain):
        t0 = time.time()
        p = capi.device_malloc(win, 0)
        tm.append(1e3 * (time.time() - t0))
        blocks.append(p)
        capi.probe_stream(p, win)
        row.append(rate(win, capi.probe_stream(p, win)))
    print("separate hipMalloc blocks of 5.9 GB (TB/s):  " + " ".join(f"{r:.2f}" for r in row))
    print("  their hipMalloc ms: " + " ".join(f"{t:.1f}" for t in tm))


if __name__ == "__main__":
    main()
