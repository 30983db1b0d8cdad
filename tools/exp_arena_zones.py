#!/usr/bin/env python3
"""EXPERIMENT (profiles/r04_a section 2): the 32 GB period of the fast windows inside a slab.

  1. fine scan of the start offset of a 5.9 GB window across the 32 GB offset,
  2. rows dealt to SEPARATE ranges: both halves in one 32 GB zone / in two zones / four quarters in four zones,
  3. a 17.7 GB window (cfg-3 weights) inside a zone and across a boundary,
  4. a second slab: where are its boundaries (relative to its base address)?
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd import capi  # noqa: E402

GB = 1 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=136.0)
    args = ap.parse_args()
    capi.accel_assign_device(1, 0, 0.0, False)
    n = int(args.gb * GB)
    capi.arena_reserve(n)
    base = capi.device_malloc(n - (64 << 20))
    capi.synchronize()
    print(f"slab base address {base:#x} (mod 32 GB = {base % (32 * GB) / GB:.3f} GB, mod 2 MB = {base % (2 << 20)})")
    win = 1024 * 720000 * 8
    rate = lambda nbytes, ms: 2.0 * nbytes / ms / 1e9
    capi.probe_stream(base, n - (64 << 20))      # first touch of everything
    print("1. 5.9 GB window, start offset 24 .. 33.5 GB in 0.5 GB steps (TB/s):")
    print("  " + " ".join(f"{24 + 0.5 * k:.1f}:{rate(win, capi.probe_stream(base + 24 * GB + k * GB // 2, win)):.2f}"
                          for k in range(20)))
    half, quarter = win // 2, win // 4
    z = lambda zone, off_gb: base + zone * 32 * GB + int(off_gb * GB)
    print("2. rows dealt round-robin to separate ranges (TB/s):")
    for name, ptrs, each in (
        ("one range (zone 0)", [z(0, 2)], win),
        ("two halves, both in zone 0 (2 GB, 18 GB)", [z(0, 2), z(0, 18)], half),
        ("two halves, zone 0 and zone 1", [z(0, 2), z(1, 2)], half),
        ("two halves, zone 1 and zone 2", [z(1, 9), z(2, 5)], half),
        ("two halves, zone 0 and zone 2", [z(0, 9), z(2, 5)], half),
        ("four quarters, zones 0 1 2 3", [z(0, 2), z(1, 2), z(2, 2), z(3, 2)], quarter),
        ("four quarters, all in zone 0", [z(0, 2), z(0, 10), z(0, 18), z(0, 26)], quarter),
        ("four quarters, zones 0 0 1 1", [z(0, 2), z(0, 18), z(1, 2), z(1, 18)], quarter),
    ):
        if max(ptrs) + each > base + n - (64 << 20):
            continue
        print(f"  {name:46s} {rate(each * len(ptrs), capi.probe_stream_split(ptrs, each)):.2f}")
    big = 3 * win
    print("3. 17.7 GB window (TB/s): " + " ".join(
        f"{o}:{rate(big, capi.probe_stream(base + o * GB, big)):.2f}" for o in (2, 8, 14, 16, 20, 24, 28, 30, 34, 40, 46, 52, 58)
        if (o + 17) * GB < n))
    # a second slab
    n2 = 40 * GB
    try:
        b2 = capi.device_malloc(n2, 0)
        capi.probe_stream(b2, n2)
        print(f"4. second allocation (plain hipMalloc, 40 GB) at {b2:#x} (mod 32 GB = {b2 % (32 * GB) / GB:.3f} GB); 5.9 GB windows at 2 GB steps:")
        print("  " + " ".join(f"{rate(win, capi.probe_stream(b2 + k * 2 * GB, win)):.2f}" for k in range(17)))
    except RuntimeError as err:
        print("4. second allocation failed:", err)


if __name__ == "__main__":
    main()
