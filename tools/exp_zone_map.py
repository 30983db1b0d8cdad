#!/usr/bin/env python3
"""EXPERIMENT (profiles/r04_a section 3): the zone map of a large allocation.

Rows that are in flight together stream at 6.1 instead of 5.05 TB/s when they come from two different "zones" of a
slab.  This maps the zones: a 1 GB reference range against every other 1 GB range of the allocation (512 rows each,
toast_hip_probe_stream_split); ranges in the reference's own zone give the slow level, ranges elsewhere the fast one.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd import capi  # noqa: E402

GB = 1 << 30


def zone_map(base, n_gb, ref_gb, each=GB, step_gb=1):
    rate = lambda nbytes, ms: 2.0 * nbytes / ms / 1e9
    out = []
    for x in range(0, n_gb, step_gb):
        if abs(x - ref_gb) * GB < each:
            out.append(None)
            continue
        out.append(rate(2 * each, capi.probe_stream_split([base + ref_gb * GB, base + x * GB], each)))
    return out


def show(title, m):
    print(title)
    print("  " + "".join("." if v is None else ("F" if v > 5.6 else "s") for v in m))
    vals = [v for v in m if v is not None]
    print(f"  min {min(vals):.2f} max {max(vals):.2f} TB/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=int, default=224)
    ap.add_argument("--vmm-gb", type=int, default=48)
    args = ap.parse_args()
    capi.accel_assign_device(1, 0, 0.0, False)
    n = args.gb * GB
    base = capi.device_malloc(n, 0)
    capi.probe_stream(base, n)
    print(f"allocation of {args.gb} GB at {base:#x}")
    m0 = zone_map(base, args.gb, 1)
    show("reference = GB 1 (F: fast together with the reference, s: slow)", m0)
    other = next((i for i, v in enumerate(m0) if v is not None and v > 5.6), None)
    if other is not None:
        m1 = zone_map(base, args.gb, other)
        show(f"reference = GB {other}", m1)
        # a third reference: the first range that is fast with both
        third = next((i for i in range(args.gb) if m0[i] and m1[i] and m0[i] > 5.6 and m1[i] > 5.6), None)
        if third is not None:
            show(f"reference = GB {third}", zone_map(base, args.gb, third))
    # do small ranges show the same map?  (128 MB each: 512 rows of 256 KB)
    show("reference = GB 1, ranges of 128 MB", zone_map(base, args.gb, 1, each=128 << 20, step_gb=2))
    show("reference = GB 1, ranges of 16 MB", zone_map(base, args.gb, 1, each=16 << 20, step_gb=2))
    capi.device_free(base)
    # the same for a virtual range built from separately created 1 GB physical chunks, mapped in the order of creation
    try:
        v = capi.device_malloc_vmm(args.vmm_gb * GB, 1024, False)
        capi.probe_stream(v, args.vmm_gb * GB)
        show(f"VMM range of {args.vmm_gb} x 1 GB chunks, reference = GB 1", zone_map(v, args.vmm_gb, 1))
    except (RuntimeError, AttributeError) as err:
        print("VMM:", err)


if __name__ == "__main__":
    main()
