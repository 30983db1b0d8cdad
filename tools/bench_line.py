#!/usr/bin/env python3
"""One summary line (plus allocator / phases lines with -v) per bench.py JSON file: `bench_line.py [-v] LABEL=FILE ...`.

Keys that a line does not carry (older rounds, `--no-*` flags) print as 0.
"""
import json
import sys


def g(d, *keys, default=0.0):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d if d is not None else default


def line(label, d, verbose):
    out = ("%s: %.2f G/s step %.3f ms bnw %.3f scan %.3f frac %.3f setup %.2f s | rw %.0f GB/s | pix %.1f sw %.1f G/s | fft %.2f long %.2f ms"
           " | lhs seq %.2f fused %.2f packed %.2f | NoiseFilter %.3f s MapMaker %.3f s PCG %.2f ms %s") % (
        label, g(d, "value") / 1e9, g(d, "ms_per_step"), g(d, "kernel_ms", "bnw"), g(d, "kernel_ms", "scan"), g(d, "roofline", "frac"),
        g(d, "setup_s"), g(d, "roofline", "stream_ceiling", "read_write_GBs"), g(d, "expansion", "pixels_healpix_Gsamp_s"),
        g(d, "expansion", "stokes_weights_IQU_Gsamp_s"), g(d, "fft_noise_weight", "ms"), g(d, "fft_noise_weight", "long", "ms"),
        g(d, "pcg_lhs_offset_templates", "operator_sequence_ms"), g(d, "pcg_lhs_offset_templates", "fused_ms"),
        g(d, "pcg_lhs_offset_templates", "packed_ms"), g(d, "operator_level", "noise_filter_s"), g(d, "operator_level", "mapmaker_s"),
        g(d, "operator_level", "pcg_iteration_ms"), g(d, "operator_level", "error", default=""))
    print(out)
    if verbose:
        a = g(d, "allocator_stats", default={})
        print("   alloc: slabs %d (%d interleaved) %.0f GB, peak used %.1f GB, hipMalloc calls %d, %.0f ms in them (max %.0f), other-zone %d/%d created %d"
              % tuple(g(a, k) for k in ("slabs", "interleaved_slabs", "slab_GB", "peak_used_GB", "slab_mallocs", "malloc_ms", "max_malloc_ms",
                                        "chunks_other_zone", "chunks", "chunks_created")))
        print("   phases", g(d, "operator_level", "phases_s", default=None))


if __name__ == "__main__":
    args = sys.argv[1:]
    verbose = "-v" in args
    for a in args:
        if a == "-v":
            continue
        label, _, path = a.rpartition("=")
        try:
            with open(path) as f:
                line(label or path, json.load(f), verbose)
        except (OSError, ValueError) as e:
            print("%s: unreadable (%s)" % (label or path, e))
