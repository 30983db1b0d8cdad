#!/usr/bin/env python3
"""Experiment: which host-side pattern makes the next GPU operation (a pageable H2D copy or a tiny kernel) stall for
20-30 ms?  Seen in workflows/mapmaker_pcg.py with TOAST_HIP_TRACE=2 (update_device of a 3.7 MB array: 20-28 ms in the
enqueue of hipMemcpyAsync; a 10 us kernel: 27 ms).  Every variant prints the time of each upload."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd.accel import (accel_data_create, accel_data_delete, accel_data_update_device,  # noqa: E402
                             accel_data_update_host, native)

N = 3_686_400


def upload(a, name="x"):
    accel_data_create(a, name)
    native().accel_synchronize()
    rep = None
    if "--stamps" in sys.argv:
        import ctypes

        try:
            rep = ctypes.CDLL(None).ioctl_trace_report
            rep(b"before")
        except AttributeError:
            rep = None
    t0 = time.perf_counter()
    accel_data_update_device(a, name)
    native().accel_synchronize()
    dt = 1e3 * (time.perf_counter() - t0)
    if rep is not None:
        rep(b"during the upload")
    if "--stamps" in sys.argv:
        print(f"[exp_stall]   {1e3 * time.monotonic():12.2f} ms  upload from {a.ctypes.data:#x} done, took {dt:.2f} ms",
              file=sys.stderr, flush=True)
    return dt


def variant(label, body, reps=6):
    out = []
    for i in range(reps):
        out.append(body(i))
    print(f"{label:58s} " + " ".join(f"{x:7.2f}" for x in out), flush=True)


def main():
    if "--no-big" not in sys.argv:
        big = np.zeros(1 << 28, dtype=np.uint8)   # a large resident buffer (pinned at its transfer)
        big[:] = 1
        accel_data_create(big, "big")
        accel_data_update_device(big, "big")
        native().accel_synchronize()
    else:
        native().accel_synchronize()

    keep = []

    def fresh_keep(i):
        a = np.zeros(N, dtype=np.uint8)
        a[:] = 1
        t = upload(a)
        accel_data_delete(a, "x")
        keep.append(a)
        return t

    def fresh_free(i):
        a = np.zeros(N, dtype=np.uint8)
        a[:] = 1
        t = upload(a)
        accel_data_delete(a, "x")
        del a
        return t

    def fresh_untouched(i):
        a = np.zeros(N, dtype=np.uint8)   # calloc: pages not yet present
        t = upload(a)
        accel_data_delete(a, "x")
        keep.append(a)
        return t

    def with_big_temp(i):
        tmp = np.zeros(8 * N, dtype=np.uint8)
        tmp[:] = 2
        a = np.zeros(N, dtype=np.uint8)
        a[:] = 1
        del tmp                            # munmap of a 29 MB neighbour
        t = upload(a)
        accel_data_delete(a, "x")
        keep.append(a)
        return t

    def after_d2h_and_free(i):
        b = np.zeros(8 * N, dtype=np.uint8)
        accel_data_create(b, "b")
        accel_data_update_host(b, "b")      # D2H into b (pinned by the manager: >= 16 MiB)
        accel_data_delete(b, "b")
        del b                               # munmap of memory that was the target of a copy
        a = np.zeros(N, dtype=np.uint8)
        a[:] = 1
        t = upload(a)
        accel_data_delete(a, "x")
        keep.append(a)
        return t

    def after_small_d2h_and_free(i):
        b = np.zeros(N, dtype=np.uint8)
        accel_data_create(b, "b")
        accel_data_update_host(b, "b")      # D2H into b, NOT pinned by the manager (< 16 MiB): runtime's path
        accel_data_delete(b, "b")
        del b
        a = np.zeros(N, dtype=np.uint8)
        a[:] = 1
        t = upload(a)
        accel_data_delete(a, "x")
        keep.append(a)
        return t

    def after_h2d_source_freed(i):
        b = np.zeros(N, dtype=np.uint8)
        b[:] = 3
        upload(b, "b")
        accel_data_delete(b, "b")
        del b                               # the SOURCE of a pageable H2D copy is unmapped
        a = np.zeros(N, dtype=np.uint8)
        a[:] = 1
        t = upload(a)
        accel_data_delete(a, "x")
        keep.append(a)
        return t

    variant("fresh array, kept alive", fresh_keep)
    variant("fresh array, freed afterwards", fresh_free)
    variant("fresh array, pages untouched before the copy", fresh_untouched)
    variant("29 MB temporary freed right before", with_big_temp)
    variant("29 MB D2H target (manager-pinned) freed before", after_d2h_and_free)
    variant("3.7 MB D2H target (runtime path) freed before", after_small_d2h_and_free)
    variant("3.7 MB H2D source freed before", after_h2d_source_freed)
    variant("fresh array, kept alive (again)", fresh_keep)


if __name__ == "__main__":
    main()
