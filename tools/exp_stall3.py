#!/usr/bin/env python3
"""Experiment 3 for the 20-30 ms stalls: MAP_PRIVATE anonymous mappings (what malloc uses), explicit munmap."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd.accel import (accel_data_create, accel_data_delete, accel_data_reset, accel_data_update_device,  # noqa: E402
                             accel_data_update_host, native)

libc = ctypes.CDLL(None, use_errno=True)
libc.mmap.restype = ctypes.c_void_p
libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
N = 3_686_400


def sync():
    native().accel_synchronize()


small = np.zeros(4096, dtype=np.uint8)


def tiny_kernel():
    sync()
    t0 = time.perf_counter()
    accel_data_reset(small, "small")
    sync()
    return 1e3 * (time.perf_counter() - t0)


def mapped(n):
    p = libc.mmap(None, n, 3, 0x22, -1, 0)   # PROT_READ|WRITE, MAP_PRIVATE|MAP_ANONYMOUS
    a = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p))
    a[:] = 1
    return p, a


def timed(fn):
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    return 1e3 * (time.perf_counter() - t0)


def row(label, vals):
    print(f"{label:66s} " + " ".join(f"{x:7.2f}" for x in vals), flush=True)


def main():
    accel_data_create(small, "small")
    tiny_kernel()
    for n in (N, 8 * N):
        out, out2 = [], []
        for i in range(5):
            p, a = mapped(n)
            accel_data_create(a, "a")
            out2.append(timed(lambda: accel_data_update_device(a, "a")))
            accel_data_delete(a, "a")
            sync()
            del a
            libc.munmap(p, n)
            out.append(tiny_kernel())
        row(f"A  {n >> 10} KB private H2D source: the copy", out2)
        row(f"A  {n >> 10} KB private H2D source unmapped -> tiny kernel", out)
    out = []
    for i in range(5):
        p, a = mapped(N)
        del a
        libc.munmap(p, N)
        out.append(tiny_kernel())
    row("B  unrelated private mapping unmapped -> tiny kernel", out)
    out, out2 = [], []
    for i in range(5):
        p, a = mapped(N)
        accel_data_create(a, "a")
        out2.append(timed(lambda: accel_data_update_host(a, "a")))
        accel_data_delete(a, "a")
        sync()
        del a
        libc.munmap(p, N)
        out.append(tiny_kernel())
    row("D  private D2H target: the copy", out2)
    row("D  private D2H target unmapped -> tiny kernel", out)
    # H: malloc'ed memory (numpy), sizes straddling the dynamic mmap threshold
    for n in (N, 8 * N):
        out = []
        for i in range(6):
            a = np.empty(n, dtype=np.uint8)
            a[:] = 1
            accel_data_create(a, "a")
            t = timed(lambda: accel_data_update_device(a, "a"))
            accel_data_delete(a, "a")
            sync()
            del a
            out.append(t)
            out.append(tiny_kernel())
        row(f"H  numpy {n >> 10} KB: (copy, tiny kernel after free) x 6", out)
    # I: brk heap growth alone: many small mallocs then a copy
    out = []
    for i in range(6):
        junk = [np.empty(100_000, dtype=np.uint8) for _ in range(300)]   # 30 MB from the heap
        for j in junk:
            j[:] = 1
        a = np.empty(N, dtype=np.uint8)
        a[:] = 1
        accel_data_create(a, "a")
        out.append(timed(lambda: accel_data_update_device(a, "a")))
        accel_data_delete(a, "a")
        del junk, a
        out.append(tiny_kernel())
    row("I  heap churn (300 x 100 KB) then (copy, tiny kernel after free)", out)


if __name__ == "__main__":
    main()
