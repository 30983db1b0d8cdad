#!/usr/bin/env python3
"""Experiment 2 for the 20-30 ms stalls: explicit mmap / munmap (no malloc heuristics in the way)."""
import mmap
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd.accel import (accel_data_create, accel_data_delete, accel_data_reset, accel_data_update_device,  # noqa: E402
                             accel_data_update_host, native)

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
    m = mmap.mmap(-1, n)
    a = np.frombuffer(m, dtype=np.uint8)
    a[:] = 1
    return m, a


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
    # A: H2D from a mapping, unmap it, then a tiny kernel
    out = []
    for i in range(5):
        m, a = mapped(N)
        accel_data_create(a, "a")
        accel_data_update_device(a, "a")
        accel_data_delete(a, "a")
        sync()
        del a
        m.close()
        out.append(tiny_kernel())
    row("A  H2D source unmapped -> tiny kernel", out)
    # B: plain mmap/munmap never seen by the GPU, then tiny kernel
    out = []
    for i in range(5):
        m, a = mapped(N)
        del a
        m.close()
        out.append(tiny_kernel())
    row("B  unrelated mapping unmapped -> tiny kernel", out)
    # C: H2D from a mapping, unmap, map again (same address likely), H2D from the new one
    out = []
    addr = []
    for i in range(5):
        m, a = mapped(N)
        addr.append(a.ctypes.data)
        accel_data_create(a, "a")
        out.append(timed(lambda: accel_data_update_device(a, "a")))
        accel_data_delete(a, "a")
        sync()
        del a
        m.close()
    row("C  H2D from a mapping at a recycled address", out)
    print("   addresses", [hex(x) for x in addr])
    # D: D2H into a mapping, unmap, tiny kernel
    out = []
    for i in range(5):
        m, a = mapped(N)
        accel_data_create(a, "a")
        accel_data_update_host(a, "a")
        accel_data_delete(a, "a")
        sync()
        del a
        m.close()
        out.append(tiny_kernel())
    row("D  D2H target unmapped -> tiny kernel", out)
    # E: like A but the time between the copy and the unmap is long (runtime's deferred unpin done?)
    out = []
    for i in range(5):
        m, a = mapped(N)
        accel_data_create(a, "a")
        accel_data_update_device(a, "a")
        accel_data_delete(a, "a")
        sync()
        time.sleep(0.05)
        del a
        m.close()
        time.sleep(0.05)
        out.append(tiny_kernel())
    row("E  as A with 50 ms pauses around the unmap", out)
    # F: large (64 MB, manager-pinned) H2D source unmapped after accel_delete (hipHostUnregister) -> tiny kernel
    out = []
    for i in range(5):
        m, a = mapped(64 << 20)
        accel_data_create(a, "a")
        accel_data_update_device(a, "a")
        accel_data_delete(a, "a")
        sync()
        del a
        m.close()
        out.append(tiny_kernel())
    row("F  64 MB registered + unregistered source unmapped -> tiny kernel", out)
    # G: copy of 512 KB (below any pinning threshold of the runtime?) source unmapped
    for n in (64 << 10, 512 << 10, 1 << 20, 2 << 20):
        out = []
        for i in range(5):
            m, a = mapped(n)
            accel_data_create(a, "a")
            accel_data_update_device(a, "a")
            accel_data_delete(a, "a")
            sync()
            del a
            m.close()
            out.append(tiny_kernel())
        row(f"G  {n >> 10} KB H2D source unmapped -> tiny kernel", out)


if __name__ == "__main__":
    main()
