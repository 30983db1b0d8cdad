#!/usr/bin/env python3
"""Experiment: first upload of a large pageable buffer = page-locking + DMA; how much is which."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd.accel import accel_data_create, accel_data_delete, accel_data_update_device, accel_data_update_host, native  # noqa: E402

n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 6 << 30
a = np.ones(n, dtype=np.uint8)
accel_data_create(a, "a")
native().accel_synchronize()
if "--touch" in sys.argv:
    # first touch of the fresh device allocation by a fill kernel instead of by the DMA engine
    from toast_amd.accel import accel_data_reset

    t0 = time.perf_counter()
    accel_data_reset(a, "a")
    native().accel_synchronize()
    print(f"{'fill kernel over the new allocation':36s} {1e3 * (time.perf_counter() - t0):8.1f} ms", flush=True)
for label in ("first upload (page-lock + DMA)", "second upload (DMA)", "third upload (DMA)"):
    t0 = time.perf_counter()
    accel_data_update_device(a, "a")
    native().accel_synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:36s} {1e3 * dt:8.1f} ms  {n / dt / 1e9:6.1f} GB/s", flush=True)
t0 = time.perf_counter()
accel_data_update_host(a, "a")
dt = time.perf_counter() - t0
print(f"{'download (DMA)':36s} {1e3 * dt:8.1f} ms  {n / dt / 1e9:6.1f} GB/s", flush=True)
t0 = time.perf_counter()
accel_data_delete(a, "a")
print(f"{'delete (unlock)':36s} {1e3 * (time.perf_counter() - t0):8.1f} ms", flush=True)
