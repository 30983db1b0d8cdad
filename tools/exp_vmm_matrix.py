#!/usr/bin/env python3
"""EXPERIMENT: physical chunk x virtual slot matrix of pair rates (toast_hip_exp_vmm_pair_matrix)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toast_amd import capi  # noqa: E402

n_phys, n_slots = int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 72
capi.accel_assign_device(1, 0, 0.0, False)
out = np.zeros((n_phys, n_slots))
rc = capi.real_lib().toast_hip_exp_vmm_pair_matrix(C.c_int(n_phys), C.c_int(n_slots), out.ctypes.data_as(C.c_void_p))
assert rc == 0, capi.real_lib().toast_hip_last_error()
print("rows: physical chunk j (paired with chunk 0 at slot 0); columns: virtual slot v = 1 .. %d;  F > 5.45 TB/s, s below" % (n_slots - 1))
for j in range(1, n_phys):
    print("%3d " % j + "".join("F" if x > 5.45 else "s" for x in out[j, 1:]) + "   min %.2f max %.2f" % (out[j, 1:].min(), out[j, 1:].max()))
