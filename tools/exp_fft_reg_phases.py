#!/usr/bin/env python3
"""EXPERIMENT: phase clocks of the register row pass (csrc/fft_reg.hip k_fft_rows_reg).  Needs a library built with
TOAST_HIP_EXTRA_FLAGS=-DTOAST_FFT_REG_CLOCK (python -m toast_amd.build --force): lane 0 of every wave adds the 100 MHz
wall-clock ticks between phase boundaries (every boundary waits for the wave's outstanding memory operations)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from toast_amd import capi  # noqa: E402
from toast_amd import fft as hipfft  # noqa: E402
from toast_amd.noise import AnalyticNoise  # noqa: E402

n_det, n_samp, rate = 1024, 720000, 200.0
dev = torch.device("cuda", 0)
capi.accel_assign_device(1, 0, 0.0, False)
work = torch.empty((n_det, n_samp), dtype=torch.float64, device=dev).normal_(0.0, 1.0)
nse = AnalyticNoise(rate={"d": rate}, fmin={"d": 1.0e-5}, detectors=["d"], fknee={"d": 0.05}, alpha={"d": 1.0}, NET={"d": 50.0e-6})
kfreq, psd = nse.freq("d"), nse.psd("d")
kern = (50.0e-6) ** 2 / np.maximum(psd, 1.0e-3 * (50.0e-6) ** 2)
kern[0] = 0.0
kernels = np.tile(kern, (n_det, 1)) * np.linspace(0.9, 1.1, n_det)[:, None]
idx = np.arange(n_det, dtype=np.int32)
call = lambda: hipfft.convolve_dev(work.data_ptr(), idx, n_samp, rate, kfreq, kernels, stream=0)
call()
lib = capi.real_lib()
ticks = (C.c_ulonglong * 16)()
lib.toast_hip_fft_reg_ticks(ticks, 1)
call()
lib.toast_hip_fft_reg_ticks(ticks, 1)
n_wave = n_det * 128 * 4      # waves per call at n_fft = 2^21: 128 workgroups of 4 per detector
names = ["entry -> row and tables landed", "forward transform", "hand-over + barrier", "bin pairs",
         "barrier + take-back", "inverse transform", "stores"]
tot = 0.0
for k, nm in enumerate(names):
    us = ticks[k] / n_wave / 100.0
    tot += us
    print("  phase %d %-32s %8.2f us per wave" % (k, nm, us))
print("  sum %.2f us per wave" % tot)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
call()
e1.record()
e1.synchronize()
print("  call %.2f ms" % e0.elapsed_time(e1))
