"""Where the time of the FFT row pass goes: run with the phase-clock library
(tools/build_fft_phase_clock.sh; TOAST_HIP_LIBRARY=toast_amd/build/libtoast_hip_phase.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from toast_amd import capi, fft as hipfft
from toast_amd.accel import ensure_assigned

ensure_assigned()
n_det, n_samp, rate = 1024, 720000, 200.0
x = torch.randn(n_det, n_samp, dtype=torch.float64, device="cuda")
freq = np.linspace(0, rate / 2, 70)
kern = 1.0 / (1.0 + (0.05 / np.maximum(freq, 1e-4)) ** 1.0)
kernels = np.tile(kern, (n_det, 1)) * np.linspace(0.9, 1.1, n_det)[:, None]
idx = np.arange(n_det, dtype=np.int32)
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["r:load", "r:fft", "r:pair", "r:ifft", "r:store", "cf:load", "cf:fft", "cf:tw+st", "cf:drain", "", "ci:ld+tw", "ci:fft", "ci:store", "ci:drain"]
ticks = (C.c_ulonglong * 16)()
for split in (0,):
    hipfft.set_rows_split(split)
    hipfft.convolve_dev(x.data_ptr(), idx, n_samp, rate, freq, kernels)
    capi.lib().toast_hip_fft_phase_ticks(ticks, C.c_int(1))
    hipfft.convolve_dev(x.data_ptr(), idx, n_samp, rate, freq, kernels)
    capi.lib().toast_hip_fft_phase_ticks(ticks, C.c_int(1))
    t = np.array(list(ticks), dtype=np.float64)
    n_wg = 1024 * 512      # row-pair workgroups per call
    tot = t.sum()
    print(f"rows {'split' if split else 'pair'}: ticks per workgroup (10 ns), share")
    for i, v in enumerate(t):
        if v > 0:
            print(f"   phase {i} {names[i] if i < len(names) else '':8s} {v / n_wg:9.1f}  {100 * v / tot:5.1f} %")
    print(f"   total {tot / n_wg:9.1f} ticks = {tot / n_wg / 100:.2f} us per workgroup")
