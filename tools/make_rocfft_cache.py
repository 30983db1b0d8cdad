"""Populate a rocFFT runtime-compilation cache for the transform lengths of the BASELINE
configurations with the rocFFT pipeline forced (run on an MI355X with ROCFFT_RTC_CACHE_PATH pointing
at the output file).  Nothing is shipped or seeded automatically any more: the default noise
weighting runs on the fused kernels (csrc/fft_fused.hip), which need no run-time compilation; users
of TOAST_HIP_FFT=rocfft who want to skip rocFFT's ~1.9 s first-plan build point
ROCFFT_RTC_CACHE_PATH at the file this script writes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "ROCFFT_RTC_CACHE_PATH" in os.environ, "set ROCFFT_RTC_CACHE_PATH to the output file"
import numpy as np
import torch

from toast_amd import fft as hipfft
from toast_amd.accel import ensure_assigned

ensure_assigned()
hipfft.select(True)   # the rocFFT pipeline for every length
rate = 200.0
freq = np.linspace(0, rate / 2, 64)
kern = 1.0 / (1.0 + (0.05 / np.maximum(freq, 1e-4)))
for n_samp, n_det in ((6000, 4), (60000, 4), (360000, 64), (720000, 64), (720000, 8), (2880000, 16), (1440000, 16)):
    x = torch.randn(n_det, n_samp, dtype=torch.float64, device="cuda")
    t0 = time.time()
    hipfft.convolve_dev(x.data_ptr(), np.arange(n_det, dtype=np.int32), n_samp, rate, freq, np.tile(kern, (n_det, 1)))
    torch.cuda.synchronize()
    print(n_samp, n_det, "n_fft", hipfft.fft_length(n_samp), f"{time.time() - t0:.2f} s", flush=True)
    # the FFTPlanReal1D counterpart (half-complex forward / backward)
    y = np.random.default_rng(0).standard_normal((2, hipfft.fft_length(n_samp) // 4))
    hipfft.r1d_backward(hipfft.r1d_forward(y))
print("cache:", os.environ["ROCFFT_RTC_CACHE_PATH"], os.path.getsize(os.environ["ROCFFT_RTC_CACHE_PATH"]), "bytes")
