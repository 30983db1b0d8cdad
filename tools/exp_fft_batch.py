"""FFT noise weighting: detectors per batch vs time.  With a small batch the [batch, M] work buffer of the three
passes (16 MB per detector at n_fft = 2^21) stays in the 256 MB Infinity Cache between the passes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from toast_amd import fft as hipfft
from toast_amd.accel import ensure_assigned

ensure_assigned()
n_det, n_samp, rate = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 720000, 200.0
x = torch.randn(n_det, n_samp, dtype=torch.float64, device="cuda")
freq = np.linspace(0, rate / 2, 400)
kern = 1.0 / (1.0 + (0.05 / np.maximum(freq, 1e-4)) ** 1.0)
kernels = np.tile(kern, (n_det, 1))
idx = np.arange(n_det, dtype=np.int32)
hipfft.convolve_dev(x.data_ptr(), idx, n_samp, rate, freq, kernels)
torch.cuda.synchronize()
import ctypes as C
from toast_amd import capi
mag_c, ang_c = hipfft.kernel_coefficients(freq, kernels, False)
n_fft = hipfft.fft_length(n_samp)
n_reflect = min((n_fft - n_samp) // 2, n_samp)
apod = hipfft.apodization(n_reflect)
_p = hipfft._p


def call(batch, st):
    capi._check(capi.lib().toast_hip_fft_convolve_dev(
        C.c_void_p(x.data_ptr()), _p(idx), C.c_int64(idx.size), C.c_int64(n_samp), C.c_double(rate),
        _p(freq), C.c_int64(freq.size), _p(mag_c), _p(ang_c), C.c_int64(mag_c.shape[0]),
        C.c_int(0), _p(apod), C.c_int64(apod.size), C.c_int64(batch), C.c_void_p(st)))


for batch in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 128, 512):
    ts = []
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        call(batch, torch.cuda.current_stream().cuda_stream)
        e0.record()
        call(batch, torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"batch {batch:4d}: " + " ".join(f"{t:7.2f}" for t in ts) + " ms", flush=True)
