"""Row pass of the fused FFT: pair tile (64 KB) vs split rows (32 KB), kernel time of one convolve call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from toast_amd import capi, fft as hipfft
from toast_amd.accel import ensure_assigned

ensure_assigned()
n_det, n_samp, rate = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 720000, 200.0
g = torch.Generator(device="cuda").manual_seed(1)
x0 = torch.randn(n_det, n_samp, dtype=torch.float64, device="cuda", generator=g)
freq = np.linspace(0, rate / 2, 70)
kern = 1.0 / (1.0 + (0.05 / np.maximum(freq, 1e-4)) ** 1.0)
kernels = np.tile(kern, (n_det, 1)) * np.linspace(0.9, 1.1, n_det)[:, None]
idx = np.arange(n_det, dtype=np.int32)
mag_c, ang_c = hipfft.kernel_coefficients(freq, kernels, False)
n_fft = hipfft.fft_length(n_samp)
apod = hipfft.apodization(min((n_fft - n_samp) // 2, n_samp))
_p = hipfft._p


def call(x):
    capi._check(capi.lib().toast_hip_fft_convolve_dev(
        C.c_void_p(x.data_ptr()), _p(idx), C.c_int64(idx.size), C.c_int64(n_samp), C.c_double(rate),
        _p(freq), C.c_int64(freq.size), _p(mag_c), _p(ang_c), C.c_int64(mag_c.shape[0]),
        C.c_int(0), _p(apod), C.c_int64(apod.size), C.c_int64(0), C.c_void_p(torch.cuda.current_stream().cuda_stream)))


res = {}
for split in (0, 1):
    hipfft.set_rows_split(split)
    x = x0.clone()
    call(x)
    torch.cuda.synchronize()
    res[split] = x.clone()
    ts = []
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(x)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"rows {'split' if split else 'pair '}: " + " ".join(f"{t:7.2f}" for t in ts) + " ms", flush=True)
d = (res[0] - res[1]).abs().max().item() / res[0].abs().max().item()
print(f"max |pair - split| / max |pair| = {d:.3e}")
