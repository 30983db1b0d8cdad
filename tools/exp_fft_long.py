#!/usr/bin/env python3
"""EXPERIMENT: the fused FFT call at the configs[3] shard shape (512 x 2 880 000 samples, n_fft 2^23: N1 = N2 = 2048, two
columns = 32 bytes per piece in the column passes) -- ms per call and a checksum.   exp_fft_long.py [n_det] [n_samp]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from toast_amd import capi  # noqa: E402
from toast_amd import fft as hipfft  # noqa: E402
from toast_amd.noise import AnalyticNoise  # noqa: E402

n_det = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_samp = int(sys.argv[2]) if len(sys.argv) > 2 else 2880000
rate = 200.0
dev = torch.device("cuda", 0)
capi.accel_assign_device(1, 0, 0.0, False)
gen = torch.Generator(device=dev)
gen.manual_seed(5)
tod = torch.empty((n_det, n_samp), dtype=torch.float64, device=dev).normal_(0.0, 1.0, generator=gen)
work = tod.clone()
nse = AnalyticNoise(rate={"d": rate}, fmin={"d": 1.0e-5}, detectors=["d"], fknee={"d": 0.05}, alpha={"d": 1.0}, NET={"d": 50.0e-6})
kfreq, psd = nse.freq("d"), nse.psd("d")
net_sq = (50.0e-6) ** 2
kern = net_sq / np.maximum(psd, 1.0e-3 * net_sq)
kern[0] = 0.0
kernels = np.tile(kern, (n_det, 1)) * np.linspace(0.9, 1.1, n_det)[:, None]
idx = np.arange(n_det, dtype=np.int32)
stream = torch.cuda.current_stream().cuda_stream
max_batch = int(os.environ.get("EXP_FFT_BATCH", "0"))     # detectors per work batch (0: the library's default)
call = lambda: hipfft.convolve_dev(work.data_ptr(), idx, n_samp, rate, kfreq, kernels, max_batch=max_batch, stream=stream)
call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(3):
    e0.record()
    for _ in range(2):
        call()
    e1.record()
    e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 2)
work.copy_(tod)
call()
torch.cuda.synchronize()
print("batch %d  %d x %d  n_fft %d  env %s  ms per call: %s   checksum %.17g" % (
    max_batch, n_det, n_samp, hipfft.fft_length(n_samp), {k: v for k, v in os.environ.items() if k.startswith("TOAST_HIP_FFT")},
    " ".join("%.3f" % t for t in ts), float(work.double().sum())))
