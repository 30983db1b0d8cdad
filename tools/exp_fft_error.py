"""Actual error level of the two FFT noise-weighting implementations against the NumPy restatement (per-detector
NoiseFilter kernels, max |difference| / max |reference|): DESIGN.md section 6."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from toast_amd import capi, fft as pf
from oracle import fft_oracle as fo
capi.accel_assign_device(1,0,1.0,False)
rng=np.random.default_rng(1)
rate=200.0
freq=np.concatenate([[0.0], np.geomspace(1e-5, rate/2, 70)])
for n_samp in (50001, 720000, 2880000):
    n_det=2
    kernels=[]
    for d in range(n_det):
        net=1.0+0.1*d; fknee=0.05*(d+1)
        psd=net**2*(freq+fknee)/np.maximum(freq+1e-5,1e-12)
        kernels.append(fo.noise_filter_kernel(psd, net))
    kernels=np.array(kernels)
    x=rng.standard_normal((n_det,n_samp)).cumsum(axis=1)*0.01+rng.standard_normal((n_det,n_samp))
    want=x.copy(); fo.convolve(want, rate, kernel_freq=freq, kernels=kernels)
    got=x.copy(); pf.convolve_buffer(got, np.arange(n_det,dtype=np.int32), rate, freq, kernels)
    pf.select(True); lib=x.copy(); pf.convolve_buffer(lib, np.arange(n_det,dtype=np.int32), rate, freq, kernels); pf.select(False)
    s=np.max(np.abs(want))
    print(n_samp, "fused vs numpy %.2e  rocfft vs numpy %.2e  fused vs rocfft %.2e"%(np.max(np.abs(got-want))/s, np.max(np.abs(lib-want))/s, np.max(np.abs(got-lib))/s))
