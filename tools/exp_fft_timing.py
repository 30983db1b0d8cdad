"""Time toast_hip_fft_convolve_dev: first call (plan creation / rocFFT kernel build) vs steady state."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from toast_amd import fft as hipfft
from toast_amd.accel import ensure_assigned

ensure_assigned()
n_det = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_samp = int(os.environ.get("FFT_N_SAMP", "720000"))
rate = 200.0
x = torch.randn(n_det, n_samp, dtype=torch.float64, device="cuda")
freq = np.linspace(0, rate / 2, 70)      # NoiseFilter kernels have ~70 knots
kern = 1.0 / (1.0 + (0.05 / np.maximum(freq, 1e-4)) ** 1.0)
kernels = np.tile(kern, (n_det, 1))
idx = np.arange(n_det, dtype=np.int32)
if len(sys.argv) > 4:
    hipfft.set_points(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.time()
    mag_c, ang_c = hipfft.kernel_coefficients(freq, kernels, False)
    t1 = time.time()
    hipfft.convolve_dev(x.data_ptr(), idx, n_samp, rate, freq, kernels)
    torch.cuda.synchronize()
    t2 = time.time()
    print(f"call {it}: coefficients (host) {1e3*(t1-t0):8.1f} ms   convolve_dev total {1e3*(t2-t1):8.1f} ms "
          f"= {n_det*n_samp/(t2-t1)/1e9:6.2f} G samples/s", flush=True)
