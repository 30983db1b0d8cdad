"""Run the kernels whose bound is in question a few times each at cfg-3 shapes, for rocprofv3
(--kernel-trace --stats or --pmc ...): k_pixels_healpix, the from-boresight expansions, the
on-the-fly accumulate / scan, and the fused FFT passes.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -d out -o name -- python3 tools/prof_hot_kernels.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from toast_amd import capi, synth
from toast_amd import fft as hipfft

n_det = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_samp, rate, nside, nps, nnz = 720000, 200.0, 1024, 3072, 3
dev = torch.device("cuda", 0)
D = capi.dev
st = torch.cuda.current_stream().cuda_stream
fp, gamma = synth.hex_focalplane(n_det, fov_deg=10.0)
bore = synth.satellite_boresight(n_samp, rate, 600.0, 30.0, 3000.0, 65.0)
ivl = synth.make_intervals(n_samp, 1, rate)
idx = np.arange(n_det, dtype=np.int32)
n_submap = 12 * nside * nside // nps
d_bore = torch.from_numpy(bore).to(dev)
d_sflags = torch.zeros(n_samp, dtype=torch.uint8, device=dev)
d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
d_pixels = torch.empty((n_det, n_samp), dtype=torch.int64, device=dev)
d_tod = torch.randn((n_det, n_samp), dtype=torch.float64, device=dev)
d_dflags = torch.zeros((n_det, n_samp), dtype=torch.uint8, device=dev)
d_quats = torch.empty((n_det, n_samp, 4), dtype=torch.float64, device=dev)
D.pointing_detector(fp, d_bore.data_ptr(), idx, d_quats.data_ptr(), n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, st)
for _ in range(3):
    D.pixels_healpix(idx, d_quats.data_ptr(), d_sflags.data_ptr(), n_samp, 1, idx, d_pixels.data_ptr(), n_samp, ivl,
                     d_hsub.data_ptr(), n_submap, nps, nside, True, st)
del d_quats
pt = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sflags.data_ptr(),
                       n_shared_flags=n_samp, shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma,
                       cal=np.ones(n_det))
for _ in range(3):
    D.otf_pixels_healpix(pt, idx, d_pixels.data_ptr(), n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, st)
g2l_h, hit = synth.global_to_local(d_hsub.cpu().numpy())
d_g2l = torch.from_numpy(g2l_h).to(dev)
d_zmap = torch.zeros((hit.size, nps, nnz), dtype=torch.float64, device=dev)
det_scale = np.ones(n_det)
for _ in range(3):
    D.otf_build_noise_weighted(pt, d_g2l.data_ptr(), d_zmap.data_ptr(), nps, idx, d_tod.data_ptr(), idx,
                               d_dflags.data_ptr(), n_samp, det_scale, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, st)
    D.otf_scan_map(pt, d_g2l.data_ptr(), d_zmap.data_ptr(), nps, d_tod.data_ptr(), idx, n_samp, ivl, 1.0, False, True,
                   det_scale, st)
freq = np.concatenate([[0.0], np.geomspace(1e-5, rate / 2, 70)])
kern = 1.0 / (1.0 + (0.05 / np.maximum(freq, 1e-5)))
kern[0] = 0
kernels = np.tile(kern, (n_det, 1))
for _ in range(3):
    hipfft.convolve_dev(d_tod.data_ptr(), idx, n_samp, rate, freq, kernels, stream=st)
torch.cuda.synchronize()
print("done")
