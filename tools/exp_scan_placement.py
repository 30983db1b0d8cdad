"""Experiment: is the scan_map bimodality (6.3 vs 7.1 ms) tied to buffer placement?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from toast_amd import capi, synth
D = capi.dev
dev = torch.device("cuda", 0)
n_det, n_samp, rate, nside = 1024, 720000, 200.0, 1024
nps, nnz = 3072, 3
n_submap = 12 * nside * nside // nps
stream = torch.cuda.current_stream().cuda_stream
fp, gamma = synth.hex_focalplane(n_det)
bore = synth.satellite_boresight(n_samp, rate)
ivl = synth.make_intervals(n_samp, 1, rate)
idx = np.arange(n_det, dtype=np.int32)
d_bore = torch.from_numpy(bore).to(dev)
d_quats = torch.empty((n_det, n_samp, 4), dtype=torch.float64, device=dev)
d_pixels = torch.empty((n_det, n_samp), dtype=torch.int64, device=dev)
d_weights = torch.empty((n_det, n_samp, 3), dtype=torch.float64, device=dev)
d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
D.pointing_detector(fp, d_bore.data_ptr(), idx, d_quats.data_ptr(), n_samp, ivl, 0, 0, 0, stream)
D.pixels_healpix(idx, d_quats.data_ptr(), 0, 0, 0, idx, d_pixels.data_ptr(), n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, nside, True, stream)
D.stokes_weights_IQU(idx, d_quats.data_ptr(), idx, d_weights.data_ptr(), n_samp, 0, 0, ivl, np.zeros(n_det), gamma, np.ones(n_det), False, stream)
del d_quats; torch.cuda.empty_cache()
g2l_h, hit = synth.global_to_local(d_hsub.cpu().numpy())
d_g2l = torch.from_numpy(g2l_h).to(dev)
d_zmap = torch.rand((hit.size, nps, nnz), dtype=torch.float64, device=dev)
det_w = np.linspace(0.5, 0.9, n_det)
def t_scan(tod, reps=5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, tod, idx, d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False, True, False, det_w, stream)
    e0.record()
    for _ in range(reps):
        D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, tod, idx, d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False, True, False, det_w, stream)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
print("ptrs pixels %x weights %x" % (d_pixels.data_ptr(), d_weights.data_ptr()))
big = torch.zeros(n_det * n_samp * 3 + (64 << 20) // 8, dtype=torch.float64, device=dev)
base = big.data_ptr()
print("slab %x" % base)
for off in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 6 << 20, 32 << 20, n_det * n_samp * 8, n_det * n_samp * 8 + (2 << 20), n_det * n_samp * 16):
    t = t_scan(base + off)
    print("offset %12d B  scan %.3f ms" % (off, t))
for k in range(4):
    b = torch.zeros(n_det * n_samp + 4096, dtype=torch.float64, device=dev)
    print("separate buf %d ptr %x scan %.3f ms" % (k, b.data_ptr(), t_scan(b.data_ptr())))
    del b
