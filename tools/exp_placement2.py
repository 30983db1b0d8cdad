"""Which property of a 5.9 GB timestream buffer decides how fast scan_map / a plain stream run on it?
Ten buffers allocated one after the other and all kept alive; per buffer: address, scan_map time
(read-modify-write of the buffer + reads of pixels / weights) and k_noise_weight time (pure stream)."""
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
d_pixels = torch.empty((n_det, n_samp), dtype=torch.int64, device=dev)
d_weights = torch.empty((n_det, n_samp, 3), dtype=torch.float64, device=dev)
d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
pt = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, 3, gamma=gamma)
D.otf_pixels_healpix(pt, idx, d_pixels.data_ptr(), n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, stream)
D.otf_stokes_weights(pt, idx, d_weights.data_ptr(), n_samp, ivl, stream)
g2l_h, hit = synth.global_to_local(d_hsub.cpu().numpy())
d_g2l = torch.from_numpy(g2l_h).to(dev)
d_zmap = torch.rand((hit.size, nps, nnz), dtype=torch.float64, device=dev)
det_w = np.linspace(0.5, 0.9, n_det)
ones = np.ones(n_det)

def timed(fn, reps=5):
    fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

print("pixels %x  weights %x" % (d_pixels.data_ptr(), d_weights.data_ptr()))
bufs = []
for k in range(10):
    b = torch.zeros(n_det * n_samp, dtype=torch.float64, device=dev)
    bufs.append(b)
for k, b in enumerate(bufs):
    p = b.data_ptr()
    ts = timed(lambda: D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, p, idx, d_pixels.data_ptr(), idx,
                                  d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False, True, False, det_w, stream))
    tn = timed(lambda: D.noise_weight(p, n_samp, idx, ivl, ones, stream))
    print("buffer %d  ptr %x  (GiB offset from first %.2f)  scan %.3f ms  stream %.3f ms" % (k, p, (p - bufs[0].data_ptr()) / 2**30, ts, tn), flush=True)
