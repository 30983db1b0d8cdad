"""Is the slow / fast allocation effect (profiles/r01_b section 14) a channel-conflict effect of the
TIME-MAJOR grid (all detectors stream the same time offset of rows that lie 5.76 MB apart)?  scan_map with
the written timestream placed in plain (fast or slow, as they come) and physically contiguous (always
slow) allocations, time-major vs detector-major workgroup order."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from toast_amd import capi, synth

n_det, n_samp, rate, nside, nps, nnz = 1024, 720000, 200.0, 1024, 3072, 3
dev = torch.device("cuda", 0)
D = capi.dev
lib = capi.real_lib()
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
d_weights = torch.empty((n_det, n_samp, 3), dtype=torch.float64, device=dev)
pt = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sflags.data_ptr(), n_shared_flags=n_samp,
                       shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma, cal=np.ones(n_det))
D.otf_pixels_healpix(pt, idx, d_pixels.data_ptr(), n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, st)
D.otf_stokes_weights(pt, idx, d_weights.data_ptr(), n_samp, ivl, st)
g2l_h, hit = synth.global_to_local(d_hsub.cpu().numpy())
d_g2l = torch.from_numpy(g2l_h).to(dev)
d_zmap = torch.randn((hit.size, nps, nnz), dtype=torch.float64, device=dev)
det_w = np.ones(n_det)
ones = np.ones(n_det)
nbytes = n_det * n_samp * 8


def timed(fn, reps=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


bufs = []
for i in range(8):
    flags = 4 if i % 2 else 0
    p = C.c_void_p(0)
    assert lib.toast_hip_device_malloc(C.c_size_t(nbytes), C.c_int(flags), C.byref(p)) == 0
    lib.toast_hip_memset_dev(p, C.c_int(0), C.c_size_t(nbytes), C.c_void_p(st))
    bufs.append((flags, p))
# (the chunk-order stagger needs the experimental toast_hip_set_stagger entry point of commit "stagger experiment";
#  it is not part of the library: result in profiles/r02_d_placement_experiments.txt)
STAGGERS = [(0, 0, 0)]
print("buffer                 stream / scan per stagger (mod, shift, step) " + " ".join(str(x) for x in STAGGERS) + "  | det-major   [ms]")
for flags, p in bufs:
    row_s, row = [], []
    lib.toast_hip_set_tuning(b"det_major", C.c_int(0))
    for sg in STAGGERS:
        if sg != (0, 0, 0):
            lib.toast_hip_set_stagger(*[C.c_int(x) for x in sg])
        row_s.append(timed(lambda: D.noise_weight(p.value, n_samp, idx, ivl, ones, st)))
        row.append(timed(lambda: D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, p.value, idx,
                                            d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False,
                                            True, False, det_w, st)))
    lib.toast_hip_set_tuning(b"det_major", C.c_int(1))
    dm = timed(lambda: D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, p.value, idx,
                                  d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False,
                                  True, False, det_w, st))
    lib.toast_hip_set_tuning(b"det_major", C.c_int(0))
    print("%-10s %#x" % ("contiguous" if flags else "plain", p.value))
    print("   stream " + " ".join("%6.3f" % x for x in row_s))
    print("   scan   " + " ".join("%6.3f" % x for x in row) + "  | %6.3f" % dm, flush=True)
