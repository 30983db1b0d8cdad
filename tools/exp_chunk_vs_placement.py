"""Headline step (build_noise_weighted + scan_map, cfg3) on slow (hipDeviceMallocContiguous) and plain allocations for the
chunk size given by TOAST_HIP_CHUNK (samples per workgroup pass; default 1024): does a longer contiguous run per workgroup
amortise the address translations that make slow allocations slow?  (profiles/r02_d_placement_experiments.txt section 5)"""
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
d_sflags = torch.from_numpy(synth.shared_flags_block(n_samp, 0.01, value=1)).to(dev)
d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
nds = n_det * n_samp
pt = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sflags.data_ptr(), n_shared_flags=n_samp,
                       shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma, cal=np.ones(n_det))


def alloc(nbytes, flags):
    p = C.c_void_p(0)
    assert lib.toast_hip_device_malloc(C.c_size_t(nbytes), C.c_int(flags), C.byref(p)) == 0
    lib.toast_hip_memset_dev(p, C.c_int(0), C.c_size_t(nbytes), C.c_void_p(st))
    return p.value


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


d_zmap = d_g2l = None
print("TOAST_HIP_CHUNK =", os.environ.get("TOAST_HIP_CHUNK", "1024 (default)"))
for kind, flags in (("contiguous", 4), ("plain", 0), ("contiguous", 4), ("plain", 0)):
    pix, wts, tod, tod2, dfl = alloc(nds * 8, flags), alloc(nds * 24, flags), alloc(nds * 8, flags), alloc(nds * 8, flags), alloc(nds, flags)
    D.otf_pixels_healpix(pt, idx, pix, n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, st)
    D.otf_stokes_weights(pt, idx, wts, n_samp, ivl, st)
    if d_g2l is None:
        g2l_h, hit = synth.global_to_local(d_hsub.cpu().numpy())
        d_g2l = torch.from_numpy(g2l_h).to(dev)
        d_zmap = torch.zeros((hit.size, nps, nnz), dtype=torch.float64, device=dev)
    tt = torch.randn(nds, dtype=torch.float64, device=dev)
    lib.toast_hip_copy_dev(C.c_void_p(tod), C.c_void_p(tt.data_ptr()), C.c_size_t(nds * 8), C.c_void_p(st))
    lib.toast_hip_copy_dev(C.c_void_p(tod2), C.c_void_p(tt.data_ptr()), C.c_size_t(nds * 8), C.c_void_p(st))
    del tt
    det_scale, det_w = np.ones(n_det), np.linspace(0.5, 0.9, n_det)
    t_s = timed(lambda: D.noise_weight(tod2, n_samp, idx, ivl, np.ones(n_det), st))
    t_b = timed(lambda: D.build_noise_weighted(d_g2l.data_ptr(), d_zmap.data_ptr(), nps, nnz, idx, pix, idx, wts, idx, tod, idx, dfl,
                                               n_samp, det_scale, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, st))
    t_c = timed(lambda: D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, tod2, idx, pix, idx, wts, idx, n_samp,
                                   ivl, 1.0, False, True, False, det_w, st))
    print("%-10s stream %.3f  bnw %.3f  scan %.3f  step %.3f ms" % (kind, t_s, t_b, t_c, t_b + t_c), flush=True)
    for p in (pix, wts, tod, tod2, dfl):
        lib.toast_hip_device_free(C.c_void_p(p))
