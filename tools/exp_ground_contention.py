"""Ground-scan accumulate: is the A^T kernel limited by same-address atomic contention between
detectors?  Same total work, launched (a) once for all detectors into one zmap, (b) as G
sequential launches over detector groups (fewer concurrent colliders)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from toast_amd import capi, synth

D = capi.dev
n_det, n_samp, rate, nside = 256, 720000, 200.0, 2048
nps, nnz = 3072, 3
n_submap = 12 * nside * nside // nps
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
fp, gamma = synth.hex_focalplane(n_det, fov_deg=10.0)
scan_rate = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
bore, ivl, sfl = synth.ground_scan(n_samp, rate, scan_rate_deg_s=scan_rate)
idx = np.arange(n_det, dtype=np.int32)
d_bore = torch.from_numpy(bore).to(dev)
d_sfl = torch.from_numpy(sfl).to(dev)
d_quats = torch.empty((n_det, n_samp, 4), dtype=torch.float64, device=dev)
d_pix = torch.empty((n_det, n_samp), dtype=torch.int64, device=dev)
d_w = torch.empty((n_det, n_samp, 3), dtype=torch.float64, device=dev)
d_tod = torch.randn((n_det, n_samp), dtype=torch.float64, device=dev)
d_fl = torch.zeros((n_det, n_samp), dtype=torch.uint8, device=dev)
d_hs = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
D.pointing_detector(fp, d_bore.data_ptr(), idx, d_quats.data_ptr(), n_samp, ivl, d_sfl.data_ptr(), n_samp, 1, stream)
D.pixels_healpix(idx, d_quats.data_ptr(), d_sfl.data_ptr(), n_samp, 1, idx, d_pix.data_ptr(), n_samp, ivl,
                 d_hs.data_ptr(), n_submap, nps, nside, True, stream)
D.stokes_weights_IQU(idx, d_quats.data_ptr(), idx, d_w.data_ptr(), n_samp, 0, 0, ivl, np.zeros(n_det), gamma,
                     np.ones(n_det), False, stream)
g2l_h, hit = synth.global_to_local(d_hs.cpu().numpy())
n_local = hit.size
d_g2l = torch.from_numpy(g2l_h).to(dev)
ds = np.ones(n_det)
n_in = int(np.sum(ivl["last"] - ivl["first"]))
pix_h = d_pix[0].cpu().numpy()
p = pix_h[int(ivl[0]["first"]):int(ivl[0]["last"])]
runs = 1 + np.count_nonzero(np.diff(p))
print(f"scan rate {scan_rate} deg/s: intervals {ivl.size}, local submaps {n_local}, in-view samples/det {n_in}, "
      f"mean run length {p.size / runs:.1f}", flush=True)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def launch(sub, zmap):
    D.build_noise_weighted(d_g2l.data_ptr(), zmap.data_ptr(), nps, nnz, sub, d_pix.data_ptr(), sub, d_w.data_ptr(), sub,
                           d_tod.data_ptr(), sub, d_fl.data_ptr(), n_samp, ds[: sub.size], 1, n_samp, ivl,
                           d_sfl.data_ptr(), n_samp, 1, stream)


zmaps = [torch.zeros((n_local, nps, nnz), dtype=torch.float64, device=dev) for _ in range(8)]
alg = 41.0 * n_det * n_in
t = timed(lambda: launch(idx, zmaps[0]))
print(f"one launch, one zmap:                 {t:7.3f} ms  {alg / t / 1e6:7.0f} GB/s")
for G in (2, 4, 8):
    groups = [np.ascontiguousarray(idx[g::G]) for g in range(G)]
    t = timed(lambda: [launch(s, zmaps[0]) for s in groups])
    print(f"{G} sequential launches (interleaved dets), one zmap: {t:7.3f} ms  {alg / t / 1e6:7.0f} GB/s")
    groups = [np.ascontiguousarray(idx[g * (n_det // G):(g + 1) * (n_det // G)]) for g in range(G)]
    t = timed(lambda: [launch(s, zmaps[0]) for s in groups])
    print(f"{G} sequential launches (contiguous dets),  one zmap: {t:7.3f} ms  {alg / t / 1e6:7.0f} GB/s")
# (The XCD-private replica variant of this experiment -- kernel selecting one of eight map copies by
# s_getreg HW_REG_XCC_ID -- needed a temporary kernel argument that is no longer in the tree; its
# result is recorded in profiles/r01_b_tuning_experiments.txt section 6.)
