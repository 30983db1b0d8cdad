"""How many global atomics could an LDS-staged accumulator tile save?  (Analysis of the pixel
streams; the pixels come from the HIP library, so this runs on the GPU box.)  For tiles of 1024 consecutive samples x G detectors, count the distinct pixels (= the
flushes an ideal LDS image needs) and compare with the atomics the run-reduction kernels issue:
one per run of equal pixels inside each 64-sample wave, half of that when the A/B detectors of
a focalplane pixel are merged.  Output per det-sample (multiply by nnz for atomic instructions)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from toast_amd import capi, synth

D = capi.dev

for label, (nside, rate, scan) in {"cfg3 satellite": (1024, 200.0, "sat"), "cfg5g ground": (2048, 200.0, "ground")}.items():
    n_det, n_samp = 64, 200000
    fp, gamma = synth.hex_focalplane(1024, fov_deg=10.0)
    fp = fp[:n_det]
    bore = (synth.satellite_boresight(n_samp, rate, 600.0, 30.0, 3000.0, 65.0) if scan == "sat"
            else synth.ground_scan(n_samp, rate)[0])
    ivl = synth.make_intervals(n_samp, 1, rate)
    idx = np.arange(n_det, dtype=np.int32)
    d_bore = torch.from_numpy(bore).cuda()
    d_pix = torch.zeros((n_det, n_samp), dtype=torch.int64, device="cuda")
    d_hs = torch.zeros(12 * nside * nside // 3072, dtype=torch.uint8, device="cuda")
    pt = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, 1)
    D.otf_pixels_healpix(pt, idx, d_pix.data_ptr(), n_samp, ivl, d_hs.data_ptr(), d_hs.numel(), 3072)
    torch.cuda.synchronize()
    pix = d_pix.cpu().numpy()
    chunk, stride = 1024, 7
    runs_single = runs_pair = 0
    distinct = {2: 0, 16: 0, 64: 0}
    picks = range(0, n_samp // chunk, stride)
    for c in picks:
        blk = pix[:, c * chunk:(c + 1) * chunk]
        per_det = [int(np.sum(1 + np.count_nonzero(np.diff(blk[d].reshape(-1, 64), axis=1), axis=1))) for d in range(n_det)]
        runs_single += sum(per_det)
        for d in range(0, n_det, 2):
            runs_pair += per_det[d] if np.array_equal(blk[d], blk[d + 1]) else per_det[d] + per_det[d + 1]
        for g in distinct:
            distinct[g] += sum(np.unique(blk[d0:d0 + g]).size for d0 in range(0, n_det, g))
    tot = len(picks) * chunk * n_det
    print(f"{label}: per det-sample: runs per wave {runs_single / tot:.4f}, with pair merge {runs_pair / tot:.4f}; "
          + "; ".join(f"distinct pixels per (1024 x {g} det) tile {distinct[g] / tot:.4f}" for g in distinct))
