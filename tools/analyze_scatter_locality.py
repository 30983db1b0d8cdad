"""How many global atomics could an LDS-staged accumulator tile save?  (CPU analysis, oracle
pixels.)  For tiles of 1024 consecutive samples x G detectors, count the distinct pixels (= the
flushes an ideal LDS image needs) and compare with the atomics the run-reduction kernels issue:
one per run of equal pixels inside each 64-sample wave, half of that when the A/B detectors of
a focalplane pixel are merged.  Output per det-sample (multiply by nnz for atomic instructions)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from toast_amd import synth

for label, (nside, rate, scan) in {"cfg3 satellite": (1024, 200.0, "sat"), "cfg5g ground": (2048, 200.0, "ground")}.items():
    n_det, n_samp = 64, 200000
    fp, gamma = synth.hex_focalplane(1024, fov_deg=10.0)
    fp = fp[:n_det]
    bore = (synth.satellite_boresight(n_samp, rate, 600.0, 30.0, 3000.0, 65.0) if scan == "sat"
            else synth.ground_scan(n_samp, rate)[0])
    ivl = synth.make_intervals(n_samp, 1, rate)
    idx = np.arange(n_det, dtype=np.int32)
    quats = np.zeros((n_det, n_samp, 4))
    oracle.pointing_detector(fp, bore, idx, quats, ivl, np.zeros(1, np.uint8), 0)
    pix = np.zeros((n_det, n_samp), dtype=np.int64)
    hs = np.zeros(12 * nside * nside // 3072, dtype=np.uint8)
    oracle.pixels_healpix(idx, quats, np.zeros(1, np.uint8), 0, idx, pix, ivl, hs, 3072, nside, True)
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
