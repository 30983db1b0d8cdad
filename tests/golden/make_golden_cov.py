#!/usr/bin/env python3
"""Generate tests/golden/cov_filter.npz by RUNNING THE REFERENCE ITSELF (oracle/_ref, built in
place by oracle/ref_build.sh; the reference's own no-LAPACK configuration of
toast_math_linearalgebra.cpp is used, so only its LAPACK-free entry points are exercised):

* hit map and inverse pixel covariance accumulated the way BuildHitMap / BuildInverseCovariance
  drive `cov_accum_diag_hits` / `cov_accum_diag_invnpp`
  (src/toast/ops/mapmaker_utils/mapmaker_utils.py:178-204, 464-513;
  src/libtoast/src/toast_map_cov.cpp:66-153) on the pointing of the three golden chains;
* `cov_apply_diag` (toast_map_cov.cpp:471-528);
* the ground-filter kernels `legendre_templates`, `bin_proj`, `bin_invcov`, `add_templates`
  (src/libtoast/src/toast_tod_filter.cpp:160-215, 269-355).

    python tests/golden/make_golden_cov.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402

ref = oracle.load_ref()
assert ref is not None, "oracle/_ref missing: run oracle/ref_build.sh where /root/reference exists"

out = {}
for name in ("chain_a", "chain_b", "chain_c"):
    z = np.load(os.path.join(HERE, name + ".npz"))
    pixels, weights, g2l = z["out_pixels"], z["out_weights"], z["out_g2l"]
    nps = int(z["meta_n_pix_submap"])
    n_local = int(g2l.max()) + 1
    dflags, sflags = z["in_det_flags"], z["in_shared_flags"]
    pidx, widx, fidx = z["in_pixel_index"], z["in_weight_index"], z["in_flag_index"]
    det_scale = z["in_det_scale"]
    hits = np.zeros(n_local * nps, dtype=np.int64)
    invcov = np.zeros(n_local * nps * 6, dtype=np.float64)
    for d in range(pidx.size):
        for iv in z["in_intervals"]:
            sl = slice(int(iv["first"]), int(iv["last"]))
            pix = pixels[pidx[d], sl]
            # PixelDistribution.global_pixel_to_submap (src/toast/pixels.py:300-330)
            good = pix >= 0
            local_sm = np.where(good, g2l[np.where(good, pix // nps, 0)], -1).astype(np.int64)
            local_pix = np.where(good, pix % nps, -1).astype(np.int64)
            if dflags.shape[1] == pixels.shape[1]:
                local_pix[(dflags[fidx[d], sl] & 1) != 0] = -1
            local_pix[(sflags[sl] & 1) != 0] = -1
            ref.cov_accum_diag_hits(n_local, nps, 1, local_sm, local_pix, hits, False)
            w = np.ascontiguousarray(weights[widx[d], sl]).reshape(-1)
            ref.cov_accum_diag_invnpp(n_local, nps, 3, local_sm, local_pix, w, float(det_scale[d]), invcov, False)
    out[name + "_hits"] = hits.reshape(n_local, nps, 1)
    out[name + "_invcov"] = invcov.reshape(n_local, nps, 6)

rng = np.random.default_rng(77)
for nnz in (1, 2, 3):
    nsub, subsize = 3, 19
    mat = rng.standard_normal((nsub, subsize, nnz * (nnz + 1) // 2))
    vec = rng.standard_normal((nsub, subsize, nnz))
    res = vec.copy()
    ref.cov_apply_diag(nsub, subsize, nnz, mat.reshape(-1), res.reshape(-1))
    out[f"apply{nnz}_mat"], out[f"apply{nnz}_vec"], out[f"apply{nnz}_out"] = mat, vec, res

# ground-filter kernels: a triangle-wave azimuth like a constant-elevation scan
n = 5003
t = np.arange(n) / 100.0
az = 0.6 + 0.35 * (2.0 * np.abs((t / 17.0) % 1.0 - 0.5))
phase = (az - az.min()) / (az.max() - az.min()) * 2 - 1
x = np.arange(n) / n * 2 - 1
trend = np.zeros((3, n))
ref.legendre_templates(x, trend, 1, 4)
poly = np.zeros((6, n))
ref.legendre_templates(phase, poly, 0, 6)
templates = np.vstack([trend, poly])
signal = rng.standard_normal((2, n)) + 3.0 * poly[2] - 1.5 * trend[0]
good = (rng.random((2, n)) > 0.03).astype(np.uint8)
out.update(gf_x=x, gf_phase=phase, gf_trend=trend, gf_poly=poly, gf_signal=signal, gf_good=good)
nt = templates.shape[0]
for d in range(2):
    proj = np.zeros(nt)
    invc = np.zeros((nt, nt))
    ref.bin_proj(signal[d].copy(), templates, good[d], proj)
    ref.bin_invcov(templates, good[d], invc)
    coeff = np.linalg.inv(invc) @ proj
    fit = np.zeros(n)
    ref.add_templates(fit, templates[3:], coeff[3:])
    out[f"gf_proj{d}"], out[f"gf_invcov{d}"], out[f"gf_coeff{d}"], out[f"gf_fit{d}"] = proj, invc, coeff, fit
np.savez_compressed(os.path.join(HERE, "cov_filter.npz"), **out)
print("cov_filter.npz:", sorted(out))
