#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE ITSELF
(oracle/_ref = hpc4cmb/toast's own hot-path C++ compiled in place from /root/reference by
oracle/ref_build.sh).  Build container only; the fixtures (inputs + the reference's outputs)
are committed, the reference never travels.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
import oracle  # noqa: E402

ref = oracle.load_ref()
assert ref is not None, "oracle/_ref missing: run oracle/ref_build.sh where /root/reference exists"

# ----------------------------------------------------------------------------- HEALPix KAT
# Same point set as the reference's own test (src/toast/tests/healpix.py:29-61): extreme
# angles with eps32/eps64 perturbations plus a regular grid, nside in {1, 256, 16384}.
eps32 = np.finfo(np.float32).eps
eps64 = np.finfo(np.float64).eps
th = [0.0, eps64, eps32, np.radians(90.0) - eps32, np.radians(90.0) - eps64, np.radians(90.0),
      np.radians(90.0) + eps64, np.radians(90.0) + eps32, np.radians(180.0) - eps32,
      np.radians(180.0) - eps64, np.radians(180.0)]
ph = []
for pts in [0.0, 90.0, 180.0, 270.0, 360.0]:
    ph += [np.radians(pts) - eps32, np.radians(pts) - eps64, np.radians(pts), np.radians(pts) + eps64,
           np.radians(pts) + eps32]
ext = np.array([(t, p) for t in th for p in ph])
nreg = 40
reg = np.array([(np.radians(180.0 * (i + 0.5) / nreg), np.radians(360.0 * (j + 0.5) / nreg))
                for i in range(nreg) for j in range(nreg)])
pts = np.concatenate([ext, reg])
theta = np.ascontiguousarray(pts[:, 0])
phi = np.ascontiguousarray(pts[:, 1])
kat = dict(theta=theta, phi=phi)
vec = np.zeros((theta.size, 3))
ref.healpix_ang2vec(theta, phi, vec)
kat["vec"] = vec
for nside in (1, 256, 16384):
    for name, fn in (("nest", ref.healpix_ang2nest), ("ring", ref.healpix_ang2ring)):
        out = np.zeros(theta.size, dtype=np.int64)
        fn(nside, theta, phi, out)
        kat["ang2%s_%d" % (name, nside)] = out
    for name, fn in (("nest", ref.healpix_vec2nest), ("ring", ref.healpix_vec2ring)):
        out = np.zeros(theta.size, dtype=np.int64)
        fn(nside, vec, out)
        kat["vec2%s_%d" % (name, nside)] = out
    rp = kat["ang2ring_%d" % nside]
    out = np.zeros_like(rp)
    ref.healpix_ring2nest(nside, rp, out)
    kat["ring2nest_%d" % nside] = out
    out2 = np.zeros_like(rp)
    ref.healpix_nest2ring(nside, out, out2)
    kat["nest2ring_%d" % nside] = out2
# regression quaternion of src/toast/tests/ops_pixels_healpix.py:35-42 (nside 4096)
q = np.array([[[-0.51308546259679089, 0.81748419984459697, -0.13909683464480427, 0.22161895602152878]]])
iv = np.zeros(1, cases.interval_dtype)
iv["last"] = 1
for nest in (True, False):
    pix = np.zeros((1, 1), np.int64)
    hs = np.zeros(12 * 4096 * 4096 // 3072, np.uint8)
    ref.pixels_healpix(np.zeros(1, np.int32), q, np.zeros(1, np.uint8), 0, np.zeros(1, np.int32), pix, iv, hs,
                       3072, 4096, nest, False)
    kat["regress_quat_%s" % ("nest" if nest else "ring")] = pix
    kat["regress_quat_%s_submap" % ("nest" if nest else "ring")] = np.flatnonzero(hs)
kat["regress_quat"] = q
np.savez_compressed(os.path.join(HERE, "healpix_kat.npz"), **kat)
print("healpix_kat.npz:", len(pts), "points")

# ----------------------------------------------------------------------------- kernel chains
CHAINS = {
    "chain_a": (dict(n_det=3, n_samp=1500, nside=64, n_split=2, gap=3), dict(nest=True, iau=False)),
    "chain_b": (dict(n_det=2, n_samp=1200, nside=512, with_hwp=True, extra_rows=1, seed=11), dict(nest=False, iau=True)),
    "chain_c": (dict(n_det=1, n_samp=2000, nside=1024, with_det_flags=False, random_pointing=True, seed=5),
                dict(nest=True, iau=False)),
}
INPUT_KEYS = ("focalplane", "gamma", "epsilon", "cal", "boresight", "intervals", "quat_index", "pixel_index",
              "weight_index", "data_index", "flag_index", "shared_flags", "det_flags", "hwp", "tod", "det_scale")
for name, (ckw, rkw) in CHAINS.items():
    c = cases.make_case(**ckw)
    out = cases.run_chain(ref, c, tail=(False,), **rkw)
    blob = {"in_" + k: c[k] for k in INPUT_KEYS}
    blob.update({"meta_" + k: np.array(c[k]) for k in ("n_det", "n_samp", "nside", "rows", "n_pix_submap", "n_submap")})
    blob["meta_nest"] = np.array(rkw["nest"])
    blob["meta_iau"] = np.array(rkw["iau"])
    blob.update({"out_" + k: v for k, v in out.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **blob)
    print(name, {k: v.shape for k, v in out.items()})

# ----------------------------------------------------------------------------- offset template + stokes_I
rng = np.random.default_rng(21)
c = cases.make_case(n_det=2, n_samp=900, n_split=3, gap=4, seed=9)
ivl = c["intervals"]
step = 23
n_amp_views = np.array([(v["last"] - v["first"] + step - 1) // step for v in ivl], dtype=np.int64)
amp_offset = 4
n_amp = int(amp_offset + n_amp_views.sum() + 2)
amps = rng.standard_normal(n_amp)
aflags = (rng.random(n_amp) < 0.15).astype(np.uint8)
tod_add = c["tod"].copy()
ref.template_offset_add_to_signal(step, amp_offset, n_amp_views, amps, aflags, 1, tod_add, ivl, False)
proj = {}
for fidx in (-1, 0):
    a = amps.copy()
    ref.template_offset_project_signal(1, c["tod"], fidx, c["det_flags"], 1, step, amp_offset, n_amp_views, a,
                                       aflags, ivl, False)
    proj[fidx] = a
var = rng.random(n_amp)
pre = np.full(n_amp, 7.0)
ref.template_offset_apply_diag_precond(var, amps, aflags, pre, False)
wI = np.zeros((2, 900))
ref.stokes_weights_I(np.arange(2, dtype=np.int32), wI, ivl, c["cal"], False)
np.savez_compressed(os.path.join(HERE, "offset_template.npz"), intervals=ivl, tod=c["tod"], det_flags=c["det_flags"],
                    step=np.array(step), amp_offset=np.array(amp_offset), n_amp_views=n_amp_views, amps=amps,
                    aflags=aflags, var=var, cal=c["cal"], out_add=tod_add, out_proj_noflag=proj[-1],
                    out_proj_flag=proj[0], out_precond=pre, out_stokes_I=wI)
print("offset_template.npz")
