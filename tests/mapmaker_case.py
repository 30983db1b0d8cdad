"""Seeded inputs of the end-to-end MapMaker fixture (tests/golden/mapmaker_e2e.npz): the SAME observation is built in the
build container by tests/golden/make_golden_mapmaker.py -- which drives the reference's own compiled kernels (oracle/_ref)
through the reference's operator order and its own ``solve()`` -- and on the GPU box by tests/test_gpu_mapmaker_e2e.py,
which hands it to ``toast_amd.ops.MapMaker``.  NumPy + the host-side data model only (no device, no oracle).

Two cases:
  small     configs[0] size: 4 detectors x 60 000 samples @ 100 Hz, Nside 64, 1 s baselines
  cfg3cut   a cut of configs[2]: 8 detectors x 720 000 samples @ 200 Hz, Nside 1024, 1 s baselines
"""
import numpy as np

CASES = {
    "small": dict(n_det=4, n_samp=60000, rate=100.0, nside=64, step_time=1.0, iters=12, seed=601),
    "cfg3cut": dict(n_det=8, n_samp=720000, rate=200.0, nside=1024, step_time=1.0, iters=10, seed=602),
}


def build(case):
    """-> (data, cfg): one satellite observation with sky signal + white noise + per-baseline drifts, 1 % of the samples
    flagged per detector, a flagged stretch of shared flags (toast_amd.sim.create_satellite_data)."""
    from toast_amd.data import defaults
    from toast_amd.sim import create_satellite_data

    cfg = dict(CASES[case])
    n_det, n_samp, rate = cfg["n_det"], cfg["n_samp"], cfg["rate"]
    data = create_satellite_data(comm=None, n_det=n_det, total_det=n_det, first_det=0, n_samp=n_samp, rate=rate,
                                 spin_period_s=600.0, spin_angle_deg=30.0, prec_period_s=3000.0, prec_angle_deg=65.0,
                                 net=1.0, fknee=0.05, seed=cfg["seed"])
    ob = data.obs[0]
    sig = ob.detdata[defaults.det_data].data
    bore = ob.shared[defaults.boresight_radec].data
    # a smooth "sky": a low-order function of the boresight direction (the same for every detector up to its offset on
    # the focalplane is not needed: the fixture pins arithmetic, not astrophysics)
    z = 1.0 - 2.0 * (bore[:, 0] ** 2 + bore[:, 1] ** 2)
    x = 2.0 * (bore[:, 0] * bore[:, 2] + bore[:, 1] * bore[:, 3])
    sky = 3.0 * z + 2.0 * x * z
    step = int(np.rint(cfg["step_time"] * rate))
    for d in range(n_det):
        rng = np.random.default_rng(cfg["seed"] * 1000 + d)
        sig[d] = sky * (1.0 + 0.01 * d) + rng.standard_normal(n_samp)
        # baseline drifts: a random walk sampled once per two baselines
        walk = np.cumsum(rng.standard_normal((n_samp + 2 * step - 1) // (2 * step))) * 0.5
        sig[d] += np.repeat(walk, 2 * step)[:n_samp]
    return data, cfg
