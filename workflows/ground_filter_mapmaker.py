#!/usr/bin/env python3
"""The structure of BASELINE configs[4] on one GPU's share of the detectors: constant-elevation
scans (sweeps as intervals, flagged turnarounds), a ground-synchronous signal injected into the
timestreams, `GroundFilter` (ground-template subtraction), then `MapMaker` at Nside 2048 with
baseline offsets, everything through the reference's Operator names.  The atmosphere simulation of
configs[4] is upstream of this path and is not part of it.

    python workflows/ground_filter_mapmaker.py [--ndet 256] [--minutes 60] [--rate 200] [--nside 2048]
                                               [--iter 10] [--split] [--filter-order 5] [--trend-order 5]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd import ops  # noqa: E402
from toast_amd.accel import native  # noqa: E402
from toast_amd.data import defaults  # noqa: E402
from toast_amd.sim import create_ground_data  # noqa: E402
from toast_amd.templates import Offset  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ndet", type=int, default=256)
    ap.add_argument("--minutes", type=float, default=60.0)
    ap.add_argument("--rate", type=float, default=200.0)
    ap.add_argument("--nside", type=int, default=2048)
    ap.add_argument("--iter", type=int, default=10)
    ap.add_argument("--step-time", type=float, default=1.0)
    ap.add_argument("--trend-order", type=int, default=5)
    ap.add_argument("--filter-order", type=int, default=5)
    ap.add_argument("--split", action="store_true", help="separate templates for left- and right-going sweeps")
    ap.add_argument("--scheduled", type=int, default=0, metavar="N",
                    help="build the observations with ops.SimGround from a schedule of N constant-elevation scans of "
                         "--minutes each (finite-acceleration turnarounds, one observation per scan) instead of "
                         "the synthetic single-observation generator")
    ap.add_argument("--elnods", action="store_true",
                    help="with --scheduled: an el-nod (+1, -1, 0 degrees) before and after every scan and elevation steps of "
                         "0.05 degrees after every scan pair (ops.SimGround elnod_start / elnod_end / el_mod_step); the "
                         "el-nod samples carry the `irregular` flag bit and stay out of the maps")
    args = ap.parse_args(argv)
    n_samp = int(args.minutes * 60 * args.rate)
    t = time.time()

    def lap(name):
        nonlocal t
        native().accel_synchronize()
        now = time.time()
        print(f"  {name:40s} {now - t:8.2f} s", flush=True)
        t = now

    if args.scheduled > 0:
        from toast_amd.ops.sim_ground import create_ground_data_from_schedule
        from toast_amd.schedule import make_ces_schedule

        schedule = make_ces_schedule(args.scheduled, scan_seconds=args.minutes * 60.0, az_min=40.0, az_max=110.0, el=50.0)
        el_motion = dict(elnod_start=True, elnod_end=True, elnods=[1.0, -1.0, 0.0], scan_rate_el=1.0, scan_accel_el=1.0,
                         el_mod_step=0.05) if args.elnods else {}
        data = create_ground_data_from_schedule(schedule, n_det=args.ndet, rate=args.rate, fov_deg=8.0,
                                                scan_rate_az=1.0, scan_accel_az=1.0, fix_rate_on_sky=False, **el_motion)
        for ob in data.obs:     # the map-maker skips samples with any non-science bit: raise "invalid" too
            ob.shared[defaults.shared_flags].data[ob.shared[defaults.shared_flags].data != 0] |= defaults.shared_mask_invalid
    else:
        data = create_ground_data(n_det=args.ndet, n_samp=n_samp, rate=args.rate, az_min_deg=40.0, az_max_deg=110.0,
                                  scan_rate_deg_s=1.0, fov_deg=8.0)
    data.lazy_host = True
    rng = np.random.default_rng(1)
    for ob in data.obs:
        az = ob.shared[defaults.azimuth].data
        phase = (az - az.min()) / (az.max() - az.min()) * 2 - 1
        ground = 20.0 * (np.sin(3 * phase) + 0.5 * phase ** 2)
        sig = ob.detdata[defaults.det_data].data
        for d in range(sig.shape[0]):
            sig[d] = rng.standard_normal(sig.shape[1]) + ground * (1.0 + 0.1 * rng.standard_normal())
    ob = data.obs[0]
    sig = ob.detdata[defaults.det_data].data
    n_samp = ob.n_local_samples
    lap("simulate (host)")
    good = (ob.shared[defaults.shared_flags].data & 1) == 0
    rms_before = float(np.std(sig[0][good]))
    gf = ops.GroundFilter(trend_order=args.trend_order, filter_order=args.filter_order, split_template=args.split,
                          name="groundfilter")
    gf.apply(data)
    lap("GroundFilter (first call: upload + JIT)")
    rms_after = float(np.std(ob.detdata[defaults.det_data].data[0][good]))
    t = time.time()
    det_pointing = ops.PointingDetectorSimple()
    pixels = ops.PixelsHealpix(detector_pointing=det_pointing, nside=args.nside, nest=True,
                               view=defaults.scanning_interval)
    weights = ops.StokesWeights(detector_pointing=det_pointing, mode="IQU", view=defaults.scanning_interval)
    binner = ops.BinMap(pixel_dist="pixel_dist", pixel_pointing=pixels, stokes_weights=weights, full_pointing=True)
    tmatrix = ops.TemplateMatrix(templates=[Offset(step_time=args.step_time, noise_model=defaults.noise_model,
                                                   name="baselines")], view=defaults.scanning_interval)
    mapper = ops.MapMaker(name="mapmaker", det_data=defaults.det_data, binning=binner, template_matrix=tmatrix,
                          iter_min=args.iter, iter_max=args.iter, convergence=1e-30, keep_solver_products=True)
    mapper.apply(data)
    lap("MapMaker (cov + RHS + PCG + bin)")
    nds = args.ndet * sum(o.n_local_samples for o in data.obs)
    n_views = len(ob.intervals[defaults.scanning_interval])
    print(f"detectors {args.ndet}  samples/det {n_samp}  sweeps {n_views}  nside {args.nside}  templates "
          f"{len(next(iter(gf.coefficients.values())))}  rcond mean {gf.rcondsum / max(gf.ngood + gf.nsingular, 1):.2e}")
    print(f"ground signal: rms of detector 0 over good samples {rms_before:.2f} -> {rms_after:.3f} (white noise rms 1)")
    its = getattr(mapper, "iteration_seconds", None)
    if its:
        med = float(np.median(its))
        print(f"PCG iteration (median wall time): {1e3 * med:.1f} ms  = {nds / med / 1e9:.1f} G det-samples/s")
    hits = data["mapmaker_hits"]
    print(f"hit pixels {int(np.count_nonzero(hits.data))}  relative residual {mapper.history[-1]:.2e}")
    return data


if __name__ == "__main__":
    main()
