#!/usr/bin/env python3
"""Simulated satellite scan -> binned map, with the reference's operator names.

Counterpart of the reference's ``workflows/toast_sim_satellite_simple.py`` restated with its
current API (PixelsHealpix + StokesWeights + BinMap) -- BASELINE.json
configs[0] (4 detectors x 10 min @ 100 Hz, Nside 64) by default, configs[1] with
``--ndet 64 --minutes 60 --nside 512``.  Everything numerical runs on the MI355X.

    python workflows/sim_satellite_simple.py [--ndet 4] [--minutes 10] [--rate 100] [--nside 64]
                                             [--destripe] [--out map.npz]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from toast_amd import ops  # noqa: E402
from toast_amd.data import defaults  # noqa: E402
from toast_amd.sim import create_satellite_data  # noqa: E402
from toast_amd.templates import Offset  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ndet", type=int, default=4)
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--rate", type=float, default=100.0)
    ap.add_argument("--nside", type=int, default=64)
    ap.add_argument("--destripe", action="store_true", help="solve for baseline offsets (MapMaker PCG)")
    ap.add_argument("--full-pointing", action="store_true", help="cache pixels / weights instead of recomputing")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)

    n_samp = int(args.minutes * 60 * args.rate)
    t0 = time.time()
    data = create_satellite_data(n_det=args.ndet, n_samp=n_samp, rate=args.rate, spin_period_s=600.0,
                                 spin_angle_deg=30.0, prec_period_s=3000.0, prec_angle_deg=65.0, net=1.0)
    rng = np.random.default_rng(1)
    for ob in data.obs:
        sig = ob.detdata[defaults.det_data].data
        sig[:] = rng.standard_normal(sig.shape)
        if args.destripe:
            sig += (rng.standard_normal((sig.shape[0], 1)) * 5.0)  # one offset per detector
    det_pointing = ops.PointingDetectorSimple()
    pixels = ops.PixelsHealpix(detector_pointing=det_pointing, nside=args.nside, nest=True)
    weights = ops.StokesWeights(detector_pointing=det_pointing, mode="IQU", hwp_angle=defaults.hwp_angle)
    binner = ops.BinMap(pixel_dist="pixel_dist", pixel_pointing=pixels, stokes_weights=weights,
                        full_pointing=args.full_pointing)
    templates = None
    if args.destripe:
        templates = ops.TemplateMatrix(templates=[Offset(step_time=60.0, noise_model=defaults.noise_model,
                                                         name="baselines")])
    mapper = ops.MapMaker(name="mapmaker", det_data=defaults.det_data, binning=binner, template_matrix=templates,
                          iter_max=50, convergence=1e-12)
    mapper.apply(data)
    dt = time.time() - t0
    hits = data["mapmaker_hits"].data
    m = data["mapmaker_map"].data
    good = hits[:, :, 0] > 0
    print(f"detectors {args.ndet}  samples/det {n_samp}  nside {args.nside}  "
          f"local submaps {data['pixel_dist'].n_local_submap}  hit pixels {np.count_nonzero(good)}  "
          f"total hits {int(hits.sum())}  PCG iterations {len(mapper.history)}  wall {dt:.2f} s")
    print("map rms  I %.6g  Q %.6g  U %.6g" % tuple(np.sqrt(np.mean(m[good] ** 2, axis=0))))
    if args.out:
        np.savez_compressed(args.out, map=m, hits=hits, submaps=data["pixel_dist"].local_submaps,
                            rcond=data["mapmaker_rcond"].data)
    return data


if __name__ == "__main__":
    main()
