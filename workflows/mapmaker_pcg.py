#!/usr/bin/env python3
"""BASELINE configs[2] at the operator level: N detectors x 1 h @ 200 Hz satellite scan,
Nside 1024, rocFFT noise weighting (NoiseFilter) followed by the full MapMaker PCG with
offset (baseline) templates, everything through the reference's Operator names.

    python workflows/mapmaker_pcg.py [--ndet 1024] [--minutes 60] [--rate 200] [--nside 1024]
                                     [--iter 10] [--step-time 1.0] [--no-filter] [--uncached [--compact | --no-packed]]

Detector-sharded over several GPUs (configs[3]: --ndet is the number of detectors PER RANK):

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
        workflows/mapmaker_pcg.py --ndet 512 --minutes 240

Prints wall time per phase and the PCG iteration rate in det-samples/s.
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
from toast_amd.sim import create_satellite_data  # noqa: E402
from toast_amd.templates import Offset  # noqa: E402


#: timings of the last main() call of this process (bench.py reports them as its "operator_level" entry)
LAST_STATS = {}


class Phase:
    def __init__(self, quiet=False):
        self.t = time.time()
        self.quiet = quiet

    def lap(self, name):
        native().accel_synchronize()
        now = time.time()
        if not self.quiet:
            print(f"  {name:34s} {now - self.t:8.2f} s", flush=True)
        LAST_STATS.setdefault("laps", {})[name] = now - self.t
        self.t = now


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ndet", type=int, default=1024)
    ap.add_argument("--minutes", type=float, default=60.0)
    ap.add_argument("--rate", type=float, default=200.0)
    ap.add_argument("--nside", type=int, default=1024)
    ap.add_argument("--iter", type=int, default=10)
    ap.add_argument("--step-time", type=float, default=1.0, help="baseline length [s]")
    ap.add_argument("--no-filter", action="store_true")
    ap.add_argument("--noise-prior", action="store_true",
                    help="Offset template with the amplitude-domain noise prior (use_noise_prior=True)")
    ap.add_argument("--precond-width", type=int, default=20, help="band width of the prior's preconditioner")
    ap.add_argument("--uncached", action="store_true",
                    help="full_pointing=False (the reference default): no pointing cache, on-the-fly kernels")
    ap.add_argument("--compact", action="store_true",
                    help="with --uncached: keep a 4 B/det-sample int32 pixel cache, weights on the fly")
    ap.add_argument("--no-packed", action="store_true",
                    help="with --uncached: evaluate the pointing in both sweeps of every iteration instead of letting "
                         "the solver keep its packed pointing cache (18-20 B/det-sample, expanded from the boresight in "
                         "batches of detectors) for the duration of the solve")
    ap.add_argument("--profile", action="store_true", help="cProfile of the MapMaker call (top functions by own time)")
    ap.add_argument("--mem-gb", type=float, default=None,
                    help="device memory the arena takes at accel_assign_device (the reference's TOAST_GPU_MEM_GB); default: "
                         "76 B per local detector-sample + 2 GB -- cached pointing, timestreams, the solver's packed cache "
                         "and one timestream-sized temporary")
    args = ap.parse_args(argv)

    n_samp = int(args.minutes * 60 * args.rate)
    mem_gb = args.mem_gb
    if mem_gb is None:
        mem_gb = float(os.environ.get("TOAST_GPU_MEM_GB", 76.0 * args.ndet * n_samp / 2.0 ** 30 + 2.0))
    # one process per GPU: rank r owns detectors [r * ndet, (r + 1) * ndet) of one focalplane
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    comm = None
    if world > 1:
        import torch
        import torch.distributed as dist

        from toast_amd.accel import accel_assign_device
        from toast_amd.data import Comm

        share = os.environ.get("TOAST_BENCH_SHARE_GPU", "0") == "1"   # tests: ranks share the GPUs, gloo
        local = int(os.environ.get("LOCAL_RANK", "0"))
        nloc = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        accel_assign_device(nloc, local, mem_gb, False)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        comm = Comm()
    else:
        # The pool is taken from the driver now (one hipMalloc, first touch enqueued) and is ready long before the host
        # has simulated its inputs; no operator below calls hipMalloc.
        from toast_amd.accel import accel_assign_device

        accel_assign_device(1, 0, mem_gb, False)
    # the timestreams (signal and the solver's timestream-sized temporaries) are swept read + write: their part of the
    # arena is built from chunks of two HBM zones (toast_hip_arena_reserve_streamed)
    from toast_amd import capi

    capi.arena_reserve(int(16.0 * args.ndet * n_samp) + (2 << 30), streamed=True)
    quiet = rank != 0
    LAST_STATS.clear()
    ph = Phase(quiet)
    data = create_satellite_data(comm=comm, n_det=args.ndet, total_det=args.ndet * world, first_det=args.ndet * rank,
                                 n_samp=n_samp, rate=args.rate, spin_period_s=600.0, spin_angle_deg=30.0,
                                 prec_period_s=3000.0, prec_angle_deg=65.0, net=1.0, fknee=0.05)
    ob = data.obs[0]
    sig = ob.detdata[defaults.det_data].data
    for d in range(sig.shape[0]):  # white noise + one random-walk-ish drift per detector
        # (seeded by the detector's index in the whole focalplane: the same problem however it is sharded)
        rng = np.random.default_rng(1 + args.ndet * rank + d)
        sig[d] = rng.standard_normal(n_samp)
        sig[d] += np.repeat(rng.standard_normal((n_samp + 1999) // 2000), 2000)[:n_samp]
        ob.detdata[defaults.det_flags].data[d] = (rng.random(n_samp) < 0.01).astype(np.uint8) * defaults.det_mask_invalid
    ph.lap("simulate (host)")
    if not args.no_filter:
        ops.NoiseFilter(noise_model=defaults.noise_model).apply(data)
        ph.lap("NoiseFilter")
    det_pointing = ops.PointingDetectorSimple()
    pixels = ops.PixelsHealpix(detector_pointing=det_pointing, nside=args.nside, nest=True)
    weights = ops.StokesWeights(detector_pointing=det_pointing, mode="IQU", hwp_angle=defaults.hwp_angle)
    binner = ops.BinMap(pixel_dist="pixel_dist", pixel_pointing=pixels, stokes_weights=weights,
                        full_pointing=not args.uncached, compact_cache=args.compact, packed_cache=not args.no_packed)
    tmatrix = ops.TemplateMatrix(templates=[Offset(step_time=args.step_time, noise_model=defaults.noise_model,
                                                   name="baselines", use_noise_prior=args.noise_prior,
                                                   precond_width=args.precond_width)])
    mapper = ops.MapMaker(name="mapmaker", keep_solver_products=True, det_data=defaults.det_data, binning=binner, template_matrix=tmatrix,
                          iter_min=args.iter, iter_max=args.iter, convergence=1e-30)
    t0 = time.time()
    if args.profile:
        import cProfile
        import pstats

        prof = cProfile.Profile()
        prof.runcall(mapper.apply, data)
        if not quiet:
            pstats.Stats(prof).sort_stats("tottime").print_stats(45)
    else:
        mapper.apply(data)
    native().accel_synchronize()
    total = time.time() - t0
    ph.lap("MapMaker (cov + RHS + PCG + bin)")
    n_it = len(mapper.history)
    nds = args.ndet * world * n_samp
    its = getattr(mapper, "iteration_seconds", None)
    LAST_STATS.update(mapmaker_s=total, iterations=n_it, relative_residual=float(mapper.history[-1]) if n_it else None,
                      phases_s=dict(getattr(mapper, "timing_log", {})),
                      pcg_iteration_ms=1e3 * float(np.median(its)) if its else None,
                      pcg_Gsamp_s=nds / float(np.median(its)) / 1e9 if its else None,
                      detectors=args.ndet * world, samples_per_detector=n_samp, nside=args.nside,
                      lhs_route=list(getattr(mapper, "lhs_route", ())),
                      lhs_pack_bytes=list(getattr(mapper, "lhs_pack_bytes", ())))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
    if quiet:
        if world > 1:
            dist.destroy_process_group()
        return data
    print(f"ranks {world}  detectors {args.ndet * world}  samples/det {n_samp}  nside {args.nside}  amplitudes "
          f"{data['mapmaker_solve_amplitudes']['baselines'].n_global}  PCG iterations {n_it}  "
          f"relative residual {mapper.history[-1]:.3e}")
    if hasattr(mapper, "timing_log"):
        # (host enqueue time per phase: the device is not waited for at a phase boundary unless TOAST_HIP_PHASE_SYNC=1;
        #  "MapMaker total" and the PCG iteration's median are synchronised wall times)
        print("phases (host side, not synchronised):")
        for k, v in mapper.timing_log.items():
            print(f"  {k:34s} {v:8.2f} s")
        it = mapper.timing_log.get("pcg_iterations", None)
        if it:
            print(f"PCG phase / iterations: {1e3 * it / n_it:.3f} ms (includes the start-up LHS and vector set-up)")
        its = getattr(mapper, "iteration_seconds", None)
        if its:
            med = float(np.median(its))
            print(f"PCG iteration (median wall time): {1e3 * med:.3f} ms  = {nds / med / 1e9:.1f} G det-samples/s")
    print(f"left-hand side route {list(getattr(mapper, 'lhs_route', ()))}, packed bytes per det-sample "
          f"{list(getattr(mapper, 'lhs_pack_bytes', ()))}")
    print(f"MapMaker total {total:.2f} s")
    if world > 1:
        dist.destroy_process_group()
    return data


if __name__ == "__main__":
    main()
