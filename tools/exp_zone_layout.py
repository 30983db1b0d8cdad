#!/usr/bin/env python3
"""EXPERIMENT (profiles/r04_a section 4): which arrays of the headline step want to be where?

One plain allocation of --gb; its zone boundaries are found with split probes (1 GB ranges against GB 1); the five
TOD-domain arrays of the cfg-3 step are then placed at chosen offsets -- all in one zone, read-only arrays and written
timestream in different zones, arrays straddling a boundary -- and build_noise_weighted / scan_map / the read + write
stream are timed for every layout (same contents: pixels and weights expanded from the boresight into each place).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from toast_amd import capi, synth  # noqa: E402

GB = 1 << 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=int, default=232)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    capi.accel_assign_device(1, 0, 0.0, False)
    D = capi.dev
    stream = torch.cuda.current_stream().cuda_stream
    n_det, n_samp, rate, nside = 1024, 720000, 200.0, 1024
    nnz, nps = 3, 3072
    n_submap = 12 * nside * nside // nps
    n = args.gb * GB
    base = capi.device_malloc(n, 0)
    capi.probe_stream(base, n)
    tb = lambda nbytes, ms: 2.0 * nbytes / ms / 1e9
    zmap_ = [None if x == 1 else tb(2 * GB, capi.probe_stream_split([base + GB, base + x * GB], GB)) for x in range(args.gb)]
    cls = ["." if v is None else ("F" if v > 5.45 else "s") for v in zmap_]
    print("zones vs GB 1: " + "".join(cls))
    bounds = [i for i in range(2, args.gb) if cls[i] != cls[i - 1] and cls[i - 1] != "."]
    print("class changes at GB", bounds)
    # boundaries = the first two clean changes (a run of at least 8 equal classes on both sides)
    clean = [b for b in bounds if all(cls[k] == cls[b] for k in range(b, min(b + 8, args.gb))) and
             all(cls[k] == cls[b - 1] for k in range(max(b - 8, 2), b))]
    print("clean boundaries at GB", clean)
    if not clean:
        return
    b1 = clean[0]
    b2 = clean[1] if len(clean) > 1 else None

    fp, gamma = synth.hex_focalplane(n_det, fov_deg=10.0)
    bore = synth.satellite_boresight(n_samp, rate, 600.0, 30.0, 3000.0, 65.0)
    ivl = synth.make_intervals(n_samp, 1, rate)
    sflags_h = synth.shared_flags_block(n_samp, 0.01, value=1)
    idx = np.arange(n_det, dtype=np.int32)
    d_bore = torch.from_numpy(bore).to(dev)
    d_sflags = torch.from_numpy(sflags_h).to(dev)
    d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
    nds = n_det * n_samp
    sizes = dict(pixels=nds * 8, weights=nds * 24, tod=nds * 8, tod2=nds * 8, dflags=nds)
    sigma = 50.0e-6 * np.sqrt(rate)
    pt_x = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sflags.data_ptr(),
                             n_shared_flags=n_samp, shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma,
                             cal=np.ones(n_det))

    def view(ptr, dtype, shape):
        class _B:
            pass
        b = _B()
        typestr = {torch.int64: "<i8", torch.float64: "<f8", torch.uint8: "|u1"}[dtype]
        b.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(ptr, False), version=3)
        return torch.as_tensor(b, device=dev)

    def timed(fn, reps=3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    g2l_cache = {}

    def run(name, off, zmap_off=None):
        """off: array -> offset in GB (float) from the base; zmap_off: the map's offset (default: torch's own block)"""
        p = {k: base + int(v * GB) // 4096 * 4096 for k, v in off.items()}
        for k in p:
            assert p[k] + sizes[k] <= base + n, (k, off[k])
        gen = torch.Generator(device=dev)
        gen.manual_seed(1)
        d_tod = view(p["tod"], torch.float64, (n_det, n_samp))
        d_tod2 = view(p["tod2"], torch.float64, (n_det, n_samp))
        d_df = view(p["dflags"], torch.uint8, (n_det, n_samp))
        d_tod.normal_(0.0, sigma, generator=gen)
        d_tod2.normal_(0.0, sigma, generator=gen)
        d_df.zero_()
        D.otf_pixels_healpix(pt_x, idx, p["pixels"], n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, stream)
        D.otf_stokes_weights(pt_x, idx, p["weights"], n_samp, ivl, stream)
        if "g2l" not in g2l_cache:
            g2l_h, hit = synth.global_to_local(d_hsub.to(torch.int32).cpu().numpy())
            g2l_cache["g2l"] = torch.from_numpy(g2l_h).to(dev)
            g2l_cache["n_local"] = int(hit.size)
            g2l_cache["zmap"] = torch.zeros((int(hit.size), nps, nnz), dtype=torch.float64, device=dev)
        d_g2l, d_zmap = g2l_cache["g2l"], g2l_cache["zmap"]
        if zmap_off is not None:
            d_zmap = view(base + int(zmap_off * GB) // 4096 * 4096, torch.float64, (g2l_cache["n_local"], nps, nnz))
            d_zmap.zero_()
        det_scale = np.full(n_det, 1.0 / (sigma * sigma))
        det_w = np.linspace(0.5, 0.9, n_det)
        bnw = lambda: D.build_noise_weighted(d_g2l.data_ptr(), d_zmap.data_ptr(), nps, nnz, idx, p["pixels"], idx,
                                             p["weights"], idx, p["tod"], idx, p["dflags"], n_samp, det_scale, 1, n_samp,
                                             ivl, d_sflags.data_ptr(), n_samp, 1, stream)
        scan = lambda: D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, p["tod2"], idx, p["pixels"],
                                  idx, p["weights"], idx, n_samp, ivl, 1.0, False, True, False, det_w, stream)
        rw = lambda: D.noise_weight(p["tod2"], n_samp, idx, ivl, np.ones(n_det), stream)
        pix = lambda: D.otf_pixels_healpix(pt_x, idx, p["pixels"], n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, stream)
        t_b, t_s, t_r, t_p = timed(bnw), timed(scan), timed(rw), timed(pix, 2)
        print(f"  {name:58s} bnw {t_b:6.3f}  scan {t_s:6.3f}  sum {t_b + t_s:6.3f}  rw-stream {t_r:5.3f} ms  otf_pixels {t_p:5.3f}")

    A = 2.0                   # start of the usable part of zone A (GB 1 is the reference chunk; irrelevant here)
    B = float(b1) + 1.0
    C = float(b2) + 1.0 if b2 else None
    g = 1.0 / GB
    s_pix, s_w, s_t, s_f = sizes["pixels"] * g, sizes["weights"] * g, sizes["tod"] * g, sizes["dflags"] * g
    print(f"zone A from GB 0, zone B from GB {b1}" + (f", zone C from GB {b2}" if b2 else ""))
    seq = lambda start, names: {nm: start + sum(sizes[k] * g + 0.01 for k in names[:i]) for i, nm in enumerate(names)}
    run("all five arrays in zone A", seq(A, ["pixels", "weights", "tod", "tod2", "dflags"]))
    run("all five arrays in zone B", seq(B, ["pixels", "weights", "tod", "tod2", "dflags"]))
    lay = seq(A, ["pixels", "weights"])
    lay.update(seq(B, ["tod", "tod2", "dflags"]))
    run("pixels weights in A | tod tod2 flags in B", lay)
    lay = seq(A, ["pixels", "weights", "tod", "dflags"])
    lay.update(seq(B, ["tod2"]))
    run("pixels weights tod flags in A | tod2 in B", lay)
    lay = seq(A, ["pixels", "tod", "dflags"])
    lay.update(seq(B, ["weights", "tod2"]))
    run("pixels tod flags in A | weights tod2 in B", lay)
    # straddling: the boundary in the middle of the array
    lay = seq(A, ["pixels", "weights", "tod", "dflags"])
    lay["tod2"] = b1 - s_t / 2
    run("tod2 astride A|B, the rest in A", lay)
    lay = seq(B + 4, ["pixels", "weights", "tod", "dflags"])
    lay["tod2"] = b1 - s_t / 2
    run("tod2 astride A|B, the rest in B", lay)
    # where the MAP lives (the target of build_noise_weighted's atomics), streams in zone A
    layA = seq(A, ["pixels", "weights", "tod", "tod2", "dflags"])
    endA = max(layA[k] + sizes[k] * g for k in layA) + 0.1
    run("all arrays in A, map in A right behind them", layA, zmap_off=endA)
    run("all arrays in A, map in B", layA, zmap_off=B + 40.0 if B + 41 < args.gb else B + 1)
    if b2:
        run("all arrays in A, map in C", layA, zmap_off=C + 30.0 if C + 31 < args.gb else C + 1)
    run("all arrays in A, map astride A|B", layA, zmap_off=b1 - 0.05)
    if b2:
        lay = seq(A, ["pixels", "tod", "dflags"])
        lay["weights"] = b1 - s_w / 2
        lay["tod2"] = b2 - s_t / 2
        run("weights astride A|B, tod2 astride B|C, pixels tod flags in A", lay)
        lay = {"pixels": A, "weights": B, "tod": C, "tod2": C + s_t + 0.01, "dflags": C + 2 * s_t + 0.02}
        run("pixels in A | weights in B | tod tod2 flags in C", lay)
        lay = {"pixels": A, "weights": A + s_pix + 0.01, "tod": B, "dflags": B + s_t + 0.01, "tod2": C}
        run("pixels weights in A | tod flags in B | tod2 in C", lay)
        lay = {"pixels": b1 - s_pix / 2, "weights": b2 - s_w / 2, "tod": A, "dflags": A + s_t + 0.01, "tod2": C + s_w / 2 + 1}
        run("pixels astride A|B, weights astride B|C, tod in A, tod2 in C", lay)


if __name__ == "__main__":
    main()
