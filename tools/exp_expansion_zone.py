#!/usr/bin/env python3
"""EXPERIMENT: do the expansion kernels -- k_pointing_detector (writes quaternions),
k_pixels_healpix (reads quaternions, writes pixels), k_stokes_iqu (reads quaternions, writes weights) -- run faster when
their OUTPUT lies in a block whose chunks come from two HBM zones (a "streamed" block of the arena, csrc/vmm_slab.cpp)
instead of next to their input in the read-mostly slab?  cfg-3 shapes; ms per launch, best of 3."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from toast_amd import capi, synth  # noqa: E402

n_det, n_samp, rate, nside, nps = 1024, 720000, 200.0, 1024, 3072
dev = torch.device("cuda", 0)
D = capi.dev
st = torch.cuda.current_stream().cuda_stream
capi.arena_reserve(int(70.0 * n_det * n_samp))
capi.arena_reserve(int(34.0 * n_det * n_samp) + (4 << 30), streamed=True)      # room for the quaternions (23.6 GB) + pixels
fp, gamma = synth.hex_focalplane(n_det, fov_deg=10.0)
bore = synth.satellite_boresight(n_samp, rate, 600.0, 30.0, 3000.0, 65.0)
ivl = synth.make_intervals(n_samp, 1, rate)
idx = np.arange(n_det, dtype=np.int32)
n_submap = 12 * nside * nside // nps
d_bore = torch.from_numpy(bore).to(dev)
d_sflags = torch.zeros(n_samp, dtype=torch.uint8, device=dev)
d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)


def block(nbytes, streamed):
    return capi.device_malloc(nbytes, -3 if streamed else -1)


def timed(fn):
    fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


nq, npx, nw = 32 * n_det * n_samp, 8 * n_det * n_samp, 24 * n_det * n_samp
q_plain = block(nq, False)
rows = []
for out_streamed in (False, True):
    tag = "two zones" if out_streamed else "plain slab"
    q_out = block(nq, True) if out_streamed else q_plain
    t_pd = timed(lambda: D.pointing_detector(fp, d_bore.data_ptr(), idx, q_out, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, st))
    rows.append(("k_pointing_detector, quaternions in a %s" % tag, t_pd, 32.0))
    if out_streamed:
        capi.device_free(q_out)
    # inputs: the quaternions in the plain slab (where the operators' cached arrays live)
    D.pointing_detector(fp, d_bore.data_ptr(), idx, q_plain, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, st)
    p_out = block(npx, out_streamed)
    t_px = timed(lambda: D.pixels_healpix(idx, q_plain, d_sflags.data_ptr(), n_samp, 1, idx, p_out, n_samp, ivl,
                                          d_hsub.data_ptr(), n_submap, nps, nside, True, st))
    rows.append(("k_pixels_healpix, pixels in a %s" % tag, t_px, 40.0))
    capi.device_free(p_out)
    w_out = block(nw, out_streamed)
    t_sw = timed(lambda: D.stokes_weights_IQU(idx, q_plain, idx, w_out, n_samp, 0, 0, ivl, np.zeros(n_det), gamma,
                                              np.ones(n_det), False, st))
    rows.append(("k_stokes_iqu, weights in a %s" % tag, t_sw, 56.0))
    where = capi.arena_block_zone(w_out, nw)
    capi.device_free(w_out)
    rows.append(("   (weights block: interleaved slab %s, chunks own / other zone %d / %d)" % where, 0.0, 0.0))
tot = float(n_det) * n_samp
for name, ms, bytes_per in rows:
    if ms:
        print("%-64s %7.3f ms  %5.2f TB/s  %.2f of 8 TB/s" % (name, ms, bytes_per * tot / ms / 1e9, bytes_per * tot / ms / 1e9 / 8.0))
    else:
        print(name)
print(capi.alloc_stats())
