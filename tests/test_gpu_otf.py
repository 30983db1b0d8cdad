"""GPU: pointing-on-the-fly kernels (toast_amd/csrc/otf_kernels.hip) against
(i) the CPU oracle's operator chain pointing_detector -> pixels_healpix -> stokes_weights ->
build_noise_weighted / scan_map / noise_weight (the reference's full_pointing=False sequence,
src/toast/ops/mapmaker_binning.py:265-271) and (ii) our cached-pointing device kernels, with
which the gather side must agree bit for bit (same device functions, no atomics)."""
import numpy as np
import pytest

from cases import make_case, run_chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from toast_amd import capi

    assert torch.cuda.is_available()
    capi.lib()
    return torch, capi


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _descriptor(torch, capi, c, nest, iau, nnz, shared_mask, hold, compact=None, hwp_table=False):
    bore = _dev(torch, c["boresight"])
    sfl = _dev(torch, c["shared_flags"])
    hwp = _dev(torch, c["hwp"])
    hold += [bore, sfl, hwp]
    extra = {}
    if hwp_table and c["hwp"].size == c["n_samp"]:
        tab = torch.empty((c["n_samp"], 2), dtype=torch.float64, device="cuda")
        capi.dev.hwp_table(hwp.data_ptr(), c["n_samp"], tab.data_ptr())
        torch.cuda.synchronize()
        want = np.stack([np.cos(4.0 * c["hwp"]), np.sin(4.0 * c["hwp"])], axis=1)
        assert np.max(np.abs(tab.cpu().numpy() - want)) < 1e-15
        hold.append(tab)
        extra["d_hwp_table"] = tab.data_ptr()
    if compact is not None:
        extra = dict(d_compact_pixels=compact.data_ptr(), compact_index=c["pixel_index"])
    return capi.otf_pointing(bore.data_ptr(), c["focalplane"], c["nside"], nest, nnz,
                             d_shared_flags=sfl.data_ptr(), n_shared_flags=c["shared_flags"].size,
                             shared_flag_mask=shared_mask, d_hwp=hwp.data_ptr(), n_hwp=c["hwp"].size,
                             epsilon=c["epsilon"], gamma=c["gamma"], cal=c["cal"], IAU=iau, **extra), sfl


def _compact(torch, capi, c, g2l, pixels_dev, n_local):
    """int32 local-index cache from int64 pixels (toast_hip_compact_pixels_dev), checked against
    the definition."""
    cp = torch.full((c["rows"], c["n_samp"]), -9, dtype=torch.int32, device="cuda")
    capi.dev.compact_pixels(g2l.data_ptr(), c["n_pix_submap"], n_local, c["pixel_index"], pixels_dev.data_ptr(),
                            c["pixel_index"], cp.data_ptr(), c["n_samp"], c["intervals"])
    torch.cuda.synchronize()
    pix = pixels_dev.cpu().numpy()
    g = g2l.cpu().numpy()
    want = np.full(pix.shape, -9, dtype=np.int64)
    for iv in c["intervals"]:
        sl = slice(int(iv["first"]), int(iv["last"]))
        for r in c["pixel_index"]:
            p = pix[r, sl]
            ok = p >= 0
            sm = np.where(ok, p // c["n_pix_submap"], 0)
            want[r, sl] = np.where(ok, g[sm] * c["n_pix_submap"] + p % c["n_pix_submap"], -1)
    assert np.array_equal(cp.cpu().numpy(), want)
    return cp


CASES = {
    "plain": dict(n_det=4, n_samp=5000, nside=64),
    "hwp_split": dict(n_det=5, n_samp=7001, nside=256, with_hwp=True, n_split=4, gap=13),
    "index_rows": dict(n_det=3, n_samp=3000, nside=32, extra_rows=2, with_hwp=True),
    "no_flags": dict(n_det=2, n_samp=2500, nside=1024, with_shared_flags=False, with_det_flags=False),
    "random": dict(n_det=3, n_samp=4000, nside=128, random_pointing=True),
    "unpaired": dict(n_det=6, n_samp=4000, nside=128, fp_roll=1),
    # 64-bit pixel arithmetic (Nside > 8192)
    "nside16384": dict(n_det=2, n_samp=3000, nside=16384, random_pointing=True),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("nest", [True, False])
def test_otf_matches_operator_chain(env, oracle, name, nest):
    torch, capi = env
    D = capi.dev
    c = make_case(**CASES[name])
    iau = name == "hwp_split"
    want = run_chain(oracle, c, nest=nest, iau=iau, shared_mask=1, det_mask=1)
    n_samp, n_det = c["n_samp"], c["n_det"]
    hold = []
    pt, sfl = _descriptor(torch, capi, c, nest, iau, 3, 1, hold)
    g2l = _dev(torch, want["g2l"])
    tod = _dev(torch, c["tod"])
    dfl = _dev(torch, c["det_flags"])
    n_flag = n_samp if c["det_flags"].shape[-1] == n_samp else 0
    zmap = torch.zeros(want["zmap"].shape, dtype=torch.float64, device="cuda")
    D.otf_build_noise_weighted(pt, g2l.data_ptr(), zmap.data_ptr(), c["n_pix_submap"], c["data_index"],
                               tod.data_ptr(), c["flag_index"], dfl.data_ptr(), n_flag, c["det_scale"], 1, n_samp,
                               c["intervals"], sfl.data_ptr(), c["shared_flags"].size, 1)
    torch.cuda.synchronize()
    z = zmap.cpu().numpy()
    scale = np.max(np.abs(want["zmap"]))
    assert scale > 0
    # tolerance: weights agree with the oracle to 1e-12 relative, the scatter order is free
    assert np.max(np.abs(z - want["zmap"])) < 1e-11 * scale
    # untouched pixels stay exactly zero
    assert np.array_equal(z == 0, want["zmap"] == 0)

    # gather side: scan_map(subtract) + noise_weight fused, map = the oracle's zmap
    zin = _dev(torch, want["zmap"])
    tod2 = _dev(torch, c["tod"])
    D.otf_scan_map(pt, g2l.data_ptr(), zin.data_ptr(), c["n_pix_submap"], tod2.data_ptr(), c["data_index"], n_samp,
                   c["intervals"], 1.0, False, True, det_weights=c["det_scale"])
    torch.cuda.synchronize()
    t = tod2.cpu().numpy()
    tscale = np.max(np.abs(want["tod"]))
    assert np.max(np.abs(t - want["tod"])) < 1e-11 * tscale
    # rows / samples outside the call are untouched
    rows = np.ones(c["rows"], dtype=bool)
    rows[c["data_index"]] = False
    assert np.array_equal(t[rows], c["tod"][rows])

    # (ii) bit-for-bit against the cached-pointing device kernels
    rows_n = c["rows"]
    quats = torch.zeros((rows_n, n_samp, 4), dtype=torch.float64, device="cuda")
    pixels = torch.full((rows_n, n_samp), -7, dtype=torch.int64, device="cuda")
    weights = torch.zeros((rows_n, n_samp, 3), dtype=torch.float64, device="cuda")
    hsub = torch.zeros(c["n_submap"], dtype=torch.uint8, device="cuda")
    bore, hwp = hold[0], hold[2]
    D.pointing_detector(c["focalplane"], bore.data_ptr(), c["quat_index"], quats.data_ptr(), n_samp, c["intervals"],
                        sfl.data_ptr(), c["shared_flags"].size, 1)
    D.pixels_healpix(c["quat_index"], quats.data_ptr(), sfl.data_ptr(), c["shared_flags"].size, 1, c["pixel_index"],
                     pixels.data_ptr(), n_samp, c["intervals"], hsub.data_ptr(), c["n_submap"], c["n_pix_submap"],
                     c["nside"], nest)
    D.stokes_weights_IQU(c["quat_index"], quats.data_ptr(), c["weight_index"], weights.data_ptr(), n_samp,
                         hwp.data_ptr(), c["hwp"].size, c["intervals"], c["epsilon"], c["gamma"], c["cal"], iau)
    tod3 = _dev(torch, c["tod"])
    D.scan_map(np.float64, g2l.data_ptr(), c["n_pix_submap"], zin.data_ptr(), 3, tod3.data_ptr(), c["data_index"],
               pixels.data_ptr(), c["pixel_index"], weights.data_ptr(), c["weight_index"], n_samp, c["intervals"],
               1.0, False, True, False, det_weights=c["det_scale"])
    torch.cuda.synchronize()
    assert np.array_equal(tod3.cpu().numpy(), t)
    zc = torch.zeros_like(zmap)
    D.build_noise_weighted(g2l.data_ptr(), zc.data_ptr(), c["n_pix_submap"], 3, c["pixel_index"], pixels.data_ptr(),
                           c["weight_index"], weights.data_ptr(), c["data_index"], tod.data_ptr(), c["flag_index"],
                           dfl.data_ptr(), n_flag, c["det_scale"], 1, n_samp, c["intervals"], sfl.data_ptr(),
                           c["shared_flags"].size, 1)
    torch.cuda.synchronize()
    assert np.max(np.abs(zc.cpu().numpy() - z)) < 1e-13 * scale

    # quaternion-free expansion: same pixels / hit submaps / weights as the three-kernel chain
    pix_q = torch.full((rows_n, n_samp), -7, dtype=torch.int64, device="cuda")
    hs_q = torch.zeros(c["n_submap"], dtype=torch.uint8, device="cuda")
    w_q = torch.zeros((rows_n, n_samp, 3), dtype=torch.float64, device="cuda")
    pth, _ = _descriptor(torch, capi, c, nest, iau, 3, 1, hold, hwp_table=True)
    D.otf_pixels_healpix(pth, c["pixel_index"], pix_q.data_ptr(), n_samp, c["intervals"], hs_q.data_ptr(),
                         c["n_submap"], c["n_pix_submap"])
    D.otf_stokes_weights(pth, c["weight_index"], w_q.data_ptr(), n_samp, c["intervals"])
    torch.cuda.synchronize()
    assert np.array_equal(pix_q.cpu().numpy(), want["pixels"])          # == the CPU oracle, bit for bit
    assert np.array_equal(hs_q.cpu().numpy(), want["hsub"])
    assert bool(torch.equal(w_q, weights))                               # == k_stokes_iqu, bit for bit

    # (iii) compact mode: int32 local pixel cache + weights on the fly
    cp = _compact(torch, capi, c, g2l, pixels, want["zmap"].shape[0])
    # ... with the per-observation HWP table (identical values: same bits as without)
    ptc, _ = _descriptor(torch, capi, c, nest, iau, 3, 1, hold, compact=cp, hwp_table=True)
    tod4 = _dev(torch, c["tod"])
    D.otf_scan_map(ptc, g2l.data_ptr(), zin.data_ptr(), c["n_pix_submap"], tod4.data_ptr(), c["data_index"], n_samp,
                   c["intervals"], 1.0, False, True, det_weights=c["det_scale"])
    z4 = torch.zeros_like(zmap)
    D.otf_build_noise_weighted(ptc, g2l.data_ptr(), z4.data_ptr(), c["n_pix_submap"], c["data_index"],
                               tod.data_ptr(), c["flag_index"], dfl.data_ptr(), n_flag, c["det_scale"], 1, n_samp,
                               c["intervals"], sfl.data_ptr(), c["shared_flags"].size, 1)
    torch.cuda.synchronize()
    assert np.array_equal(tod4.cpu().numpy(), t)
    assert np.max(np.abs(z4.cpu().numpy() - z)) < 1e-13 * scale


def test_otf_intensity_only(env, oracle):
    torch, capi = env
    D = capi.dev
    c = make_case(n_det=3, n_samp=4000, nside=128)
    n_samp = c["n_samp"]
    want = run_chain(oracle, c, nest=True)
    w1 = np.zeros((c["rows"], n_samp), dtype=np.float64)
    oracle.stokes_weights_I(c["weight_index"], w1, c["intervals"], c["cal"])
    z1 = np.zeros(want["zmap"].shape[:2] + (1,), dtype=np.float64)
    oracle.build_noise_weighted(want["g2l"], z1, c["pixel_index"], want["pixels"], c["weight_index"],
                                w1.reshape(c["rows"], n_samp, 1), c["data_index"], c["tod"], c["flag_index"],
                                c["det_flags"], c["det_scale"], 1, c["intervals"], c["shared_flags"], 1)
    hold = []
    pt, sfl = _descriptor(torch, capi, c, True, False, 1, 1, hold)
    g2l = _dev(torch, want["g2l"])
    tod = _dev(torch, c["tod"])
    dfl = _dev(torch, c["det_flags"])
    zmap = torch.zeros(z1.shape, dtype=torch.float64, device="cuda")
    D.otf_build_noise_weighted(pt, g2l.data_ptr(), zmap.data_ptr(), c["n_pix_submap"], c["data_index"],
                               tod.data_ptr(), c["flag_index"], dfl.data_ptr(), n_samp, c["det_scale"], 1, n_samp,
                               c["intervals"], sfl.data_ptr(), c["shared_flags"].size, 1)
    torch.cuda.synchronize()
    assert np.max(np.abs(zmap.cpu().numpy() - z1)) < 1e-12 * np.max(np.abs(z1))
    tod_o = c["tod"].copy()
    oracle.scan_map(want["g2l"], c["n_pix_submap"], z1, tod_o, c["data_index"], want["pixels"], c["pixel_index"],
                    w1.reshape(c["rows"], n_samp, 1), c["weight_index"], c["intervals"], 1.0, True, False, False)
    zin = _dev(torch, z1)
    tod2 = _dev(torch, c["tod"])
    D.otf_scan_map(pt, g2l.data_ptr(), zin.data_ptr(), c["n_pix_submap"], tod2.data_ptr(), c["data_index"], n_samp,
                   c["intervals"], 1.0, True, False)
    torch.cuda.synchronize()
    got = tod2.cpu().numpy()
    assert np.max(np.abs(got - tod_o)) < 1e-12 * np.max(np.abs(tod_o))


@pytest.mark.parametrize("with_hwp", [False, True])
def test_otf_offset_lhs_matches_cached(env, oracle, with_hwp):
    """The on-the-fly fused PCG halves equal the cached-pointing fused kernels (which are tested
    against the operator sequence in test_gpu_ops / test_gpu_parity)."""
    torch, capi = env
    D = capi.dev
    c = make_case(n_det=4, n_samp=6000, nside=64, with_hwp=with_hwp, n_split=3, gap=7)
    n_samp, n_det = c["n_samp"], c["n_det"]
    want = run_chain(oracle, c, nest=True)
    hold = []
    pt, sfl = _descriptor(torch, capi, c, True, False, 3, 1, hold)
    step = 37
    ivl = c["intervals"]
    n_amp_views = np.array([-(-(int(v["last"]) - int(v["first"])) // step) for v in ivl], dtype=np.int64)
    per_det = int(n_amp_views.sum())
    amp_offsets = np.arange(n_det, dtype=np.int64) * per_det
    rng = np.random.default_rng(5)
    amps = rng.standard_normal(n_det * per_det)
    aflags = (rng.random(amps.size) < 0.05).astype(np.uint8)
    d_amps, d_afl = _dev(torch, amps), _dev(torch, aflags)
    g2l = _dev(torch, want["g2l"])
    dfl = _dev(torch, c["det_flags"])
    pix = _dev(torch, want["pixels"])
    wts = _dev(torch, want["weights"])
    # cached pixels / weights of the oracle feed the cached kernels; bit-identical pixels, weights
    # to 1e-12 -> compare with a tolerance
    z_c = torch.zeros(want["zmap"].shape, dtype=torch.float64, device="cuda")
    z_o = torch.zeros_like(z_c)
    D.offset_accumulate(step, amp_offsets, n_amp_views, d_amps.data_ptr(), d_afl.data_ptr(), g2l.data_ptr(),
                        z_c.data_ptr(), c["n_pix_submap"], 3, c["pixel_index"], pix.data_ptr(), c["weight_index"],
                        wts.data_ptr(), c["flag_index"], dfl.data_ptr(), n_samp, c["det_scale"], 1, n_samp, ivl,
                        sfl.data_ptr(), c["shared_flags"].size, 1)
    D.otf_offset_accumulate(pt, step, amp_offsets, n_amp_views, d_amps.data_ptr(), d_afl.data_ptr(), g2l.data_ptr(),
                            z_o.data_ptr(), c["n_pix_submap"], c["flag_index"], dfl.data_ptr(), n_samp,
                            c["det_scale"], 1, n_samp, ivl, sfl.data_ptr(), c["shared_flags"].size, 1)
    torch.cuda.synchronize()
    zc, zo = z_c.cpu().numpy(), z_o.cpu().numpy()
    assert np.max(np.abs(zc)) > 0
    assert np.max(np.abs(zc - zo)) < 1e-11 * np.max(np.abs(zc))
    out_c = torch.zeros(amps.size, dtype=torch.float64, device="cuda")
    out_o = torch.zeros_like(out_c)
    D.offset_scan_project(step, amp_offsets, n_amp_views, d_amps.data_ptr(), out_c.data_ptr(), d_afl.data_ptr(),
                          g2l.data_ptr(), z_c.data_ptr(), c["n_pix_submap"], 3, c["pixel_index"], pix.data_ptr(),
                          c["weight_index"], wts.data_ptr(), c["flag_index"], dfl.data_ptr(), 4, c["det_scale"],
                          n_samp, ivl)
    D.otf_offset_scan_project(pt, step, amp_offsets, n_amp_views, d_amps.data_ptr(), out_o.data_ptr(),
                              d_afl.data_ptr(), g2l.data_ptr(), z_c.data_ptr(), c["n_pix_submap"], c["flag_index"],
                              dfl.data_ptr(), n_samp, 4, c["det_scale"], n_samp, ivl)
    torch.cuda.synchronize()
    oc, oo = out_c.cpu().numpy(), out_o.cpu().numpy()
    assert np.max(np.abs(oc)) > 0
    assert np.max(np.abs(oc - oo)) < 1e-11 * np.max(np.abs(oc))
    assert np.array_equal(oc == 0, oo == 0)
    # compact mode
    cp = _compact(torch, capi, c, g2l, pix, want["zmap"].shape[0])
    ptc, _ = _descriptor(torch, capi, c, True, False, 3, 1, hold, compact=cp)
    z_k = torch.zeros_like(z_c)
    out_k = torch.zeros_like(out_c)
    D.otf_offset_accumulate(ptc, step, amp_offsets, n_amp_views, d_amps.data_ptr(), d_afl.data_ptr(), g2l.data_ptr(),
                            z_k.data_ptr(), c["n_pix_submap"], c["flag_index"], dfl.data_ptr(), n_samp,
                            c["det_scale"], 1, n_samp, ivl, sfl.data_ptr(), c["shared_flags"].size, 1)
    D.otf_offset_scan_project(ptc, step, amp_offsets, n_amp_views, d_amps.data_ptr(), out_k.data_ptr(),
                              d_afl.data_ptr(), g2l.data_ptr(), z_c.data_ptr(), c["n_pix_submap"], c["flag_index"],
                              dfl.data_ptr(), n_samp, 4, c["det_scale"], n_samp, ivl)
    torch.cuda.synchronize()
    assert np.max(np.abs(z_k.cpu().numpy() - zo)) < 1e-13 * np.max(np.abs(zc))
    assert np.max(np.abs(out_k.cpu().numpy() - oo)) < 1e-13 * np.max(np.abs(oc))


@pytest.mark.parametrize("with_hwp", [False, True])
def test_otf_clean_accumulate_matches_subtract_then_bin(env, oracle, with_hwp):
    """Round 6: zmap += P^T N^-1 (d - M a) with the pointing evaluated in the kernel
    (toast_hip_otf_offset_clean_accumulate_dev, k_otf_accumulate<.., SIG = 2, ..>; also from the compact pixel cache)
    against the steps it stands in for: the offset template added into a zeroed buffer, d - template on the device,
    otf_build_noise_weighted of the result -- the same kernel family, so the map values agree to the rounding of the
    atomic additions and the same pixels are touched; and against the cached-pointing form on the oracle's pointing."""
    torch, capi = env
    D = capi.dev
    c = make_case(n_det=4, n_samp=6000, nside=64, with_hwp=with_hwp, n_split=3, gap=7)
    n_samp, n_det = c["n_samp"], c["n_det"]
    want = run_chain(oracle, c, nest=True)
    hold = []
    pt, sfl = _descriptor(torch, capi, c, True, False, 3, 1, hold)
    step = 37
    ivl = c["intervals"]
    n_amp_views = np.array([-(-(int(v["last"]) - int(v["first"])) // step) for v in ivl], dtype=np.int64)
    per_det = int(n_amp_views.sum())
    amp_offsets = np.arange(n_det, dtype=np.int64) * per_det
    rng = np.random.default_rng(6)
    amps = rng.standard_normal(n_det * per_det)
    aflags = (rng.random(amps.size) < 0.05).astype(np.uint8)
    d_amps, d_afl = _dev(torch, amps), _dev(torch, aflags)
    g2l = _dev(torch, want["g2l"])
    dfl = _dev(torch, c["det_flags"])
    sig = _dev(torch, c["tod"])
    rows = c["data_index"]
    tmpl = torch.zeros_like(sig)
    D.offset_add_to_signal_multi(step, amp_offsets, n_amp_views, d_amps.data_ptr(), d_afl.data_ptr(), rows, tmpl.data_ptr(),
                                 n_samp, ivl)
    cleaned = sig - tmpl
    z_ref = torch.zeros(want["zmap"].shape, dtype=torch.float64, device="cuda")
    D.otf_build_noise_weighted(pt, g2l.data_ptr(), z_ref.data_ptr(), c["n_pix_submap"], rows, cleaned.data_ptr(),
                               c["flag_index"], dfl.data_ptr(), n_samp, c["det_scale"], 1, n_samp, ivl, sfl.data_ptr(),
                               c["shared_flags"].size, 1)
    z = torch.zeros_like(z_ref)
    D.otf_offset_clean_accumulate(pt, step, amp_offsets, n_amp_views, d_amps.data_ptr(), d_afl.data_ptr(), g2l.data_ptr(),
                                  z.data_ptr(), c["n_pix_submap"], rows, sig.data_ptr(), c["flag_index"], dfl.data_ptr(),
                                  n_samp, c["det_scale"], 1, n_samp, ivl, sfl.data_ptr(), c["shared_flags"].size, 1)
    torch.cuda.synchronize()
    zr, zo = z_ref.cpu().numpy(), z.cpu().numpy()
    assert np.max(np.abs(zr)) > 0 and np.array_equal(zr != 0, zo != 0)
    assert np.max(np.abs(zr - zo)) < 1e-12 * np.max(np.abs(zr))
    # the cached-pointing form on the oracle's pixels / weights (weights agree to 1e-12: tolerance)
    if n_samp % 2 == 0:
        pix, wts = _dev(torch, want["pixels"]), _dev(torch, want["weights"])
        z_c = torch.zeros_like(z_ref)
        D.offset_clean_accumulate(step, amp_offsets, n_amp_views, d_amps.data_ptr(), d_afl.data_ptr(), g2l.data_ptr(),
                                  z_c.data_ptr(), c["n_pix_submap"], 3, c["pixel_index"], pix.data_ptr(), c["weight_index"],
                                  wts.data_ptr(), rows, sig.data_ptr(), c["flag_index"], dfl.data_ptr(), n_samp,
                                  c["det_scale"], 1, n_samp, ivl, sfl.data_ptr(), c["shared_flags"].size, 1)
        torch.cuda.synchronize()
        assert np.max(np.abs(z_c.cpu().numpy() - zo)) < 1e-11 * np.max(np.abs(zr))
    # from the compact pixel cache
    cp = _compact(torch, capi, c, g2l, _dev(torch, want["pixels"]), want["zmap"].shape[0])
    ptc, _ = _descriptor(torch, capi, c, True, False, 3, 1, hold, compact=cp)
    z_k = torch.zeros_like(z_ref)
    D.otf_offset_clean_accumulate(ptc, step, amp_offsets, n_amp_views, d_amps.data_ptr(), d_afl.data_ptr(), g2l.data_ptr(),
                                  z_k.data_ptr(), c["n_pix_submap"], rows, sig.data_ptr(), c["flag_index"], dfl.data_ptr(),
                                  n_samp, c["det_scale"], 1, n_samp, ivl, sfl.data_ptr(), c["shared_flags"].size, 1)
    torch.cuda.synchronize()
    assert np.max(np.abs(z_k.cpu().numpy() - zr)) < 1e-12 * np.max(np.abs(zr))


def test_otf_argument_errors(env):
    torch, capi = env
    c = make_case(n_det=2, n_samp=100, nside=16)
    hold = []
    with pytest.raises(RuntimeError, match="nnz"):
        pt, sfl = _descriptor(torch, capi, c, True, False, 2, 1, hold)
        z = torch.zeros(10, dtype=torch.float64, device="cuda")
        capi.dev.otf_scan_map(pt, z.data_ptr(), z.data_ptr(), c["n_pix_submap"], z.data_ptr(), c["data_index"],
                              c["n_samp"], c["intervals"])
    c["nside"] = 48
    with pytest.raises(RuntimeError, match="power of two"):
        pt, sfl = _descriptor(torch, capi, c, True, False, 3, 1, hold)
        z = torch.zeros(10, dtype=torch.float64, device="cuda")
        capi.dev.otf_scan_map(pt, z.data_ptr(), z.data_ptr(), c["n_pix_submap"], z.data_ptr(), c["data_index"],
                              c["n_samp"], c["intervals"])
