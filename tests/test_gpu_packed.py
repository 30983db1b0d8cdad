"""GPU: the solver's packed pointing cache (csrc/packed_pointing.hip: 20 instead of 33 bytes per detector-sample) against
the sweeps over the original arrays: the projection bit for bit, the accumulation to the rounding of its atomic
additions, the complete MapMaker to rounding; and the cases in which packing must refuse."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(n_det=6, n_samp=6000, nside=64, seed=5, odd_views=True, pair_cal=False):
    import torch

    from toast_amd import capi, synth

    rng = np.random.default_rng(seed)
    nps = 12 * nside * nside // 48 if nside >= 2 else 12
    n_submap = 12 * nside * nside // nps
    fp, gamma = synth.hex_focalplane(n_det, fov_deg=6.0)
    bore = synth.satellite_boresight(n_samp, 50.0, 60.0, 30.0, 600.0, 65.0)
    if odd_views:      # three views with odd first samples and odd lengths: peeled heads and tails
        ivl = np.zeros(3, dtype=synth.interval_dtype)
        for k, (a, b) in enumerate(((3, 1500), (1701, 4000), (4100, n_samp - 1))):
            ivl[k]["first"], ivl[k]["last"] = a, b
            ivl[k]["start"], ivl[k]["stop"] = a / 50.0, b / 50.0
    else:
        ivl = synth.make_intervals(n_samp, 1, 50.0)
    dev = torch.device("cuda", 0)
    D = capi.dev
    idx = np.arange(n_det, dtype=np.int32)
    d_bore = torch.from_numpy(bore).to(dev)
    d_quats = torch.empty((n_det, n_samp, 4), dtype=torch.float64, device=dev)
    d_pix = torch.full((n_det, n_samp), -1, dtype=torch.int64, device=dev)
    d_w = torch.zeros((n_det, n_samp, 3), dtype=torch.float64, device=dev)
    d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)
    sflags = (rng.random(n_samp) < 0.02).astype(np.uint8)
    d_sflags = torch.from_numpy(sflags).to(dev)
    D.pointing_detector(fp, d_bore.data_ptr(), idx, d_quats.data_ptr(), n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, 0)
    D.pixels_healpix(idx, d_quats.data_ptr(), d_sflags.data_ptr(), n_samp, 1, idx, d_pix.data_ptr(), n_samp, ivl,
                     d_hsub.data_ptr(), n_submap, nps, nside, True, 0)
    cal = 0.5 + rng.random(n_det)
    if pair_cal:       # both detectors of a pair with the same calibration (and efficiency): Q / U weights of opposite sign
        cal = np.repeat(cal[: (n_det + 1) // 2], 2)[:n_det].copy()
    D.stokes_weights_IQU(idx, d_quats.data_ptr(), idx, d_w.data_ptr(), n_samp, 0, 0, ivl, np.zeros(n_det), gamma, cal,
                         False, 0)
    torch.cuda.synchronize()
    hit = d_hsub.cpu().numpy() != 0
    # leave one hit submap out of the local map: samples whose pixel is not local
    hit_idx = np.flatnonzero(hit)
    hit[hit_idx[len(hit_idx) // 2]] = False
    g2l = np.full(n_submap, -1, dtype=np.int64)
    g2l[hit] = np.arange(int(hit.sum()))
    n_local = int(hit.sum())
    dflags = (rng.random((n_det, n_samp)) < 0.03).astype(np.uint8) * 3        # bits 0 and 1
    pflags = (rng.random((n_det, n_samp)) < 0.05).astype(np.uint8)
    step = 250
    n_amp_view = [int(np.ceil((int(v["last"]) - int(v["first"])) / step)) for v in ivl]
    n_amp_det = int(np.sum(n_amp_view))
    ao = np.arange(n_det, dtype=np.int64) * n_amp_det
    amps = rng.standard_normal(n_det * n_amp_det)
    amp_flags = (rng.random(n_det * n_amp_det) < 0.05).astype(np.uint8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    return dict(D=D, torch=torch, dev=dev, n_det=n_det, n_samp=n_samp, nps=nps, n_local=n_local, idx=idx, ivl=ivl, step=step,
                nav=np.array(n_amp_view, dtype=np.int64), ao=ao, d_pix=d_pix, d_w=d_w, d_g2l=t(g2l), d_dflags=t(dflags),
                d_pflags=t(pflags), d_sflags=d_sflags, d_amps=t(amps), d_aflags=t(amp_flags), n_amp=n_det * n_amp_det,
                detw=0.5 + rng.random(n_det), cal=cal, zmap0=rng.standard_normal((n_local, nps, 3)))


def _pack(s, dmask=1, smask=1, pmask=1, pair_words=False):
    torch, D = s["torch"], s["D"]
    key = torch.zeros((s["n_det"], s["n_samp"]), dtype=torch.int32, device=s["dev"])
    qu = torch.zeros((s["n_det"], s["n_samp"], 2), dtype=torch.float64, device=s["dev"])
    cal = torch.zeros(s["n_det"], dtype=torch.float64, device=s["dev"])
    ok = D.offset_pack_pointing(s["d_g2l"].data_ptr(), s["nps"], s["idx"], s["d_pix"].data_ptr(), s["idx"],
                                s["d_w"].data_ptr(), s["idx"], s["d_dflags"].data_ptr(), s["n_samp"], dmask,
                                s["d_sflags"].data_ptr(), s["n_samp"], smask, s["idx"], s["d_pflags"].data_ptr(),
                                s["n_samp"], pmask, s["n_samp"], s["ivl"], key.data_ptr(), qu.data_ptr(), cal.data_ptr(),
                                pair_words=pair_words)
    if pair_words:
        return ok, key, qu, cal
    assert ok[1] is False
    return ok[0], key, qu, cal


@pytest.mark.parametrize("n_det,odd_views", [(6, True), (5, True), (6, False)])
def test_packed_sweeps_equal_the_sweeps_over_the_original_arrays(n_det, odd_views):
    s = _setup(n_det=n_det, odd_views=odd_views)
    torch, D = s["torch"], s["D"]
    ok, key, qu, cal = _pack(s)
    assert ok
    assert np.array_equal(cal.cpu().numpy(), s["cal"])
    # the packed words say what the original arrays say
    k = key.cpu().numpy().view(np.uint32)
    pix = s["d_pix"].cpu().numpy()
    g2l = s["d_g2l"].cpu().numpy()
    in_view = np.zeros(s["n_samp"], dtype=bool)
    for v in s["ivl"]:
        in_view[int(v["first"]):int(v["last"])] = True
    gsm = np.where(pix >= 0, pix // s["nps"], 0)
    local = (pix >= 0) & (g2l[gsm] >= 0)
    off = np.where(local, g2l[gsm] * s["nps"] + pix % s["nps"] + 1, 0)
    assert np.array_equal((k & 0x3fffffff)[:, in_view], off[:, in_view])
    acc = ((s["d_dflags"].cpu().numpy() & 1) != 0) | ((s["d_sflags"].cpu().numpy() & 1) != 0)[None, :]
    assert np.array_equal(((k >> 30) & 1)[:, in_view].astype(bool), acc[:, in_view])
    assert np.array_equal((k >> 31)[:, in_view].astype(bool), ((s["d_pflags"].cpu().numpy() & 1) != 0)[:, in_view])
    assert local.sum() < (pix >= 0).sum()         # (the case "pixel outside the local map" is present)
    assert np.array_equal(qu.cpu().numpy()[:, in_view], s["d_w"].cpu().numpy()[:, in_view, 1:])

    # projection: bit for bit
    zmap = torch.from_numpy(s["zmap0"]).to(s["dev"])
    out_a = torch.zeros(s["n_amp"], dtype=torch.float64, device=s["dev"])
    out_b = torch.zeros_like(out_a)
    D.offset_scan_project(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), out_a.data_ptr(), s["d_aflags"].data_ptr(),
                          s["d_g2l"].data_ptr(), zmap.data_ptr(), s["nps"], 3, s["idx"], s["d_pix"].data_ptr(), s["idx"],
                          s["d_w"].data_ptr(), s["idx"], s["d_pflags"].data_ptr(), 1, s["detw"], s["n_samp"], s["ivl"])
    D.offset_scan_project_packed(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), out_b.data_ptr(),
                                 s["d_aflags"].data_ptr(), zmap.data_ptr(), key.data_ptr(), qu.data_ptr(), cal.data_ptr(),
                                 s["detw"], s["n_samp"], s["ivl"])
    torch.cuda.synchronize()
    a, b = out_a.cpu().numpy(), out_b.cpu().numpy()
    assert np.any(a != 0)
    # (the amplitude sums are atomic additions of run totals: equal to rounding, and equal bit for bit wherever a
    # baseline received one run)
    np.testing.assert_allclose(b, a, rtol=0, atol=1e-12 * np.max(np.abs(a)))

    # accumulation
    za = torch.zeros((s["n_local"], s["nps"], 3), dtype=torch.float64, device=s["dev"])
    zb = torch.zeros_like(za)
    D.offset_accumulate(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(), s["d_g2l"].data_ptr(),
                        za.data_ptr(), s["nps"], 3, s["idx"], s["d_pix"].data_ptr(), s["idx"], s["d_w"].data_ptr(), s["idx"],
                        s["d_dflags"].data_ptr(), s["n_samp"], s["detw"], 1, s["n_samp"], s["ivl"], s["d_sflags"].data_ptr(),
                        s["n_samp"], 1)
    D.offset_accumulate_packed(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(), zb.data_ptr(),
                               key.data_ptr(), qu.data_ptr(), cal.data_ptr(), s["detw"], s["n_samp"], s["ivl"])
    torch.cuda.synchronize()
    a, b = za.cpu().numpy(), zb.cpu().numpy()
    assert np.any(a != 0) and np.array_equal(a != 0, b != 0)
    np.testing.assert_allclose(b, a, rtol=0, atol=1e-12 * np.max(np.abs(a)))


def _sweeps(s, key=None, qu=None, cal=None, pair=False):
    """(amplitudes out, zmap) of the projection and the accumulation: from the original arrays, or from a packed cache."""
    torch, D = s["torch"], s["D"]
    zmap = torch.from_numpy(s["zmap0"]).to(s["dev"])
    out = torch.zeros(s["n_amp"], dtype=torch.float64, device=s["dev"])
    z = torch.zeros((s["n_local"], s["nps"], 3), dtype=torch.float64, device=s["dev"])
    if key is None:
        D.offset_scan_project(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), out.data_ptr(), s["d_aflags"].data_ptr(),
                              s["d_g2l"].data_ptr(), zmap.data_ptr(), s["nps"], 3, s["idx"], s["d_pix"].data_ptr(),
                              s["idx"], s["d_w"].data_ptr(), s["idx"], s["d_pflags"].data_ptr(), 1, s["detw"], s["n_samp"],
                              s["ivl"])
        D.offset_accumulate(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(),
                            s["d_g2l"].data_ptr(), z.data_ptr(), s["nps"], 3, s["idx"], s["d_pix"].data_ptr(), s["idx"],
                            s["d_w"].data_ptr(), s["idx"], s["d_dflags"].data_ptr(), s["n_samp"], s["detw"], 1, s["n_samp"],
                            s["ivl"], s["d_sflags"].data_ptr(), s["n_samp"], 1)
    else:
        D.offset_scan_project_packed(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), out.data_ptr(),
                                     s["d_aflags"].data_ptr(), zmap.data_ptr(), key.data_ptr(), qu.data_ptr(),
                                     cal.data_ptr(), s["detw"], s["n_samp"], s["ivl"], pair_words=pair)
        D.offset_accumulate_packed(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(),
                                   z.data_ptr(), key.data_ptr(), qu.data_ptr(), cal.data_ptr(), s["detw"], s["n_samp"],
                                   s["ivl"], pair_words=pair)
    torch.cuda.synchronize()
    return out.cpu().numpy(), z.cpu().numpy()


@pytest.mark.parametrize("n_det,odd_views", [(6, True), (5, True), (8, False)])
def test_pair_words(n_det, odd_views):
    """Co-pointing pairs share their pixels: one word per pair-sample (18 B per detector-sample), a lone last detector,
    and the sweeps that read them against the sweeps over the original arrays."""
    import os

    if os.environ.get("TOAST_HIP_PAIR", "1") == "0":
        pytest.skip("detector-pair kernels switched off")
    s = _setup(n_det=n_det, odd_views=odd_views)
    (ok, pair), key, qu, cal = _pack(s, pair_words=True)
    assert ok and pair
    k = key.cpu().numpy().view(np.uint32)
    ok_d, key_d, _, _ = _pack(s)                       # the per-detector words of the same rows
    kd = key_d.cpu().numpy().view(np.uint32)
    in_view = np.zeros(s["n_samp"], dtype=bool)
    for v in s["ivl"]:
        in_view[int(v["first"]):int(v["last"])] = True
    for b in range((n_det + 1) // 2):
        a_row, b_row = kd[2 * b], (kd[2 * b + 1] if 2 * b + 1 < n_det else None)
        w = k[2 * b]
        assert np.array_equal((w & 0x0fffffff)[in_view], (a_row & 0x3fffffff)[in_view])
        assert np.array_equal(((w >> 28) & 3)[in_view], (a_row >> 30)[in_view])
        if b_row is not None:
            assert np.array_equal((b_row & 0x3fffffff)[in_view], (a_row & 0x3fffffff)[in_view])
            assert np.array_equal(((w >> 30) & 3)[in_view], (b_row >> 30)[in_view])
        else:
            assert np.all(((w >> 30) & 3)[in_view] == 3)
    ref_out, ref_z = _sweeps(s)
    out, z = _sweeps(s, key, qu, cal, pair=True)
    assert np.any(ref_out != 0) and np.any(ref_z != 0)
    np.testing.assert_allclose(out, ref_out, rtol=0, atol=1e-12 * np.max(np.abs(ref_out)))
    assert np.array_equal(z != 0, ref_z != 0)
    np.testing.assert_allclose(z, ref_z, rtol=0, atol=1e-12 * np.max(np.abs(ref_z)))
    # a pair that does not agree on a pixel: no pair words, the per-detector words still pack
    first = int(s["ivl"][0]["first"])
    s["d_pix"][1, first + 10] = s["d_pix"][1, first + 400]
    assert int(s["d_pix"][1, first + 10]) != int(s["d_pix"][0, first + 10])
    (ok, pair), key, qu, cal = _pack(s, pair_words=True)
    assert ok and not pair
    ref_out, ref_z = _sweeps(s)
    out, z = _sweeps(s, key, qu, cal, pair=False)
    np.testing.assert_allclose(out, ref_out, rtol=0, atol=1e-12 * np.max(np.abs(ref_out)))
    np.testing.assert_allclose(z, ref_z, rtol=0, atol=1e-12 * np.max(np.abs(ref_z)))


@pytest.mark.parametrize("n_det,odd_views", [(6, True), (5, True), (8, False)])
def test_pair_weight_sums(n_det, odd_views):
    """Orthogonal pairs of equal calibration: q_a + q_b and u_a + u_b are exact and fit a float, the partner's weights come
    back EXACTLY as (sum) - q_a (checked here on the host, sample by sample), and the sweeps that read 4 + 16 + 8 B per
    pair-sample (14 B per detector-sample) give what the sweeps over the original arrays give.  Pairs whose calibrations
    differ are refused and keep their 18-byte form."""
    import os

    if os.environ.get("TOAST_HIP_PAIR", "1") == "0":
        pytest.skip("detector-pair kernels switched off")
    s = _setup(n_det=n_det, odd_views=odd_views, pair_cal=True)
    torch, D = s["torch"], s["D"]
    (ok, pair), key, qu, cal = _pack(s, pair_words=True)
    assert ok and pair
    n_pairs = (n_det + 1) // 2
    corr = torch.full((n_pairs, s["n_samp"], 2), 7.0, dtype=torch.float32, device=s["dev"])
    assert D.offset_pack_pair_weights(qu.data_ptr(), corr.data_ptr(), n_det, s["n_samp"], s["ivl"])
    in_view = np.zeros(s["n_samp"], dtype=bool)
    for v in s["ivl"]:
        in_view[int(v["first"]):int(v["last"])] = True
    q, c = qu.cpu().numpy(), corr.cpu().numpy()
    n_marked = 0
    for b in range(n_det // 2):
        # a NaN marker: the sum does not fit a float exactly (weights that are rounding noise around zero): the sweeps read
        # the partner's own row there.  Everywhere else the partner's weight comes back bit for bit.
        marker = np.isnan(c[b])
        n_marked += int(marker[in_view].sum())
        rebuilt = c[b].astype(np.float64) - q[2 * b]
        good = in_view[:, None] & ~marker
        assert np.array_equal(rebuilt[good], q[2 * b + 1][good])                # bit for bit
        assert np.max(np.abs(c[b][good])) < 1e-12                               # a few ulps of weights of order one
        assert np.all(np.abs(q[2 * b + 1][in_view[:, None] & marker]) < 1e-14)  # markers only where the weight is noise
    assert n_marked <= 1e-2 * 2 * in_view.sum() * (n_det // 2)
    assert np.all(c[:, ~in_view] == 7.0)                                        # samples outside the views are not touched
    ref_out, ref_z = _sweeps(s)
    zmap = torch.from_numpy(s["zmap0"]).to(s["dev"])
    out = torch.zeros(s["n_amp"], dtype=torch.float64, device=s["dev"])
    z = torch.zeros((s["n_local"], s["nps"], 3), dtype=torch.float64, device=s["dev"])
    # the partner rows of the Q / U array are NOT read any more, except behind a marker: poison the rest
    qu_poison = qu.clone()
    keep = torch.isnan(corr[: n_det // 2])
    part = qu_poison[1::2]
    part[~keep] = float("nan")
    qu_poison[1::2] = part
    D.offset_scan_project_packed(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), out.data_ptr(),
                                 s["d_aflags"].data_ptr(), zmap.data_ptr(), key.data_ptr(), qu_poison.data_ptr(),
                                 cal.data_ptr(), s["detw"], s["n_samp"], s["ivl"], pair_words=True,
                                 pair_corr=corr.data_ptr())
    D.offset_accumulate_packed(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(),
                               z.data_ptr(), key.data_ptr(), qu_poison.data_ptr(), cal.data_ptr(), s["detw"], s["n_samp"],
                               s["ivl"], pair_words=True, pair_corr=corr.data_ptr())
    torch.cuda.synchronize()
    out, z = out.cpu().numpy(), z.cpu().numpy()
    assert np.any(ref_out != 0) and np.any(ref_z != 0)
    np.testing.assert_allclose(out, ref_out, rtol=0, atol=1e-12 * np.max(np.abs(ref_out)))
    assert np.array_equal(z != 0, ref_z != 0)
    np.testing.assert_allclose(z, ref_z, rtol=0, atol=1e-12 * np.max(np.abs(ref_z)))
    # pair sums without pair words make no sense
    with pytest.raises(RuntimeError):
        D.offset_accumulate_packed(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(),
                                   z_dummy(s).data_ptr(), key.data_ptr(), qu.data_ptr(), cal.data_ptr(), s["detw"],
                                   s["n_samp"], s["ivl"], pair_words=False, pair_corr=corr.data_ptr())
    # different calibrations inside a pair: the sums are of the size of the weights themselves -> refused
    s2 = _setup(n_det=n_det, odd_views=odd_views, pair_cal=False)
    (ok2, pair2), key2, qu2, cal2 = _pack(s2, pair_words=True)
    assert ok2 and pair2
    if n_det >= 2:
        assert not D.offset_pack_pair_weights(qu2.data_ptr(), corr.data_ptr(), n_det, s2["n_samp"], s2["ivl"])


@pytest.mark.parametrize("n_det,odd_views,pair_cal", [(6, True, True), (5, True, True), (8, False, True), (6, True, False)])
def test_pack_in_one_sweep_equals_the_separate_passes(n_det, odd_views, pair_cal):
    """Round 6: toast_hip_offset_pack_pointing_onepass_dev writes the pair words, both rows of Q / U weights, the intensity
    weights and the pair weight sums in ONE sweep over pixels / weights / flags; the outputs are those of pack -> pair
    check -> pair merge -> pair weights bit for bit (row 2 b + 1 of the key array, which the pair-word sweeps never read,
    is left alone), including the NaN markers, the untouched samples outside the views and the verdict on the sums."""
    import os

    if os.environ.get("TOAST_HIP_PAIR", "1") == "0":
        pytest.skip("detector-pair kernels switched off")
    s = _setup(n_det=n_det, odd_views=odd_views, pair_cal=pair_cal)
    torch, D = s["torch"], s["D"]
    (ok, pair), key, qu, cal = _pack(s, pair_words=True)
    assert ok and pair
    n_pairs = (n_det + 1) // 2
    corr = torch.full((n_pairs, s["n_samp"], 2), 7.0, dtype=torch.float32, device=s["dev"])
    sums_ok = D.offset_pack_pair_weights(qu.data_ptr(), corr.data_ptr(), n_det, s["n_samp"], s["ivl"])
    assert sums_ok == (pair_cal or n_det < 2)
    key1 = torch.full((s["n_det"], s["n_samp"]), -3, dtype=torch.int32, device=s["dev"])
    qu1 = torch.zeros((s["n_det"], s["n_samp"], 2), dtype=torch.float64, device=s["dev"])
    cal1 = torch.zeros(s["n_det"], dtype=torch.float64, device=s["dev"])
    corr1 = torch.full((n_pairs, s["n_samp"], 2), 7.0, dtype=torch.float32, device=s["dev"])
    ok1, pair1, sums1 = D.offset_pack_pointing_onepass(
        s["d_g2l"].data_ptr(), s["nps"], s["idx"], s["d_pix"].data_ptr(), s["idx"], s["d_w"].data_ptr(), s["idx"],
        s["d_dflags"].data_ptr(), s["n_samp"], 1, s["d_sflags"].data_ptr(), s["n_samp"], 1, s["idx"], s["d_pflags"].data_ptr(),
        s["n_samp"], 1, s["n_samp"], s["ivl"], key1.data_ptr(), qu1.data_ptr(), cal1.data_ptr(), corr1.data_ptr())
    assert (ok1, pair1, sums1) == (True, True, sums_ok)
    in_view = np.zeros(s["n_samp"], dtype=bool)
    for v in s["ivl"]:
        in_view[int(v["first"]):int(v["last"])] = True
    k0, k1 = key.cpu().numpy(), key1.cpu().numpy()
    assert np.array_equal(k1[0::2][:, in_view], k0[0::2][:, in_view])          # the pair words
    assert np.all(k1[1::2] == -3) and np.all(k1[:, ~in_view] == -3)            # nothing else is written
    q0, q1 = qu.cpu().numpy(), qu1.cpu().numpy()
    assert np.array_equal(q1[:, in_view].view(np.uint64), q0[:, in_view].view(np.uint64))
    assert np.all(q1[:, ~in_view] == 0.0)
    assert np.array_equal(cal1.cpu().numpy(), cal.cpu().numpy())
    c0, c1 = corr.cpu().numpy(), corr1.cpu().numpy()
    assert np.array_equal(c1.view(np.uint32), c0.view(np.uint32))              # sums, NaN markers and the untouched 7.0s
    # pairs that do not see the same pixels: the separate passes run behind the same call and say so
    s3 = _setup(n_det=4, odd_views=odd_views, pair_cal=True)
    pix = s3["d_pix"].clone()
    pix[1] = torch.roll(pix[1], 7)
    ok3, pair3, sums3 = D.offset_pack_pointing_onepass(
        s3["d_g2l"].data_ptr(), s3["nps"], s3["idx"], pix.data_ptr(), s3["idx"], s3["d_w"].data_ptr(), s3["idx"],
        s3["d_dflags"].data_ptr(), s3["n_samp"], 1, s3["d_sflags"].data_ptr(), s3["n_samp"], 1, s3["idx"],
        s3["d_pflags"].data_ptr(), s3["n_samp"], 1, s3["n_samp"], s3["ivl"], key1[:4].data_ptr(), qu1[:4].data_ptr(),
        cal1[:4].data_ptr(), corr1[:2].data_ptr())
    assert (ok3, pair3, sums3) == (True, False, False)


@pytest.mark.parametrize("n_det,odd_views", [(6, True), (5, True), (8, False)])
def test_clean_accumulate_equals_subtract_then_bin(n_det, odd_views):
    """Round 6: toast_hip_offset_clean_accumulate_dev -- zmap += A^T N^-1 (d - M a) in one sweep, the cleaned timestream
    formed in registers -- against the two operators it stands in for on the SAME device arrays: the offset template added
    into a zeroed buffer (add_to_signal: 0 + a for unflagged amplitudes), d - template, build_noise_weighted.  Views with
    odd first samples and odd lengths (peeled heads and tails), flagged samples, flagged amplitudes, a lone last detector, a
    submap that is not local; every map value to the rounding of the atomic additions, the same set of touched pixels; a
    row selection that is a permutation, to show that the signal's row indices are honoured."""
    s = _setup(n_det=n_det, odd_views=odd_views, pair_cal=True)
    torch, D = s["torch"], s["D"]
    gen = torch.Generator(device=s["dev"])
    gen.manual_seed(7)
    sig = torch.empty((n_det, s["n_samp"]), dtype=torch.float64, device=s["dev"]).normal_(0.0, 1.0, generator=gen)
    perm = np.roll(np.arange(n_det, dtype=np.int32), 1)            # detector k's timestream lives in row perm[k]
    tmpl = torch.zeros_like(sig)
    D.offset_add_to_signal_multi(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(), perm,
                                 tmpl.data_ptr(), s["n_samp"], s["ivl"])
    cleaned = sig - tmpl
    z_ref = torch.zeros((s["n_local"], s["nps"], 3), dtype=torch.float64, device=s["dev"])
    D.build_noise_weighted(s["d_g2l"].data_ptr(), z_ref.data_ptr(), s["nps"], 3, s["idx"], s["d_pix"].data_ptr(), s["idx"],
                           s["d_w"].data_ptr(), perm, cleaned.data_ptr(), s["idx"], s["d_dflags"].data_ptr(), s["n_samp"],
                           s["detw"], 1, s["n_samp"], s["ivl"], s["d_sflags"].data_ptr(), s["n_samp"], 1)
    z = torch.zeros_like(z_ref)
    D.offset_clean_accumulate(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(),
                              s["d_g2l"].data_ptr(), z.data_ptr(), s["nps"], 3, s["idx"], s["d_pix"].data_ptr(), s["idx"],
                              s["d_w"].data_ptr(), perm, sig.data_ptr(), s["idx"], s["d_dflags"].data_ptr(), s["n_samp"],
                              s["detw"], 1, s["n_samp"], s["ivl"], s["d_sflags"].data_ptr(), s["n_samp"], 1)
    torch.cuda.synchronize()
    z, z_ref = z.cpu().numpy(), z_ref.cpu().numpy()
    assert np.any(z_ref != 0)
    assert np.array_equal(z != 0, z_ref != 0)
    np.testing.assert_allclose(z, z_ref, rtol=0, atol=1e-12 * np.max(np.abs(z_ref)))
    # shapes the one-sweep kernel does not take are refused, not mis-handled: nnz = 1
    with pytest.raises(RuntimeError):
        D.offset_clean_accumulate(s["step"], s["ao"], s["nav"], s["d_amps"].data_ptr(), s["d_aflags"].data_ptr(),
                                  s["d_g2l"].data_ptr(), z_dummy(s).data_ptr(), s["nps"], 1, s["idx"], s["d_pix"].data_ptr(),
                                  s["idx"], s["d_w"].data_ptr(), perm, sig.data_ptr(), s["idx"], s["d_dflags"].data_ptr(),
                                  s["n_samp"], s["detw"], 1, s["n_samp"], s["ivl"], s["d_sflags"].data_ptr(), s["n_samp"], 1)


@pytest.mark.parametrize("step", [128, 129, 200, 1024, 1500, 5000])
def test_wave_uniform_amplitude_lookup_equals_the_per_lane_one(step, monkeypatch):
    """Round 6: the packed pair-word sweeps look the amplitudes of a wave's 128 samples up ONCE (two scalars per detector,
    picked per lane by the position of the next baseline boundary) when a baseline has at least 128 samples.  Baseline
    lengths at the limit (128), just above, around a workgroup's 1024 samples and longer than a view, three views with
    odd first samples: the projection and the accumulation equal the per-lane look-up (TOAST_HIP_PACKED_UNIFORM_AMPS=0)
    to the rounding of the atomic additions -- the same runs are reduced in the same lanes -- with pair sums and without."""
    import os

    if os.environ.get("TOAST_HIP_PAIR", "1") == "0":
        pytest.skip("detector-pair kernels switched off")
    s = _setup(n_det=6, odd_views=True, pair_cal=True)
    torch, D = s["torch"], s["D"]
    rng = np.random.default_rng(step)
    nav = np.array([-(-(int(v["last"]) - int(v["first"])) // step) for v in s["ivl"]], dtype=np.int64)
    n_amp_det = int(nav.sum())
    ao = np.arange(s["n_det"], dtype=np.int64) * n_amp_det
    n_amp = s["n_det"] * n_amp_det
    d_amps = torch.from_numpy(rng.standard_normal(n_amp)).to(s["dev"])
    d_afl = torch.from_numpy((rng.random(n_amp) < 0.2).astype(np.uint8)).to(s["dev"])
    (ok, pair), key, qu, cal = _pack(s, pair_words=True)
    assert ok and pair
    corr = torch.zeros(((s["n_det"] + 1) // 2, s["n_samp"], 2), dtype=torch.float32, device=s["dev"])
    assert D.offset_pack_pair_weights(qu.data_ptr(), corr.data_ptr(), s["n_det"], s["n_samp"], s["ivl"])
    res = {}
    for uni in ("1", "0"):
        monkeypatch.setenv("TOAST_HIP_PACKED_UNIFORM_AMPS", uni)
        for cp in (0, corr.data_ptr()):
            zmap = torch.from_numpy(s["zmap0"]).to(s["dev"])
            out = torch.zeros(n_amp, dtype=torch.float64, device=s["dev"])
            z = torch.zeros((s["n_local"], s["nps"], 3), dtype=torch.float64, device=s["dev"])
            D.offset_scan_project_packed(step, ao, nav, d_amps.data_ptr(), out.data_ptr(), d_afl.data_ptr(), zmap.data_ptr(),
                                         key.data_ptr(), qu.data_ptr(), cal.data_ptr(), s["detw"], s["n_samp"], s["ivl"],
                                         pair_words=True, pair_corr=cp)
            D.offset_accumulate_packed(step, ao, nav, d_amps.data_ptr(), d_afl.data_ptr(), z.data_ptr(), key.data_ptr(),
                                       qu.data_ptr(), cal.data_ptr(), s["detw"], s["n_samp"], s["ivl"], pair_words=True,
                                       pair_corr=cp)
            torch.cuda.synchronize()
            res[(uni, bool(cp))] = (out.cpu().numpy(), z.cpu().numpy())
    for with_sums in (False, True):
        (o1, z1), (o0, z0) = res[("1", with_sums)], res[("0", with_sums)]
        assert np.any(o0 != 0) and np.any(z0 != 0)
        assert np.array_equal(o1 != 0, o0 != 0) and np.array_equal(z1 != 0, z0 != 0)
        assert np.max(np.abs(o1 - o0)) <= 1e-13 * np.max(np.abs(o0))
        assert np.max(np.abs(z1 - z0)) <= 1e-13 * np.max(np.abs(z0))
        # flagged amplitudes receive nothing, in either form
        assert not np.any(o1[d_afl.cpu().numpy() != 0])


def z_dummy(s):
    return s["torch"].zeros((s["n_local"], s["nps"], 3), dtype=s["torch"].float64, device=s["dev"])


def test_packing_refuses_what_it_cannot_represent():
    """An intensity weight that varies along a row (weights that did not come from stokes_weights_IQU): not packable, the
    caller keeps the original arrays.  Different masks give different flag bits."""
    s = _setup()
    ok, key, _, _ = _pack(s, dmask=2, smask=0, pmask=0)
    assert ok
    k = key.cpu().numpy().view(np.uint32)
    assert not np.any(k >> 31) and np.array_equal(((k >> 30) & 1).astype(bool)[:, 5:1400],
                                                  ((s["d_dflags"].cpu().numpy() & 2) != 0)[:, 5:1400])
    s["d_w"][2, 777, 0] += 1e-9
    ok, _, _, _ = _pack(s)
    assert not ok


@pytest.mark.parametrize("prior", [False, True])
def test_mapmaker_with_and_without_the_packed_cache(monkeypatch, prior):
    from toast_amd import capi, ops
    from toast_amd.data import defaults
    from toast_amd.templates import Offset
    from test_gpu_ops import make_solver_setup

    res = {}
    for packed in ("1", "0"):
        monkeypatch.setenv("TOAST_HIP_PACKED_POINTING", packed)
        seen = {}
        for name in ("offset_accumulate_packed", "offset_scan_project_packed", "offset_accumulate", "offset_pack_pointing",
                     "offset_pack_pointing_onepass"):
            real = getattr(capi.dev, name)

            def counted(*a, _real=real, _name=name, **k):
                seen[_name] = seen.get(_name, 0) + 1
                return _real(*a, **k)

            monkeypatch.setattr(capi.dev, name, counted)
        data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2,
                      use_noise_prior=prior, precond_width=10)
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl]), solve_rcond_threshold=1e-3,
                              map_rcond_threshold=1e-3, iter_max=12, convergence=1e-30)
        held0 = capi.alloc_stats()
        mapper.apply(data)
        res[packed] = (np.array(mapper.history), data["mm_solve_amplitudes"]["baselines"].local.copy(),
                       data["mm_map"].data.copy(), dict(seen))
        monkeypatch.undo()
        del held0
    h1, a1, m1, c1 = res["1"]
    h0, a0, m0, c0 = res["0"]
    # (with the noise prior the fused left-hand side is not used at all; the order-exact mode keeps to the operator sequence)
    can_pack = not prior and not capi.get_deterministic()
    if can_pack:
        # (the pack: in one sweep when the pair weight sums are on, else the separate passes)
        assert c1.get("offset_pack_pointing", 0) + c1.get("offset_pack_pointing_onepass", 0) >= 1, c1
        assert c1.get("offset_accumulate_packed", 0) >= 1, c1
        assert c1.get("offset_scan_project_packed", 0) >= 1 and c1.get("offset_accumulate", 0) == 0, c1
    assert c0.get("offset_pack_pointing", 0) + c0.get("offset_pack_pointing_onepass", 0) == 0, c0
    assert c0.get("offset_accumulate_packed", 0) == 0, c0
    assert len(h1) == len(h0)
    np.testing.assert_allclose(h1, h0, rtol=1e-7)
    assert np.max(np.abs(a1 - a0)) < 1e-9 * np.max(np.abs(a0))
    assert np.max(np.abs(m1 - m0)) < 1e-9 * np.max(np.abs(m0))


def test_uncached_pointing_with_the_packed_cache(monkeypatch):
    """full_pointing=False: the solver's packed cache expanded from the boresight in batches (BinMap.packed_cache) against
    the sweeps that evaluate the pointing on the fly, and against cached pointing: the same solve."""
    from toast_amd import capi, ops
    from toast_amd.data import defaults
    from toast_amd.templates import Offset
    from test_gpu_ops import make_solver_setup

    res = {}
    monkeypatch.setenv("TOAST_HIP_PACKED_POINTING", "1")
    for mode in ("otf", "otf_packed", "cached"):
        monkeypatch.setenv("TOAST_HIP_PACKED_POINTING", "1")
        seen = {}
        for name in ("offset_accumulate_packed", "otf_offset_accumulate", "offset_pack_pointing", "offset_pack_pairs",
                     "offset_pack_pointing_onepass", "otf_pixels_healpix"):
            real = getattr(capi.dev, name)

            def counted(*a, _real=real, _name=name, **k):
                seen[_name] = seen.get(_name, 0) + 1
                return _real(*a, **k)

            monkeypatch.setattr(capi.dev, name, counted)
        data, pix, sw, truth, sky = make_solver_setup(n_det=6, noise_rms=0.1)
        binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=(mode == "cached"),
                            packed_cache=(mode == "otf_packed"))
        tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
        mapper = ops.MapMaker(name="mm", keep_solver_products=True, det_data=defaults.det_data, binning=binner,
                              template_matrix=ops.TemplateMatrix(templates=[tmpl]), solve_rcond_threshold=1e-3,
                              map_rcond_threshold=1e-3, iter_max=12, convergence=1e-30)
        mapper.apply(data)
        res[mode] = (np.array(mapper.history), data["mm_solve_amplitudes"]["baselines"].local.copy(),
                     data["mm_map"].data.copy(), dict(seen))
        monkeypatch.undo()
    c = res["otf_packed"][3]
    # packed batch by batch: in one sweep each (pair words + pair weight sums), or -- pair weights switched off, or pairs
    # that do not share their pixels -- plain words per batch and one pair check / merge over all rows
    onepass, separate = c.get("offset_pack_pointing_onepass", 0), c.get("offset_pack_pointing", 0)
    assert onepass + separate >= 1 and (onepass >= 1 or c.get("offset_pack_pairs", 0) == 1), c
    assert c.get("offset_accumulate_packed", 0) >= 1 and c.get("otf_offset_accumulate", 0) == 0, c
    o = res["otf"][3]
    assert o.get("otf_offset_accumulate", 0) >= 1 and o.get("offset_pack_pointing", 0) + o.get("offset_pack_pointing_onepass", 0) == 0
    for other in ("otf", "cached"):
        h1, a1, m1, _ = res["otf_packed"]
        h0, a0, m0, _ = res[other]
        assert len(h1) == len(h0)
        np.testing.assert_allclose(h1, h0, rtol=1e-7)
        assert np.max(np.abs(a1 - a0)) < 1e-9 * np.max(np.abs(a0))
        assert np.max(np.abs(m1 - m0)) < 1e-9 * np.max(np.abs(m0))



def test_no_packed_cache_without_room(monkeypatch):
    """The packed cache is built only while a fifth of the device stays free afterwards; otherwise the sweeps over the
    original arrays run (same result)."""
    from toast_amd import capi, ops
    from toast_amd.data import defaults
    from toast_amd.templates import Offset
    from test_gpu_ops import make_solver_setup

    free, total = capi.accel_mem_info()
    assert 0 < free <= total
    monkeypatch.setattr(capi, "accel_mem_info", lambda: (total // 5 + 1000, total))
    seen = {}
    for name in ("offset_pack_pointing", "offset_accumulate", "offset_accumulate_packed"):
        real = getattr(capi.dev, name)

        def counted(*a, _real=real, _name=name, **k):
            seen[_name] = seen.get(_name, 0) + 1
            return _real(*a, **k)

        monkeypatch.setattr(capi.dev, name, counted)
    data, pix, sw, truth, sky = make_solver_setup(noise_rms=0.1)
    binner = ops.BinMap(pixel_dist="dist", pixel_pointing=pix, stokes_weights=sw, full_pointing=True)
    tmpl = Offset(step_time=20.0, noise_model=defaults.noise_model, name="baselines", good_fraction=0.2)
    mapper = ops.MapMaker(name="mm", det_data=defaults.det_data, binning=binner,
                          template_matrix=ops.TemplateMatrix(templates=[tmpl]), solve_rcond_threshold=1e-3,
                          map_rcond_threshold=1e-3, iter_max=5, convergence=1e-30)
    mapper.apply(data)
    assert seen.get("offset_pack_pointing", 0) == 0 and seen.get("offset_accumulate_packed", 0) == 0, seen
    assert seen.get("offset_accumulate", 0) >= 1 and len(mapper.history) == 5


@pytest.mark.parametrize("n_det,odd_views,n_samp", [(6, True, 6000), (5, True, 6000), (6, False, 6000), (4, True, 5999)])
def test_covariance_hits_and_signal_map_in_one_sweep(n_det, odd_views, n_samp):
    """Round 6: toast_hip_build_cov_hits_signal_dev -- inverse covariance, hits and zmap += A^T N^-1 d from ONE sweep over the
    pointing (k_build_cov_pair_v2<true, true>) -- against the calls it stands in for on the SAME device arrays: the
    inverse covariance + hits call (the same entry without a map) and build_noise_weighted with the signal's own row
    selection and scale.  Hits exactly, covariance and map to the rounding of the atomic additions, the same set of touched
    pixels; everything accumulates onto what is there; an odd row length (no 16-byte rows) falls back to the separate
    sweeps behind the same call and says so."""
    s = _setup(n_det=n_det, odd_views=odd_views, n_samp=n_samp, pair_cal=True)
    torch, D = s["torch"], s["D"]
    gen = torch.Generator(device=s["dev"])
    gen.manual_seed(11)
    sig = torch.empty((n_det, s["n_samp"]), dtype=torch.float64, device=s["dev"]).normal_(0.0, 1.0, generator=gen)
    perm = np.roll(np.arange(n_det, dtype=np.int32), 1)
    sig_scale = 0.25 + np.arange(n_det) * 0.125                     # (not the covariance's scale: both are honoured)
    shape = (s["n_local"], s["nps"])
    zeros = lambda k, dt=torch.float64: torch.zeros(shape + (k,), dtype=dt, device=s["dev"])       # noqa: E731
    common = lambda: (s["nps"], 3, s["idx"], s["d_pix"].data_ptr(), s["idx"], s["d_w"].data_ptr(), perm, sig.data_ptr(),    # noqa: E731
                      s["idx"], s["d_dflags"].data_ptr(), s["n_samp"], s["detw"], sig_scale, 1, s["n_samp"], s["ivl"],
                      s["d_sflags"].data_ptr(), s["n_samp"], 1)
    cov_ref, hits_ref, z_ref = zeros(6), zeros(1, torch.int64), zeros(3)
    assert D.build_cov_hits_signal(s["d_g2l"].data_ptr(), cov_ref.data_ptr(), hits_ref.data_ptr(), 0, *common()) is False
    D.build_noise_weighted(s["d_g2l"].data_ptr(), z_ref.data_ptr(), s["nps"], 3, s["idx"], s["d_pix"].data_ptr(), s["idx"],
                           s["d_w"].data_ptr(), perm, sig.data_ptr(), s["idx"], s["d_dflags"].data_ptr(), s["n_samp"],
                           sig_scale, 1, s["n_samp"], s["ivl"], s["d_sflags"].data_ptr(), s["n_samp"], 1)
    cov, hits, z = zeros(6) + 0.25, zeros(1, torch.int64) + 5, zeros(3) - 1.5
    fused = D.build_cov_hits_signal(s["d_g2l"].data_ptr(), cov.data_ptr(), hits.data_ptr(), z.data_ptr(), *common())
    torch.cuda.synchronize()
    import os

    expect_fused = (n_samp % 2 == 0) and n_det >= 2 and os.environ.get("TOAST_HIP_PAIR", "1") != "0" \
        and os.environ.get("TOAST_HIP_VEC2", "1") != "0" and not capi_deterministic()
    assert fused == expect_fused
    cov, hits, z = cov.cpu().numpy() - 0.25, hits.cpu().numpy() - 5, z.cpu().numpy() + 1.5
    cov_ref, hits_ref, z_ref = cov_ref.cpu().numpy(), hits_ref.cpu().numpy(), z_ref.cpu().numpy()
    assert hits_ref.sum() > 0 and np.array_equal(hits, hits_ref)
    np.testing.assert_allclose(cov, cov_ref, rtol=0, atol=1e-12 * max(np.max(np.abs(cov_ref)), 0.25))
    np.testing.assert_allclose(z, z_ref, rtol=0, atol=1e-12 * max(np.max(np.abs(z_ref)), 1.5))
    assert np.array_equal(np.abs(z) > 1e-9, np.abs(z_ref) > 1e-9) and np.any(z_ref != 0)


def capi_deterministic():
    from toast_amd import capi

    return bool(capi.get_deterministic())
