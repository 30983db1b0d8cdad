"""GPU parity tests proper: the HIP library, called through its C ABI, against the CPU
oracle on the same seeded inputs (and against the committed golden fixtures in
test_gpu_golden.py).  Bars: HEALPix pixel indices, hit-submap flags and pointing_detector
quaternions bit-exact; Stokes weights rtol 1e-12 (device libm vs glibc); accumulated maps
max|dz| / max|z| < 1e-12 (summation order differs, fp64)."""

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

ZTOL = 1e-12


@pytest.fixture(scope="module")
def hip():
    from toast_amd import capi

    assert capi.accel_enabled(), "no HIP device visible"
    capi.accel_assign_device(1, 0, 1.0, False)
    return capi


CASES = {
    "default": dict(),
    "split_gap_extra_hwp": dict(n_split=3, gap=5, extra_rows=2, with_hwp=True),
    "nside1024_single_det_noflags": dict(nside=1024, n_samp=5000, with_det_flags=False, n_det=1),
    "random_nside4096": dict(random_pointing=True, nside=4096, n_samp=20000),
    "ragged": dict(n_samp=1029, n_split=4, gap=1, n_det=3, nside=256),
    "tiny": dict(n_samp=3, n_det=2, nside=1, with_det_flags=False, with_shared_flags=False),
    "no_shared_flags": dict(with_shared_flags=False, n_samp=4097, nside=512),
    # detector pairs of the call do not share a pixel / odd detector count (pair-merge fallbacks)
    "unpaired_dets": dict(fp_roll=1, n_det=6, n_samp=6000, nside=128),
    "odd_dets": dict(n_det=5, n_samp=3000, nside=64, fp_roll=1),
    # degenerate inputs: a view without intervals, every sample flagged, one-sample intervals
    "no_intervals": dict(empty_intervals=True, n_samp=500, nside=32),
    "all_flagged": dict(all_flagged=True, n_samp=700, nside=32),
    "one_sample_intervals": dict(n_samp=24, n_split=24, n_det=2, nside=8),
    # ground CES: ~70 sweep intervals, flagged turnarounds, Nside 2048 (configs[4] structure)
    "ground_nside2048": dict(ground=True, n_samp=72000, rate=100.0, nside=2048, n_det=6, with_hwp=True),
}


def compare_chain(a, b):
    assert np.array_equal(a["quats"], b["quats"]), "pointing_detector quaternions differ"
    assert np.array_equal(a["pixels"], b["pixels"]), (
        "pixel mismatches: %d" % np.count_nonzero(a["pixels"] != b["pixels"])
    )
    assert np.array_equal(a["hsub"], b["hsub"])
    np.testing.assert_allclose(a["weights"], b["weights"], rtol=1e-12, atol=1e-14)
    scale = max(np.max(np.abs(b["zmap"])), 1e-300)
    assert np.max(np.abs(a["zmap"] - b["zmap"])) / scale < ZTOL
    tscale = max(np.max(np.abs(b["tod"])), 1e-300)
    assert np.max(np.abs(a["tod"] - b["tod"])) / tscale < 1e-11


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("nest", [True, False])
def test_chain_staged(hip, oracle, name, nest):
    """use_accel=False: host buffers staged through the GPU per call."""
    c = cases.make_case(**CASES[name])
    got = cases.run_chain(hip, c, nest=nest, tail=(False,))
    want = cases.run_chain(oracle, c, nest=nest)
    compare_chain(got, want)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("TOAST_TEST_FUZZ_SEEDS", "16")))))
def test_chain_random_cases(hip, oracle, seed):
    """Randomly drawn shapes and options of the whole chain (detector count incl. odd, ragged interval splits with gaps,
    Nside 1 .. 8192, NEST / RING, HWP, flags on / off, row indirection, broken detector pairs, random pointing, IAU)
    against the oracle: pixels bit-exact, the rest within the chain's tolerances."""
    rng = np.random.default_rng(5000 + seed)
    n_samp = int(rng.integers(1, 3000))
    kw = dict(
        n_det=int(rng.integers(1, 8)),
        n_samp=n_samp,
        nside=int(2 ** rng.integers(0, 14)),
        n_split=int(rng.integers(1, min(6, n_samp) + 1)),
        gap=int(rng.integers(0, 4)),
        with_shared_flags=bool(rng.integers(0, 2)),
        with_det_flags=bool(rng.integers(0, 2)),
        with_hwp=bool(rng.integers(0, 2)),
        extra_rows=int(rng.integers(0, 3)),
        seed=int(rng.integers(0, 1000)),
        random_pointing=bool(rng.integers(0, 2)),
        fp_roll=int(rng.integers(0, 2)),
    )
    if kw["n_det"] == 1:
        kw["extra_rows"] = 0
    nest, iau = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    c = cases.make_case(**kw)
    got = cases.run_chain(hip, c, nest=nest, iau=iau, tail=(False,))
    want = cases.run_chain(oracle, c, nest=nest, iau=iau)
    compare_chain(got, want)


@pytest.mark.parametrize("map_dtype", [np.float32, np.int64, np.int32])
def test_scan_map_dtypes(hip, oracle, map_dtype):
    c = cases.make_case(n_samp=3000, nside=128, n_split=2)
    got = cases.run_chain(hip, c, map_dtype=map_dtype, scan_scale=0.37, tail=(False,))
    want = cases.run_chain(oracle, c, map_dtype=map_dtype, scan_scale=0.37)
    tscale = np.max(np.abs(want["tod"]))
    assert np.max(np.abs(got["tod"] - want["tod"])) / tscale < 1e-11


def test_scan_map_modes(hip, oracle):
    """zero / add / subtract / scale variants; add-then-subtract returns the input exactly
    (reference test: src/toast/tests/ops_scan_map.py:99-172)."""
    c = cases.make_case(n_samp=2500, nside=64, n_split=2, gap=3)
    base = cases.run_chain(oracle, c)
    px, w, g2l = base["pixels"], base["weights"], base["g2l"]
    m = np.ascontiguousarray(base["zmap"])
    for zero, sub, mult in [(True, False, False), (False, False, False), (False, True, False), (False, False, True)]:
        t_h = c["tod"].copy()
        t_o = c["tod"].copy()
        hip.ops_scan_map_float64(g2l, c["n_pix_submap"], m, t_h, c["data_index"], px, c["pixel_index"], w,
                                 c["weight_index"], c["intervals"], 1.5, zero, sub, mult, False)
        oracle.scan_map(g2l, c["n_pix_submap"], m, t_o, c["data_index"], px, c["pixel_index"], w,
                        c["weight_index"], c["intervals"], 1.5, zero, sub, mult)
        np.testing.assert_allclose(t_h, t_o, rtol=1e-13, atol=1e-13 * np.max(np.abs(t_o)))
    t = c["tod"].copy()
    args = (g2l, c["n_pix_submap"], m, t, c["data_index"], px, c["pixel_index"], w, c["weight_index"], c["intervals"])
    hip.ops_scan_map_float64(*args, 1.0, False, False, False, False)
    hip.ops_scan_map_float64(*args, 1.0, False, True, False, False)
    np.testing.assert_allclose(t, c["tod"], rtol=0, atol=1e-12 * np.max(np.abs(m)))


def test_chain_accel_resident(hip, oracle):
    """use_accel=True: buffers registered with the memory manager, kernels run on the device
    copies, results fetched with accel_update_host (the reference's OmpManager protocol)."""
    c = cases.make_case(n_det=6, n_samp=7000, nside=256, n_split=2, extra_rows=1)
    want = cases.run_chain(oracle, c)
    rows, n_samp = c["rows"], c["n_samp"]
    quats = np.zeros((rows, n_samp, 4))
    pixels = np.full((rows, n_samp), -7, dtype=np.int64)
    weights = np.zeros((rows, n_samp, 3))
    tod = c["tod"].copy()
    hsub = np.zeros(c["n_submap"], dtype=np.uint8)
    bufs = dict(quats=quats, pixels=pixels, weights=weights, tod=tod, bore=c["boresight"],
                sflags=c["shared_flags"], dflags=c["det_flags"])
    for k, v in bufs.items():
        assert not hip.accel_present(v, k)
        hip.accel_create(v, k)
        hip.accel_update_device(v, k)
        assert hip.accel_present(v, k)
    with pytest.raises(RuntimeError):
        hip.accel_create(quats, "quats")  # already present (accelerator.cpp:339-347)
    hip.pointing_detector(c["focalplane"], c["boresight"], c["quat_index"], quats, c["intervals"],
                          c["shared_flags"], 1, True)
    hip.pixels_healpix(c["quat_index"], quats, c["shared_flags"], 1, c["pixel_index"], pixels, c["intervals"],
                       hsub, c["n_pix_submap"], c["nside"], True, True)
    hip.stokes_weights_IQU(c["quat_index"], quats, c["weight_index"], weights, c["hwp"], c["intervals"],
                           c["epsilon"], c["gamma"], c["cal"], False, True)
    assert np.array_equal(hsub, want["hsub"])
    g2l = want["g2l"]
    zmap = np.zeros_like(want["zmap"])
    hip.accel_create(zmap, "zmap")
    hip.accel_reset(zmap, "zmap")
    hip.build_noise_weighted(g2l, zmap, c["pixel_index"], pixels, c["weight_index"], weights, c["data_index"],
                             tod, c["flag_index"], c["det_flags"], c["det_scale"], 1, c["intervals"],
                             c["shared_flags"], 1, True)
    hip.ops_scan_map_float64(g2l, c["n_pix_submap"], zmap, tod, c["data_index"], pixels, c["pixel_index"],
                             weights, c["weight_index"], c["intervals"], 1.0, False, True, False, True)
    hip.noise_weight(tod, c["data_index"], c["intervals"], c["det_scale"], True)
    # host copies untouched until update_host
    assert np.all(pixels == -7)
    for k, v in dict(quats=quats, pixels=pixels, weights=weights, tod=tod, zmap=zmap).items():
        hip.accel_update_host(v, k)
    compare_chain(dict(quats=quats, pixels=pixels, hsub=hsub, weights=weights, zmap=zmap, tod=tod), want)
    for k, v in list(bufs.items()) + [("zmap", zmap)]:
        hip.accel_delete(v, k)
        assert not hip.accel_present(v, k)
    with pytest.raises(RuntimeError):
        hip.noise_weight(tod, c["data_index"], c["intervals"], c["det_scale"], True)  # not present


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("TOAST_TEST_FUZZ_SEEDS", "10")))))
def test_inverse_covariance_and_hits_in_one_call(hip, oracle, seed):  # noqa: C901
    """build_inverse_covariance_and_hits (hit map carried by the pair-merged inverse-covariance kernel, or two kernels
    behind the same call for I-only weights / a single detector / broken pairs) against the two separate calls and a
    NumPy statement of the accumulation: hits exactly, inverse covariance to rounding; it adds to existing contents."""
    rng = np.random.default_rng(17000 + seed)
    kw = dict(n_det=int(rng.integers(1, 8)), n_samp=int(rng.integers(50, 4000)), nside=int(2 ** rng.integers(2, 9)),
              n_split=int(rng.integers(1, 4)), gap=int(rng.integers(0, 5)), with_shared_flags=bool(rng.integers(0, 2)),
              with_det_flags=bool(rng.integers(0, 2)), fp_roll=int(rng.integers(0, 2)), seed=int(rng.integers(0, 99)))
    c = cases.make_case(**kw)
    ch = cases.run_chain(oracle, c, nest=True)
    pixels, g2l = ch["pixels"], ch["g2l"]
    n_local = max(int(np.count_nonzero(ch["hsub"])), 1)
    nnz = int(rng.choice([1, 3]))
    weights = ch["weights"] if nnz == 3 else np.ascontiguousarray(ch["weights"][:, :, 0])
    blk = nnz * (nnz + 1) // 2
    shape = (n_local, c["n_pix_submap"])
    args = (c["flag_index"], c["det_flags"], c["det_scale"], 1, c["intervals"], c["shared_flags"], 1)
    want_hits = np.zeros(shape + (1,), dtype=np.int64)
    want_cov = np.zeros(shape + (blk,))
    # NumPy statement of cov_accum_diag_hits / cov_accum_diag_invnpp over the intervals (toast_map_cov.cpp:66-153; the
    # kernels themselves are pinned to the reference's outputs in tests/test_gpu_golden.py)
    nps = c["n_pix_submap"]
    use_d, use_s = c["det_flags"].shape[1] == c["n_samp"], c["shared_flags"].size == c["n_samp"]
    for d in range(c["n_det"]):
        prow, wrow = pixels[c["pixel_index"][d]], weights[c["weight_index"][d]]
        for iv in c["intervals"]:
            sl = slice(int(iv["first"]), int(iv["last"]))
            p = prow[sl]
            good = p >= 0
            if use_d:
                good &= (c["det_flags"][c["flag_index"][d], sl] & 1) == 0
            if use_s:
                good &= (c["shared_flags"][sl] & 1) == 0
            pg = p[good]
            loc = g2l[pg // nps] * nps + pg % nps
            np.add.at(want_hits.reshape(-1), loc, 1)
            wg = wrow[sl][good].reshape(pg.size, nnz)
            off = 0
            for j in range(nnz):
                for k in range(j, nnz):
                    np.add.at(want_cov.reshape(-1, blk)[:, off], loc, wg[:, j] * wg[:, k] * c["det_scale"][d])
                    off += 1
    from toast_amd.accel import native

    sep_hits = np.zeros_like(want_hits)
    sep_cov = np.zeros_like(want_cov)
    hip = native()   # the pybind11 module carries these bindings
    hip.build_hit_map(g2l, sep_hits, c["pixel_index"], pixels, args[0], args[1], 1, args[4], args[5], 1, False)
    hip.build_inverse_covariance(g2l, sep_cov, c["pixel_index"], pixels, c["weight_index"], weights, *args, False)
    got_hits = np.full_like(want_hits, 5)                     # accumulates onto what is there
    got_cov = np.full_like(want_cov, 0.25)
    hip.build_inverse_covariance_and_hits(g2l, got_cov, got_hits, c["pixel_index"], pixels, c["weight_index"], weights,
                                          *args, False)
    assert np.array_equal(sep_hits, want_hits)
    assert np.array_equal(got_hits - 5, want_hits)
    scale = max(np.max(np.abs(want_cov)), 1e-300)
    assert np.max(np.abs(sep_cov - want_cov)) < 1e-12 * scale
    assert np.max(np.abs((got_cov - 0.25) - want_cov)) < 1e-12 * max(scale, 0.25)


def test_stokes_I_and_cov_and_offsets(hip, oracle):
    rng = np.random.default_rng(3)
    c = cases.make_case(n_det=3, n_samp=5000, n_split=3, gap=7)
    n_det, n_samp, ivl = c["n_det"], c["n_samp"], c["intervals"]
    w_h = np.zeros((n_det, n_samp))
    w_o = np.zeros((n_det, n_samp))
    idx = np.arange(n_det, dtype=np.int32)
    hip.stokes_weights_I(idx, w_h, ivl, c["cal"], False)
    oracle.stokes_weights_I(idx, w_o, ivl, c["cal"])
    assert np.array_equal(w_h, w_o)
    # cov_apply_diag
    for nnz in (1, 3):
        nsub, subsize = 5, 48
        blk = nnz * (nnz + 1) // 2
        mat = rng.standard_normal((nsub, subsize, blk))
        v_h = rng.standard_normal((nsub, subsize, nnz))
        v_o = v_h.copy()
        hip.cov_apply_diag(nsub, subsize, nnz, mat, v_h, False)
        oracle.cov_apply_diag(nsub, subsize, nnz, mat, v_o)
        assert np.array_equal(v_h, v_o)
    # offset template
    step = 37
    n_amp_views = np.array([(iv["last"] - iv["first"] + step - 1) // step for iv in ivl], dtype=np.int64)
    amp_offset = 5
    n_amp = int(amp_offset + n_amp_views.sum() + 3)
    amps = rng.standard_normal(n_amp)
    aflags = (rng.random(n_amp) < 0.1).astype(np.uint8)
    t_h = c["tod"].copy()
    t_o = c["tod"].copy()
    hip.template_offset_add_to_signal(step, amp_offset, n_amp_views, amps, aflags, 1, t_h, ivl, False)
    oracle.template_offset_add_to_signal(step, amp_offset, n_amp_views, amps, aflags, 1, t_o, ivl)
    assert np.array_equal(t_h, t_o)
    for fidx in (-1, 2):
        a_h = amps.copy()
        a_o = amps.copy()
        hip.template_offset_project_signal(1, c["tod"], fidx, c["det_flags"], 1, step, amp_offset,
                                           n_amp_views, a_h, aflags, ivl, False)
        oracle.template_offset_project_signal(1, c["tod"], fidx, c["det_flags"], 1, step, amp_offset,
                                              n_amp_views, a_o, aflags, ivl)
        np.testing.assert_allclose(a_h, a_o, rtol=1e-12, atol=1e-12)
    var = rng.random(n_amp)
    o_h = np.full(n_amp, 9.0)
    o_o = np.full(n_amp, 9.0)
    hip.template_offset_apply_diag_precond(var, amps, aflags, o_h, False)
    oracle.template_offset_apply_diag_precond(var, amps, aflags, o_o)
    assert np.array_equal(o_h, o_o)


def test_error_paths(hip):
    c = cases.make_case(n_samp=100)
    bad = c["intervals"].copy()
    bad["last"][0] = 1000  # beyond n_samp
    t = c["tod"].copy()
    with pytest.raises(RuntimeError):
        hip.noise_weight(t, c["data_index"], bad, c["det_scale"], False)
    with pytest.raises(RuntimeError):
        hip.noise_weight(t.astype(np.float32), c["data_index"], c["intervals"], c["det_scale"], False)
    with pytest.raises(RuntimeError):
        hip.noise_weight(t, c["data_index"].astype(np.int64), c["intervals"], c["det_scale"], False)


@pytest.mark.parametrize("name", ["split_gap_extra_hwp", "ragged"])
def test_pybind_module_matches_oracle(oracle, name):
    """The pybind11 host layer (toast._libtoast names/signatures) over the same C ABI."""
    import toast_amd

    m = toast_amd.load_native()
    assert m.accel_enabled()
    m.accel_assign_device(1, 0, 1.0, False)
    c = cases.make_case(**CASES[name])
    got = cases.run_chain(m, c, nest=True, tail=(False,))
    want = cases.run_chain(oracle, c, nest=True)
    compare_chain(got, want)
    with pytest.raises(RuntimeError, match="dimensions instead of"):
        m.noise_weight(np.ones(8), c["data_index"], c["intervals"], c["det_scale"], False)
    with pytest.raises(RuntimeError, match="instead of"):
        m.noise_weight(np.ones((4, 8), np.float32), c["data_index"], c["intervals"], c["det_scale"], False)


def test_combine_flags_matches_numpy(hip):
    """toast_hip_combine_flags_dev (MapMaker solver flags, mapmaker_templates.py:764-810)."""
    import torch

    c = cases.make_case(n_det=5, n_samp=4001, n_split=3, gap=9, extra_rows=2)
    n_samp, rows = c["n_samp"], c["rows"]
    dfl = torch.from_numpy(c["det_flags"]).cuda()
    sfl = torch.from_numpy(c["shared_flags"]).cuda()
    for outside in (1, 0, -1):
        out = torch.full((rows, n_samp), 7, dtype=torch.uint8, device="cuda")
        hip.dev.combine_flags(out.data_ptr(), c["data_index"], dfl.data_ptr(), n_samp, c["flag_index"], 4,
                              sfl.data_ptr(), n_samp, 2, n_samp, c["intervals"], n_out_rows=rows,
                              outside_value=outside)
        torch.cuda.synchronize()
        want = np.full((rows, n_samp), 7 if outside < 0 else outside, dtype=np.uint8)
        for iv in c["intervals"]:
            sl = slice(int(iv["first"]), int(iv["last"]))
            for o, f in zip(c["data_index"], c["flag_index"]):
                want[o, sl] = ((c["det_flags"][f, sl] & 4) != 0) | ((c["shared_flags"][sl] & 2) != 0)
        assert np.array_equal(out.cpu().numpy(), want)
    # absent optional inputs
    out = torch.zeros((rows, n_samp), dtype=torch.uint8, device="cuda")
    hip.dev.combine_flags(out.data_ptr(), c["data_index"], 0, 0, c["flag_index"], 4, 0, 0, 2, n_samp, c["intervals"],
                          n_out_rows=rows, outside_value=1)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    inside = np.zeros(n_samp, dtype=bool)
    for iv in c["intervals"]:
        inside[int(iv["first"]):int(iv["last"])] = True
    assert np.all(got[c["data_index"]][:, inside] == 0) and np.all(got[:, ~inside] == 1)
