"""GPU: the HIP library against the reference's own outputs (committed golden fixtures)."""
import numpy as np
import pytest

import cases
import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from toast_amd import capi

    assert capi.accel_enabled()
    capi.accel_assign_device(1, 0, 1.0, False)
    return capi


@pytest.mark.parametrize("name", gu.CHAINS)
def test_hip_matches_reference_fixture(hip, name):
    gu.check_chain(hip, name, tail=(False,), weights_rtol=1e-12, ztol=1e-12)


def test_hip_regression_quaternion(hip):
    z = gu.load("healpix_kat")
    iv = np.zeros(1, cases.interval_dtype)
    iv["last"] = 1
    for nest, val, sub in ((True, 143138818, 46594), (False, 187529588, 61044)):
        pix = np.zeros((1, 1), np.int64)
        hs = np.zeros(12 * 4096 * 4096 // 3072, np.uint8)
        hip.pixels_healpix(np.zeros(1, np.int32), z["regress_quat"], np.zeros(1, np.uint8), 0,
                           np.zeros(1, np.int32), pix, iv, hs, 3072, 4096, nest, False)
        assert pix[0, 0] == val
        assert list(np.flatnonzero(hs)) == [sub]


def test_hip_healpix_known_answers_via_quaternions(hip):
    """The KAT directions (incl. eps-perturbed poles / equator / meridians) pushed through the
    pixels_healpix kernel as rotations of the z axis must give the reference's vec2nest/ring."""
    z = gu.load("healpix_kat")
    vec = z["vec"]
    # quaternion rotating z to v: axis = z x v, angle = acos(v_z)
    vz = np.clip(vec[:, 2], -1, 1)
    ang = np.arccos(vz)
    ax = np.stack([-vec[:, 1], vec[:, 0], np.zeros(len(vec))], axis=1)
    nrm = np.linalg.norm(ax, axis=1)
    ok = nrm > 1e-3  # keep well-conditioned directions (the rotation itself must not add error)
    ax = ax[ok] / nrm[ok, None]
    q = np.concatenate([ax * np.sin(ang[ok] / 2)[:, None], np.cos(ang[ok] / 2)[:, None]], axis=1)
    n = q.shape[0]
    quats = np.ascontiguousarray(q.reshape(1, n, 4))
    iv = np.zeros(1, cases.interval_dtype)
    iv["last"] = n
    import oracle

    for nside in (1, 256, 16384):
        nps = 12 * nside * nside if nside < 16 else 3072
        for nest in (True, False):
            pix = np.zeros((1, n), np.int64)
            hs = np.zeros(12 * nside * nside // nps, np.uint8)
            hip.pixels_healpix(np.zeros(1, np.int32), quats, np.zeros(1, np.uint8), 0, np.zeros(1, np.int32), pix,
                               iv, hs, nps, nside, nest, False)
            want = np.zeros((1, n), np.int64)
            hs2 = np.zeros_like(hs)
            oracle.pixels_healpix(np.zeros(1, np.int32), quats, np.zeros(1, np.uint8), 0, np.zeros(1, np.int32),
                                  want, iv, hs2, nps, nside, nest)
            assert np.array_equal(pix, want)
            assert np.array_equal(hs, hs2)


def test_hip_offset_template_fixture(hip):
    z = gu.load("offset_template")
    ivl = z["intervals"].astype(cases.interval_dtype)
    step, off = int(z["step"]), int(z["amp_offset"])
    t = z["tod"].copy()
    hip.template_offset_add_to_signal(step, off, z["n_amp_views"], z["amps"], z["aflags"], 1, t, ivl, False)
    assert np.array_equal(t, z["out_add"])
    for fidx, key in ((-1, "out_proj_noflag"), (0, "out_proj_flag")):
        a = z["amps"].copy()
        hip.template_offset_project_signal(1, z["tod"], fidx, z["det_flags"], 1, step, off, z["n_amp_views"], a,
                                           z["aflags"], ivl, False)
        np.testing.assert_allclose(a, z[key], rtol=1e-12, atol=1e-12)
    o = np.full(z["amps"].size, 7.0)
    hip.template_offset_apply_diag_precond(z["var"], z["amps"], z["aflags"], o, False)
    assert np.array_equal(o, z["out_precond"])
    w = np.zeros((2, 900))
    hip.stokes_weights_I(np.arange(2, dtype=np.int32), w, ivl, z["cal"], False)
    assert np.array_equal(w, z["out_stokes_I"])


@pytest.mark.parametrize("name", gu.CHAINS)
def test_on_the_fly_kernels_match_reference_fixture(hip, name):
    """The pointing-on-the-fly kernels against the REFERENCE's own outputs (the committed golden
    chains were produced by oracle/_ref): pixels and hit submaps bit for bit, weights 1e-12, zmap
    and scanned + noise-weighted TOD 1e-12 / 1e-11."""
    import torch

    case, want, nest, iau = gu.load_chain(name)
    c = case
    n_samp, rows = c["n_samp"], c["rows"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    bore, sfl, hwp = dev(c["boresight"]), dev(c["shared_flags"]), dev(c["hwp"])
    kw = dict(d_shared_flags=sfl.data_ptr(), n_shared_flags=c["shared_flags"].size, shared_flag_mask=1,
              d_hwp=hwp.data_ptr(), n_hwp=c["hwp"].size, epsilon=c["epsilon"], gamma=c["gamma"], cal=c["cal"], IAU=iau)
    pt = hip.otf_pointing(bore.data_ptr(), c["focalplane"], c["nside"], nest, 3, **kw)
    D = hip.dev
    pix = torch.full((rows, n_samp), -7, dtype=torch.int64, device="cuda")
    hs = torch.zeros(c["n_submap"], dtype=torch.uint8, device="cuda")
    w = torch.zeros((rows, n_samp, 3), dtype=torch.float64, device="cuda")
    D.otf_pixels_healpix(pt, c["pixel_index"], pix.data_ptr(), n_samp, c["intervals"], hs.data_ptr(), c["n_submap"],
                         c["n_pix_submap"])
    D.otf_stokes_weights(pt, c["weight_index"], w.data_ptr(), n_samp, c["intervals"])
    torch.cuda.synchronize()
    assert np.array_equal(pix.cpu().numpy(), want["pixels"])
    assert np.array_equal(hs.cpu().numpy(), want["hsub"])
    np.testing.assert_allclose(w.cpu().numpy(), want["weights"], rtol=1e-12, atol=1e-15)
    g2l = dev(want["g2l"])
    tod, dfl = dev(c["tod"]), dev(c["det_flags"])
    n_flag = n_samp if c["det_flags"].shape[-1] == n_samp else 0
    z = torch.zeros(want["zmap"].shape, dtype=torch.float64, device="cuda")
    D.otf_build_noise_weighted(pt, g2l.data_ptr(), z.data_ptr(), c["n_pix_submap"], c["data_index"], tod.data_ptr(),
                               c["flag_index"], dfl.data_ptr(), n_flag, c["det_scale"], 1, n_samp, c["intervals"],
                               sfl.data_ptr(), c["shared_flags"].size, 1)
    zin = dev(want["zmap"])
    t2 = dev(c["tod"])
    D.otf_scan_map(pt, g2l.data_ptr(), zin.data_ptr(), c["n_pix_submap"], t2.data_ptr(), c["data_index"], n_samp,
                   c["intervals"], 1.0, False, True, det_weights=c["det_scale"])
    torch.cuda.synchronize()
    zs = np.max(np.abs(want["zmap"]))
    assert np.max(np.abs(z.cpu().numpy() - want["zmap"])) <= 1e-12 * zs
    ts = np.max(np.abs(want["tod"]))
    assert np.max(np.abs(t2.cpu().numpy() - want["tod"])) <= 1e-11 * ts


@pytest.mark.parametrize("name", gu.CHAINS)
def test_hip_hits_and_invcov_match_reference_fixture(name):
    """build_hit_map / build_inverse_covariance (toast_hip_build_cov) against the reference's own
    cov_accum_diag_hits / cov_accum_diag_invnpp outputs (tests/golden/cov_filter.npz): hits exact,
    inverse covariance 1e-12 of its largest element (summation order)."""
    import toast_amd

    m = toast_amd.load_native()
    m.accel_assign_device(1, 0, 1.0, False)
    z = gu.load(name)
    g = gu.load("cov_filter")
    want_hits, want_invcov = g[name + "_hits"], g[name + "_invcov"]
    ivl = z["in_intervals"].astype(cases.interval_dtype)
    pixels = np.ascontiguousarray(z["out_pixels"])
    weights = np.ascontiguousarray(z["out_weights"])
    g2l = np.ascontiguousarray(z["out_g2l"])
    dflags = np.ascontiguousarray(z["in_det_flags"])
    sflags = np.ascontiguousarray(z["in_shared_flags"])
    hits = np.zeros_like(want_hits)
    m.build_hit_map(g2l, hits, z["in_pixel_index"], pixels, z["in_flag_index"], dflags, 1, ivl, sflags, 1, False)
    assert np.array_equal(hits, want_hits)
    invcov = np.zeros_like(want_invcov)
    m.build_inverse_covariance(g2l, invcov, z["in_pixel_index"], pixels, z["in_weight_index"], weights,
                               z["in_flag_index"], dflags, np.ascontiguousarray(z["in_det_scale"]), 1, ivl, sflags, 1,
                               False)
    assert np.max(np.abs(invcov - want_invcov)) <= 1e-12 * np.max(np.abs(want_invcov))


def test_hip_cov_apply_diag_matches_reference_fixture(hip):
    g = gu.load("cov_filter")
    for nnz in (1, 2, 3):
        mat, vec = np.ascontiguousarray(g[f"apply{nnz}_mat"]), g[f"apply{nnz}_vec"].copy()
        hip.cov_apply_diag(mat.shape[0], mat.shape[1], nnz, mat, vec, False)
        assert np.array_equal(vec, g[f"apply{nnz}_out"])
