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
