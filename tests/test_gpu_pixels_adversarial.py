"""GPU: pixels_healpix against the oracle (itself bit-identical to the reference's compiled kernel,
tests/test_oracle_golden.py) on pointings chosen to sit on the decision boundaries of the HEALPix
projection -- the equatorial / polar transition |z| = 2/3, the poles, the face meridians
phi = k pi/2 and phi = 0 / 2 pi -- within 0 .. 1e3 ulp, at every resolution from nside 1 to 2^29
(the largest the int64 pixel index allows), NEST and RING.  The bar is bit-exact indices."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from toast_amd import capi

    assert capi.accel_enabled()
    capi.accel_assign_device(1, 0, 1.0, False)
    return capi


def quats_pointing_at(theta, phi, psi):
    """q = Rz(phi) Ry(theta) Rz(psi): the z axis goes to (theta, phi)."""
    def rz(a):
        q = np.zeros((a.size, 4))
        q[:, 2], q[:, 3] = np.sin(a / 2), np.cos(a / 2)
        return q

    def ry(a):
        q = np.zeros((a.size, 4))
        q[:, 1], q[:, 3] = np.sin(a / 2), np.cos(a / 2)
        return q

    def mult(p, q):
        r = np.empty_like(p)
        r[:, 0] = p[:, 0] * q[:, 3] + p[:, 1] * q[:, 2] - p[:, 2] * q[:, 1] + p[:, 3] * q[:, 0]
        r[:, 1] = -p[:, 0] * q[:, 2] + p[:, 1] * q[:, 3] + p[:, 2] * q[:, 0] + p[:, 3] * q[:, 1]
        r[:, 2] = p[:, 0] * q[:, 1] - p[:, 1] * q[:, 0] + p[:, 2] * q[:, 3] + p[:, 3] * q[:, 2]
        r[:, 3] = -p[:, 0] * q[:, 0] - p[:, 1] * q[:, 1] - p[:, 2] * q[:, 2] + p[:, 3] * q[:, 3]
        return r

    return mult(mult(rz(phi), ry(theta)), rz(psi))


def boundary_pointings(rng, n_each=4000):
    th_edges = [np.arccos(2.0 / 3.0), np.arccos(-2.0 / 3.0), 0.0, np.pi, np.pi / 2]
    ph_edges = [k * np.pi / 2 for k in range(5)] + [k * np.pi / 4 for k in (1, 3, 5, 7)]
    thetas, phis = [], []
    for scale in (0.0, 1e-16, 1e-15, 1e-13, 1e-10, 1e-7, 1e-3):
        for te in th_edges:
            thetas.append(np.clip(te + scale * rng.standard_normal(n_each), 0.0, np.pi))
            phis.append(rng.uniform(0, 2 * np.pi, n_each))
        for pe in ph_edges:
            thetas.append(np.arccos(rng.uniform(-1, 1, n_each)))
            phis.append(pe + scale * rng.standard_normal(n_each))
        for te in th_edges[:2]:          # both boundaries at once
            for pe in ph_edges[:4]:
                thetas.append(np.clip(te + scale * rng.standard_normal(n_each // 8), 0.0, np.pi))
                phis.append(pe + scale * rng.standard_normal(n_each // 8))
    thetas.append(np.arccos(rng.uniform(-1, 1, 20 * n_each)))
    phis.append(rng.uniform(-np.pi, 3 * np.pi, 20 * n_each))
    theta, phi = np.concatenate(thetas), np.concatenate(phis)
    return quats_pointing_at(theta, phi, rng.uniform(0, 2 * np.pi, theta.size))


@pytest.mark.parametrize("nside", [1, 2, 8, 1024, 1 << 14, 1 << 20, 1 << 26, 1 << 29])
def test_boundary_pointings_bit_exact(hip, oracle, nside):
    rng = np.random.default_rng(nside % 9973)
    q = boundary_pointings(rng)
    n = q.shape[0]
    quats = np.ascontiguousarray(q.reshape(1, n, 4))
    iv = np.zeros(1, cases.interval_dtype)
    iv["last"] = n
    npix = 12 * nside * nside
    nps = npix if nside < 16 else npix // 3072      # 3072 submaps at most: the hit array stays small
    n_submap = npix // nps
    idx = np.zeros(1, np.int32)
    flags = np.zeros(1, np.uint8)
    for nest in (True, False):
        got = np.full((1, n), -7, np.int64)
        want = np.full((1, n), -9, np.int64)
        hs_g = np.zeros(n_submap, np.uint8)
        hs_w = np.zeros(n_submap, np.uint8)
        hip.pixels_healpix(idx, quats, flags, 0, idx, got, iv, hs_g, nps, nside, nest, False)
        oracle.pixels_healpix(idx, quats, flags, 0, idx, want, iv, hs_w, nps, nside, nest)
        nbad = int(np.count_nonzero(got != want))
        assert nbad == 0, f"nside {nside} nest {nest}: {nbad} of {n} indices differ"
        assert got.min() >= 0 and got.max() < npix
        assert np.array_equal(hs_g, hs_w)


@pytest.mark.parametrize("use_hwp", [False, True])
def test_stokes_weights_on_boundaries(hip, oracle, use_hwp):
    """Same boundary pointings plus exact and near poles through stokes_weights_IQU.  Wherever the
    reference formulation is finite the weights agree to 1e-13; at pointings where rounding makes
    1 - z^2 negative the reference's -sqrt(1 - z^2) is NaN (ops_stokes_weights.cpp:50-75), and so are the
    oracle and -- by default -- the device; with toast_hip_set_stokes_reference_nan(0) the device formulation (no
    square root, hpix_math.hpp stokes_cs2alpha) stays finite there with the correct modulus eta * cal."""
    rng = np.random.default_rng(3)
    q = boundary_pointings(rng, n_each=2000)
    th = np.concatenate([np.zeros(1000), np.full(1000, np.pi), rng.uniform(0, 1e-12, 1000),
                         np.pi - rng.uniform(0, 1e-12, 1000)])
    q = np.concatenate([q, quats_pointing_at(th, rng.uniform(0, 2 * np.pi, th.size), rng.uniform(0, 2 * np.pi, th.size))])
    n = q.shape[0]
    quats = np.ascontiguousarray(q.reshape(1, n, 4))
    iv = np.zeros(1, cases.interval_dtype)
    iv["last"] = n
    idx = np.zeros(1, np.int32)
    hwp = rng.uniform(0, 2 * np.pi, n) if use_hwp else np.zeros(1)
    eps, gamma, cal = np.array([0.1]), np.array([0.3]), np.array([1.7])
    got, want = np.zeros((1, n, 3)), np.zeros((1, n, 3))
    hip.stokes_weights_IQU(idx, quats, idx, got, hwp, iv, eps, gamma, cal, False, False)
    oracle.stokes_weights_IQU(idx, quats, idx, want, hwp, iv, eps, gamma, cal, False)
    # the default: the reference's result sample for sample, its NaNs included
    assert np.array_equal(np.isnan(got), np.isnan(want))
    finite = ~np.isnan(want).any(axis=2)[0]
    assert np.count_nonzero(finite) > 0.9 * n and np.count_nonzero(~finite) > 0
    assert np.max(np.abs(got[0][finite] - want[0][finite])) < 1e-13
    eta = (1 - eps[0]) / (1 + eps[0])
    assert np.array_equal(got[0, :, 0], np.full(n, cal[0]))
    # the opt-in device formulation (no square root): finite everywhere, same values where the reference is finite
    hip.set_stokes_reference_nan(False)
    try:
        got = np.zeros((1, n, 3))
        hip.stokes_weights_IQU(idx, quats, idx, got, hwp, iv, eps, gamma, cal, False, False)
    finally:
        hip.set_stokes_reference_nan(True)
    assert not np.any(np.isnan(got))
    assert np.max(np.abs(got[0][finite] - want[0][finite])) < 1e-13
    assert np.array_equal(got[0, :, 0], np.full(n, cal[0]))
    assert np.max(np.abs(np.hypot(got[0, :, 1], got[0, :, 2]) - eta * cal[0])) < 1e-13
