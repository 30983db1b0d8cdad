"""CPU: the oracle restatement against the reference's own outputs (golden fixtures), the
reference's known-answer values, and -- when oracle/_ref is present -- the live reference."""
import os

import numpy as np
import pytest

import cases
import golden_util as gu


@pytest.mark.parametrize("name", gu.CHAINS)
def test_oracle_matches_reference_fixture(oracle, name):
    got, want = gu.check_chain(oracle, name, weights_rtol=0, ztol=0)  # same order, same libm: exact
    assert np.array_equal(got["zmap"], want["zmap"])
    assert np.array_equal(got["tod"], want["tod"])


def test_oracle_healpix_known_answers(oracle):
    z = gu.load("healpix_kat")
    theta, phi, vec = z["theta"], z["phi"], z["vec"]
    for nside in (1, 256, 16384):
        assert np.array_equal(oracle.healpix_ang2pix(nside, theta, phi, nest=True), z["ang2nest_%d" % nside])
        assert np.array_equal(oracle.healpix_ang2pix(nside, theta, phi, nest=False), z["ang2ring_%d" % nside])
        assert np.array_equal(oracle.healpix_vec2pix(nside, vec, nest=True), z["vec2nest_%d" % nside])
        assert np.array_equal(oracle.healpix_vec2pix(nside, vec, nest=False), z["vec2ring_%d" % nside])
        assert np.array_equal(oracle.healpix_ring2nest(nside, z["ang2ring_%d" % nside]), z["ring2nest_%d" % nside])
        assert np.array_equal(oracle.healpix_nest2ring(nside, z["ring2nest_%d" % nside]), z["nest2ring_%d" % nside])
        # round trip (reference test src/toast/tests/healpix.py:120-150)
        assert np.array_equal(z["nest2ring_%d" % nside], z["ang2ring_%d" % nside])


def test_survey_known_answer_vectors(oracle):
    """Values obtained from the reference during the survey."""
    th = np.array([0.0, np.pi / 2, np.pi, 1e-9, np.pi / 2 + 1e-16])
    ph = np.array([0.0, 0.0, 0.0, 2 * np.pi, np.pi])
    want = {
        1: ([0, 4, 8, 0, 6], [0, 4, 8, 0, 6]),
        64: ([4095, 19456, 32768, 4095, 26282], [0, 24192, 49148, 0, 24576]),
        1024: ([1048575, 4980736, 8388608, 1048575, 6728362], [0, 6285312, 12582908, 0, 6291456]),
    }
    for nside, (nest, ring) in want.items():
        assert list(oracle.healpix_ang2pix(nside, th, ph, nest=True)) == nest
        assert list(oracle.healpix_ang2pix(nside, th, ph, nest=False)) == ring
    # regression quaternion, src/toast/tests/ops_pixels_healpix.py:35-42
    z = gu.load("healpix_kat")
    q = z["regress_quat"]
    iv = np.zeros(1, cases.interval_dtype)
    iv["last"] = 1
    for nest, val, sub in ((True, 143138818, 46594), (False, 187529588, 61044)):
        pix = np.zeros((1, 1), np.int64)
        hs = np.zeros(12 * 4096 * 4096 // 3072, np.uint8)
        oracle.pixels_healpix(np.zeros(1, np.int32), q, np.zeros(1, np.uint8), 0, np.zeros(1, np.int32), pix, iv,
                              hs, 3072, 4096, nest)
        assert pix[0, 0] == val and pix[0, 0] < 12 * 4096 * 4096
        assert list(np.flatnonzero(hs)) == [sub]
        assert pix[0, 0] == z["regress_quat_%s" % ("nest" if nest else "ring")][0, 0]


def test_oracle_offset_template_fixture(oracle):
    z = gu.load("offset_template")
    ivl = z["intervals"].astype(cases.interval_dtype)
    step, off = int(z["step"]), int(z["amp_offset"])
    t = z["tod"].copy()
    oracle.template_offset_add_to_signal(step, off, z["n_amp_views"], z["amps"], z["aflags"], 1, t, ivl)
    assert np.array_equal(t, z["out_add"])
    for fidx, key in ((-1, "out_proj_noflag"), (0, "out_proj_flag")):
        a = z["amps"].copy()
        oracle.template_offset_project_signal(1, z["tod"], fidx, z["det_flags"], 1, step, off, z["n_amp_views"], a,
                                              z["aflags"], ivl)
        np.testing.assert_allclose(a, z[key], rtol=1e-14, atol=1e-14)
    o = np.full(z["amps"].size, 7.0)
    oracle.template_offset_apply_diag_precond(z["var"], z["amps"], z["aflags"], o)
    assert np.array_equal(o, z["out_precond"])
    w = np.zeros((2, 900))
    oracle.stokes_weights_I(np.arange(2, dtype=np.int32), w, ivl, z["cal"])
    assert np.array_equal(w, z["out_stokes_I"])


def test_oracle_cov_apply_diag_semantics(oracle):
    """vec <- Sym(packed upper triangle) . vec, against dense numpy (toast_map_cov.cpp:471-528)."""
    rng = np.random.default_rng(0)
    for nnz in (1, 2, 3):
        nsub, subsize = 3, 17
        blk = nnz * (nnz + 1) // 2
        mat = rng.standard_normal((nsub, subsize, blk))
        vec = rng.standard_normal((nsub, subsize, nnz))
        want = np.empty_like(vec)
        iu = np.triu_indices(nnz)
        for i in range(nsub):
            for j in range(subsize):
                m = np.zeros((nnz, nnz))
                m[iu] = mat[i, j]
                m = m + m.T - np.diag(np.diag(m))
                want[i, j] = m @ vec[i, j]
        oracle.cov_apply_diag(nsub, subsize, nnz, mat, vec)
        np.testing.assert_allclose(vec, want, rtol=1e-13, atol=1e-13)


def test_oracle_cov_apply_diag_matches_reference_fixture(oracle):
    """tests/golden/cov_filter.npz: outputs of the reference's own cov_apply_diag (its no-LAPACK
    configuration builds in place, tests/golden/make_golden_cov.py): bit-exact."""
    z = np.load(os.path.join(gu.GOLDEN, "cov_filter.npz"))
    for nnz in (1, 2, 3):
        mat, vec = z[f"apply{nnz}_mat"], z[f"apply{nnz}_vec"].copy()
        oracle.cov_apply_diag(mat.shape[0], mat.shape[1], nnz, mat, vec)
        assert np.array_equal(vec, z[f"apply{nnz}_out"])


@pytest.mark.parametrize("name", ["chain_a", "chain_b", "chain_c"])
def test_hits_and_invcov_fixture_vs_python_loop(name):
    """The reference's cov_accum_diag_hits / cov_accum_diag_invnpp driven like BuildHitMap /
    BuildInverseCovariance (fixture) against the plain per-sample definition."""
    z = np.load(os.path.join(gu.GOLDEN, name + ".npz"))
    g = np.load(os.path.join(gu.GOLDEN, "cov_filter.npz"))
    hits, invcov = cases.python_hits_invcov(z)
    assert np.array_equal(hits, g[name + "_hits"])
    np.testing.assert_allclose(invcov, g[name + "_invcov"], rtol=1e-13, atol=1e-13 * np.max(np.abs(invcov)))


def test_cov_apply_diag_live_reference(oracle, ref):
    rng = np.random.default_rng(4)
    for nnz in (1, 2, 3):
        mat = rng.standard_normal((4, 33, nnz * (nnz + 1) // 2))
        a = rng.standard_normal((4, 33, nnz))
        b = a.copy()
        ref.cov_apply_diag(4, 33, nnz, mat.reshape(-1), a.reshape(-1))
        oracle.cov_apply_diag(4, 33, nnz, mat, b)
        assert np.array_equal(a, b)


LIVE = {
    "default": dict(),
    "split": dict(n_split=3, gap=5, extra_rows=2, with_hwp=True),
    "nside1024": dict(nside=1024, n_samp=5000, with_det_flags=False, n_det=1),
    "random4096": dict(random_pointing=True, nside=4096, n_samp=20000),
    "ragged": dict(n_samp=1029, n_split=4, gap=1, n_det=3, nside=256),
    "ground2048": dict(ground=True, n_samp=72000, rate=100.0, nside=2048, n_det=6, with_hwp=True),
    # degenerate inputs: a view without intervals, every sample flagged, one-sample intervals, odd / unpaired detectors
    "no_intervals": dict(empty_intervals=True, n_samp=500, nside=32),
    "all_flagged": dict(all_flagged=True, n_samp=700, nside=32),
    "one_sample_intervals": dict(n_samp=24, n_split=24, n_det=2, nside=8),
    "odd_dets": dict(n_det=5, n_samp=3000, nside=64, fp_roll=1),
    "tiny": dict(n_samp=3, n_det=2, nside=1, with_det_flags=False, with_shared_flags=False),
}


@pytest.mark.parametrize("name", list(LIVE))
@pytest.mark.parametrize("nest", [True, False])
def test_oracle_vs_live_reference(oracle, ref, name, nest):
    c = cases.make_case(**LIVE[name])
    a = cases.run_chain(ref, c, nest=nest, tail=(False,))
    b = cases.run_chain(oracle, c, nest=nest)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("map_dtype", [np.float32, np.int64, np.int32])
def test_oracle_scan_map_dtypes_vs_live_reference(oracle, ref, map_dtype):
    c = cases.make_case(n_samp=3000, nside=128, n_split=2)
    a = cases.run_chain(ref, c, map_dtype=map_dtype, scan_scale=0.37, tail=(False,))
    b = cases.run_chain(oracle, c, map_dtype=map_dtype, scan_scale=0.37)
    assert np.array_equal(a["tod"], b["tod"])


@pytest.mark.parametrize("nside", [1, 1 << 14, 1 << 20, 1 << 29])
def test_oracle_pixels_on_boundaries_vs_live_reference(oracle, ref, nside):
    """The pointings of tests/test_gpu_pixels_adversarial.py (|z| = 2/3, poles, face meridians within
    0 .. 1e3 ulp) through the reference's own compiled kernel and the oracle: identical indices up
    to the largest nside."""
    from test_gpu_pixels_adversarial import boundary_pointings

    q = boundary_pointings(np.random.default_rng(nside % 9973), n_each=1500)
    n = q.shape[0]
    quats = np.ascontiguousarray(q.reshape(1, n, 4))
    iv = np.zeros(1, cases.interval_dtype)
    iv["last"] = n
    npix = 12 * nside * nside
    nps = npix if nside < 16 else npix // 3072
    idx, flags = np.zeros(1, np.int32), np.zeros(1, np.uint8)
    for nest in (True, False):
        a = np.full((1, n), -7, np.int64)
        b = np.full((1, n), -9, np.int64)
        ha, hb = np.zeros(npix // nps, np.uint8), np.zeros(npix // nps, np.uint8)
        ref.pixels_healpix(idx, quats, flags, 0, idx, a, iv, ha, nps, nside, nest, False)
        oracle.pixels_healpix(idx, quats, flags, 0, idx, b, iv, hb, nps, nside, nest)
        assert np.array_equal(a, b) and np.array_equal(ha, hb)


@pytest.mark.parametrize("seed", list(range(24)))
def test_oracle_vs_live_reference_random(oracle, ref, seed):
    """The same randomly drawn chain cases as tests/test_gpu_parity.py::test_chain_random_cases: the oracle and the
    reference's own compiled kernels agree bit for bit on every product."""
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
    a = cases.run_chain(ref, c, nest=nest, iau=iau, tail=(False,))
    b = cases.run_chain(oracle, c, nest=nest, iau=iau)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_mapmaker_e2e_fixture_is_what_the_reference_chain_produces_now():
    """tests/golden/mapmaker_e2e.npz (the destriping map-maker end to end, tests/golden/make_golden_mapmaker.py) is not
    stale: the small case regenerated here -- oracle/_ref kernels in the reference's operator order, the reference's own
    solve() compiled from its syntax tree -- equals the committed arrays bit for bit.  Build container only."""
    import importlib.util

    if not os.path.exists("/root/reference/src/toast/ops/mapmaker_solve.py"):
        pytest.skip("no /root/reference on this machine")
    import oracle as o

    if o.load_ref() is None:
        pytest.skip("oracle/_ref not built")
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden_mapmaker", os.path.join(here, "golden", "make_golden_mapmaker.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    blob = gen.run_case("small", gen.load_reference_solve())
    z = np.load(os.path.join(here, "golden", "mapmaker_e2e.npz"))
    for k, v in blob.items():
        assert np.array_equal(np.asarray(v), z[k]), k
    # the solve did what the fixture says: iters + 1 applications of the left-hand side, a falling residual
    assert int(z["small_lhs_calls"]) == len(z["small_history"]) + 1
    assert z["small_history"][-1] < 1e-2 * z["small_history"][0]
