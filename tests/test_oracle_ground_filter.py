"""CPU: pin oracle/ground_filter.py against the outputs of the reference's own tod_filter kernels
(tests/golden/cov_filter.npz, produced by tests/golden/make_golden_cov.py from oracle/_ref)."""
import numpy as np
import pytest

import golden_util as gu
from oracle import ground_filter as GF


@pytest.fixture(scope="module")
def gold():
    return gu.load("cov_filter")


def test_legendre_templates_bit_exact(gold):
    assert np.array_equal(GF.legendre_templates(gold["gf_x"], 1, 4), gold["gf_trend"])
    assert np.array_equal(GF.legendre_templates(gold["gf_phase"], 0, 6), gold["gf_poly"])


def test_projection_gram_fit_and_subtraction(gold):
    templates = np.vstack([gold["gf_trend"], gold["gf_poly"]])
    for d in range(2):
        sig, good = gold["gf_signal"][d], gold["gf_good"][d]
        proj = GF.bin_proj(sig, templates, good)
        invcov = GF.bin_invcov(templates, good)
        assert np.max(np.abs(proj - gold[f"gf_proj{d}"])) < 1e-13 * np.max(np.abs(proj))
        assert np.max(np.abs(invcov - gold[f"gf_invcov{d}"])) < 1e-13 * np.max(np.abs(invcov))
        fit = np.zeros(sig.size)
        GF.add_templates(fit, templates[3:], gold[f"gf_coeff{d}"][3:])
        assert np.array_equal(fit, gold[f"gf_fit{d}"])
        coeff, rcond = GF.fit_templates(templates, sig, good.astype(bool))
        assert rcond > 1e-6
        assert np.max(np.abs(coeff - gold[f"gf_coeff{d}"])) < 1e-10 * np.max(np.abs(coeff))
        # the injected templates are recovered: 3.0 x poly order 2 and -1.5 x trend order 1
        assert abs(coeff[3 + 2] - 3.0) < 0.1 and abs(coeff[0] + 1.5) < 0.1


def test_live_reference_kernels(ref):
    rng = np.random.default_rng(8)
    n = 2500
    x = np.sort(rng.uniform(-1, 1, n))
    for start, stop in ((0, 1), (0, 2), (1, 2), (1, 6), (0, 9), (3, 7)):
        want = np.zeros((stop - start, n))
        ref.legendre_templates(x, want, start, stop)
        assert np.array_equal(GF.legendre_templates(x, start, stop), want)
    t = GF.legendre_templates(x, 0, 5)
    sig = rng.standard_normal(n)
    good = (rng.random(n) > 0.2).astype(np.uint8)
    proj = np.zeros(5)
    inv = np.zeros((5, 5))
    ref.bin_proj(sig, t, good, proj)
    ref.bin_invcov(t, good, inv)
    assert np.allclose(GF.bin_proj(sig, t, good), proj, rtol=1e-12, atol=1e-12)
    assert np.allclose(GF.bin_invcov(t, good), inv, rtol=1e-12, atol=1e-12)
    a, b = sig.copy(), sig.copy()
    c = rng.standard_normal(5)
    ref.add_templates(a, t, c)
    GF.add_templates(b, t, c)
    assert np.array_equal(a, b)


def test_build_templates_shapes_and_split():
    n = 4000
    t = np.arange(n) / 50.0
    az = 0.7 + 0.3 * (2.0 * np.abs((t / 11.0) % 1.0 - 0.5))
    going_right = ((t / 11.0) % 1.0) >= 0.5
    lr, rl = going_right, ~going_right
    plain = GF.build_templates(n, az, 3, 4)
    assert plain.shape == (3 + 5, n)
    split = GF.build_templates(n, az, 3, 4, split=True, lr_mask=lr, rl_mask=rl)
    assert split.shape == (3 + 10, n)
    assert np.array_equal(split[3][lr], np.zeros(lr.sum())) and np.array_equal(split[3][rl], plain[3][rl])
    assert np.array_equal(split[4][rl], np.zeros(rl.sum())) and np.array_equal(split[4][lr], plain[3][lr])
    binned = GF.build_templates(n, az, None, None, bin_width=0.05)
    assert np.array_equal(binned.sum(axis=0), np.ones(n))   # every sample falls in exactly one bin
    both = GF.build_templates(n, az, 2, 2, bin_width=0.05)
    assert both.shape[0] == 2 + 3 + binned.shape[0] - 1    # one bin dropped against the polynomial
