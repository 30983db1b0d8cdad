"""CPU: pin oracle/offset_prior.py against tests/golden/offset_prior.npz (outputs of the
reference's own helper methods, tests/golden/make_golden_offset_prior.py), and check the
product's one-off host construction (toast_amd/templates/offset_prior.py) against the oracle."""
import os

import numpy as np
import pytest
import scipy.linalg

from oracle import offset_prior as OP

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "offset_prior.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def cases(gold):
    for ic, c in enumerate(gold["cases"]):
        rate, fmin, fknee, alpha, net, obstime, step, n_amp = c
        yield ic, rate, obstime, step, int(n_amp)


def test_oracle_matches_reference_helpers_bit_exactly(gold):
    for ic, rate, obstime, step, n_amp in cases(gold):
        f, p = gold[f"c{ic}_psdfreq"], gold[f"c{ic}_psd"]
        assert np.array_equal(OP.remove_white_noise(f, p), gold[f"c{ic}_corrpsd"])
        freq = OP.prior_freq(obstime, step, rate)
        assert np.array_equal(freq, gold[f"c{ic}_freq"])
        opsd = OP.offset_psd(f, p, freq, step)
        assert np.array_equal(opsd, gold[f"c{ic}_offset_psd"])
        assert np.array_equal(OP.view_filter(freq, opsd, n_amp, step), gold[f"c{ic}_noisefilter"])
        toe = OP.toeplitz_preconditioner(freq, opsd, n_amp, step, 0.0)
        assert np.array_equal(toe, gold[f"c{ic}_toeplitz"])
    assert np.array_equal(OP.interpolate_psd(gold["interp_x"], gold["interp_lf"], gold["interp_lp"]),
                          gold["interp_out"])
    assert OP.prior_freq(5.0, 10.0, 100.0) is None  # a single baseline: no prior (offset.py:212-218)


def test_filter_is_symmetric_odd_and_positive_definite(gold):
    for ic, rate, obstime, step, n_amp in cases(gold):
        filt = gold[f"c{ic}_noisefilter"]
        assert filt.size % 2 == 1 and np.allclose(filt, filt[::-1], rtol=0, atol=1e-12 * np.max(np.abs(filt)))
        var = np.full(n_amp, 1.0 / 20.0)
        cb, lower = OP.banded_preconditioner(filt, var, 20, 0.1)
        assert lower and cb.shape[1] == n_amp and np.all(cb[0] > 0)


def test_oracle_add_prior_and_precond_are_consistent():
    """apply_precond inverts (diag + truncated Toeplitz); add_prior is the full Toeplitz product."""
    rng = np.random.default_rng(1)
    n = 200
    lags = np.arange(-43, 44)
    filt = np.exp(-np.abs(lags) / 6.0) * np.cos(lags / 9.0)
    x = rng.standard_normal(n)
    flags = np.zeros(n, dtype=np.uint8)
    out = np.zeros(n)
    OP.add_prior([(0, n)], [filt], x, flags, out)
    dense = scipy.linalg.toeplitz(np.concatenate([filt[43:], np.zeros(n - 44)]))
    assert np.allclose(out, dense @ x, rtol=1e-12, atol=1e-12)
    var = 1.0 / (5.0 + rng.random(n))
    cb = OP.banded_preconditioner(filt, var, 20, 0.1)
    y = np.zeros(n)
    OP.apply_precond([(0, n)], [cb], 20, x, flags, y)
    band = scipy.linalg.toeplitz(np.concatenate([filt[43:63], np.zeros(n - 20)])) + np.diag(1.0 / var)
    assert np.allclose(band @ y, x, rtol=1e-10, atol=1e-10)


def test_product_construction_matches_oracle(gold):
    from toast_amd.templates.offset_prior import OffsetPrior, baseline_psd, prior_frequencies

    for ic, rate, obstime, step, n_amp in cases(gold):
        f, p = gold[f"c{ic}_psdfreq"], gold[f"c{ic}_psd"]
        freq = prior_frequencies(obstime, step, rate)
        assert np.array_equal(freq, gold[f"c{ic}_freq"])
        opsd = baseline_psd(f, p, freq, step)
        assert np.max(np.abs(opsd / gold[f"c{ic}_offset_psd"] - 1.0)) < 1e-13
        n_small = min(n_amp, 400)
        rng = np.random.default_rng(ic)
        var = 1.0 / (0.1 * (50 + rng.integers(0, 50, size=2 * n_small)))
        var[7] = 0.0  # a flagged amplitude (the reference would fail on 1 / 0 here)
        segs = [dict(first=0, n_amp=n_small, freq=freq, psdfreq=f, psd=p, detnoise=0.1),
                dict(first=n_small, n_amp=n_small, freq=freq, psdfreq=f, psd=p, detnoise=0.1)]
        for width in (1, 20):
            prior = OffsetPrior("t", width, factor_on_device=False).build(segs, var, step)
            assert list(prior.seg_start) == [0, n_small, 2 * n_small]
            want = OP.view_filter(freq, gold[f"c{ic}_offset_psd"], n_small, step)
            for filt in prior.filters:
                assert filt.shape == want.shape and np.max(np.abs(filt - want)) < 1e-12 * np.max(np.abs(want))
            if width == 1:
                toe = OP.toeplitz_preconditioner(freq, gold[f"c{ic}_offset_psd"], n_small, step, 0.1)
                assert np.max(np.abs(prior.precond[0] - toe)) < 1e-12 * np.max(np.abs(toe))
            else:
                cb, _ = OP.banded_preconditioner(want, var[n_small:], width, 0.1)  # second segment: no zero variance
                assert prior.precond[1].shape == cb.shape
                assert np.max(np.abs(prior.precond[1] - cb)) < 1e-11 * np.max(np.abs(cb))
                assert np.all(np.isfinite(prior.precond[0]))
