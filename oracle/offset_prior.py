"""NumPy/SciPy restatement of the Offset template's amplitude-domain noise prior and its
preconditioners (TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg are the only importers).

Follows src/toast/templates/offset/offset.py of the reference:

* offset_psd            :625-711   (`_get_offset_psd`: Keihanen et al. 2010 eq. 22-24 with the
                                    reference's algebra correction, m_max = 5)
* remove_white_noise    :590-623   (`_remove_white_noise`)
* interpolate_psd       :547-561   (`_interpolate_psd`)
* truncate              :563-571   (`_truncate`)
* prior_freq            :200-223   (log-spaced frequency grid of the prior)
* view_filter           :455-476   (Fourier filter -> real space, truncated)
* toeplitz_preconditioner / banded_preconditioner   :492-566
* add_prior             :884-960
* apply_precond         :963-1005

Parity status: the `toast` package cannot be imported here (astropy and the compiled _libtoast
are absent), but the four helper methods are pure NumPy/SciPy, so tests/golden/
make_golden_offset_prior.py extracts exactly those method bodies from the reference file in
place (ast, no copy) and runs them: tests/golden/offset_prior.npz holds their inputs and
outputs, and tests/test_oracle_offset_prior.py pins this restatement against them.  The filter
application is scipy.signal.convolve / scipy.linalg.cho_solve_banded, the same third-party calls
the reference makes (GPU parity against them is a tolerance, 1e-11 relative).
"""

import numpy as np
import scipy.linalg
import scipy.optimize
import scipy.signal


def interpolate_psd(x, lfreq, lpsd):
    thresh = 1.0e-10  # offset.py:549
    x = np.asarray(x, dtype=np.float64)
    lowf = np.abs(x) < thresh
    good = np.logical_not(lowf)
    logx = np.empty_like(x)
    logx[lowf] = np.log(thresh)
    logx[good] = np.log(np.abs(x[good]))
    return np.exp(np.interp(logx, lfreq, lpsd))  # :559-561


def truncate(noisefilter, lim=1e-4):
    icenter = noisefilter.size // 2  # :564
    ind = np.abs(noisefilter[:icenter]) > np.abs(noisefilter[0]) * lim
    icut = np.argwhere(ind)[-1][0]
    if icut % 2 == 0:
        icut += 1
    noisefilter = np.roll(noisefilter, icenter)
    return noisefilter[icenter - icut:icenter + icut + 1]  # :570


def remove_white_noise(freq, psd):
    corrpsd = psd.copy()  # :592
    n_corrpsd = len(corrpsd)
    plat_off = int(0.8 * n_corrpsd)
    if n_corrpsd - plat_off < 10:
        plat_off = 0 if n_corrpsd < 10 else n_corrpsd - 10
    cfreq = np.log(freq[plat_off:])
    cdata = np.log(corrpsd[plat_off:])

    def lin_func(x, a, b, c):
        return a * (x - b) + c

    params, _ = scipy.optimize.curve_fit(lin_func, cfreq, cdata, p0=[0.0, cfreq[-1], cdata[-1]])  # :609-611
    plat = np.exp(lin_func(cfreq, params[0], params[1], params[2]))[-1]
    corrmax = np.amax(corrpsd)
    corrthresh = 1.0e-10 * corrmax - plat  # :620
    corrpsd -= plat
    corrpsd[corrpsd < corrthresh] = corrthresh
    return corrpsd


def offset_psd(psdfreq, psd, freq, step_time):
    psd = remove_white_noise(psdfreq, psd)  # :631
    logfreq = np.log(psdfreq)
    logpsd = np.log(psd)
    m_max = 5  # :650
    tbase = step_time
    fbase = 1.0 / tbase

    def g(f, m):
        x = np.pi * tbase * (f + m * fbase)  # :657
        bad = np.abs(x) < 1.0e-30
        good = np.logical_not(bad)
        result = np.empty_like(x)
        result[bad] = 1.0
        result[good] = (np.sin(x[good]) / x[good]) ** 2
        return result

    out = interpolate_psd(freq, logfreq, logpsd) * g(freq, 0)  # :668
    for m in range(1, m_max):
        out[:] += interpolate_psd(freq + m * fbase, logfreq, logpsd) * g(freq, m)
        out[:] += interpolate_psd(freq - m * fbase, logfreq, logpsd) * g(freq, -m)
    out *= fbase  # :709
    return out


def prior_freq(obstime, step_time, rate):
    """offset.py:200-223; None when the observation holds a single baseline."""
    fbase = 1.0 / step_time
    if (obstime * fbase) < 1.0:
        return None
    powmin = np.floor(np.log10(1 / obstime)) - 1
    powmax = min(np.ceil(np.log10(1 / step_time)) + 2, np.log10(rate))
    return np.logspace(powmin, powmax, 1000)


def filter_length(n_amp_view):
    filterlen = 2  # :457-459
    while filterlen < 2 * n_amp_view:
        filterlen *= 2
    return filterlen


def view_filter(freq, opsd, n_amp_view, step_time):
    """Real-space inverse amplitude covariance of one view (offset.py:455-476)."""
    filterfreq = np.fft.rfftfreq(filter_length(n_amp_view), step_time)
    fourierfilter = interpolate_psd(filterfreq, np.log(freq), np.log(1.0 / opsd))
    return truncate(np.fft.irfft(fourierfilter))


def toeplitz_preconditioner(freq, opsd, n_amp_view, step_time, detnoise):
    """precond_width == 1 (offset.py:498-511)."""
    filterfreq = np.fft.rfftfreq(filter_length(n_amp_view), step_time)
    pre = truncate(np.fft.irfft(interpolate_psd(filterfreq, np.log(freq), np.log(opsd))))
    if detnoise != 0:
        pre[pre.size // 2] += 1.0 / detnoise
    return pre


def banded_preconditioner(noisefilter, offsetvar, width, detnoise):
    """Lower banded Cholesky factor of (diag(1/offsetvar) + Toeplitz(noisefilter)), with the
    reference's width doubling on failure (offset.py:522-566).  Returns (factor, lower)."""
    n_amp_view = offsetvar.size
    icenter = noisefilter.size // 2
    try_width = width
    while True:
        wband = min(try_width, icenter)
        precond_width = max(wband, min(try_width, n_amp_view))
        pre = np.zeros([precond_width, n_amp_view], dtype=np.float64)
        if detnoise != 0:
            with np.errstate(divide="ignore"):
                pre[0, :] = 1.0 / offsetvar
        pre[:wband, :] += np.repeat(noisefilter[icenter:icenter + wband, np.newaxis], n_amp_view, 1)
        try:
            return scipy.linalg.cholesky_banded(pre, overwrite_ab=True, lower=True, check_finite=True), True
        except scipy.linalg.LinAlgError:
            if try_width < icenter and try_width < n_amp_view:
                try_width *= 2
            else:
                raise RuntimeError("cholesky_banded failed at the maximum width")


def add_prior(segments, filters, amps_in, amp_flags, amps_out):
    """offset.py:884-960.  ``segments`` = [(first amplitude, n_amp_view)], one filter each."""
    for (off, n), filt in zip(segments, filters):
        sl = slice(off, off + n)
        amps_out[sl] += scipy.signal.convolve(amps_in[sl], filt, mode="same", method="auto")
        amps_out[sl][amp_flags[sl] != 0] = 0.0


def apply_precond(segments, preconds, precond_width, amps_in, amp_flags, amps_out):
    """offset.py:963-1005.  ``preconds`` = Toeplitz rows (width <= 1) or (factor, lower) pairs."""
    for (off, n), pre in zip(segments, preconds):
        sl = slice(off, off + n)
        if precond_width <= 1:
            out = scipy.signal.convolve(amps_in[sl], pre, mode="same", method="auto")
        else:
            out = scipy.linalg.cho_solve_banded(pre, amps_in[sl], overwrite_b=False, check_finite=True)
        out[amp_flags[sl] != 0] = 0.0
        amps_out[sl] = out
