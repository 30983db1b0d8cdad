"""NumPy/SciPy restatement of the reference's FFT noise weighting (TEST INFRASTRUCTURE ONLY).

Parity status: PINNED.  The `toast` package cannot be imported here (astropy, traitlets, the
compiled _libtoast are absent), but the reference's `AlgorithmBase` / `AlgorithmNumpy` classes
(src/toast/fft.py:121-350) are pure NumPy / SciPy: tests/golden/make_golden_fft.py compiles exactly
those class definitions from the reference file in place and runs them; this restatement
reproduces their outputs bit for bit (tests/golden/fft_convolve.npz,
tests/test_fft_oracle.py::test_oracle_matches_reference_convolve_fixture).  The reference's own
test of the path -- the two-tone low-pass of src/toast/tests/fft.py:151-237 (|diff| < 0.2) -- is
re-run as well.  Line by line:

* convolve / AlgorithmNumpy   src/toast/fft.py:163-212, 252-350, 700-945
* extend_flags                src/toast/utils.py:1055-1113
* NoiseFilter kernels         src/toast/ops/noise_filter.py:130-171
* half-complex layout         src/libtoast/src/toast_math_fft_fftw.cpp:26-128

The arithmetic below the Python level (pocketfft in NumPy, SciPy's PCHIP) is third-party and
not part of /root/reference: GPU parity against this file is a tolerance (1e-12 relative).
"""

import numpy as np
from scipy.interpolate import PchipInterpolator
from scipy.signal import windows


def fft_length(n_samp):
    order = int(np.ceil(np.log(n_samp) / np.log(2)))  # fft.py:278-279
    return 2 ** (order + 1)


def apodization(n_reflect):
    # fft.py:163-171
    return windows.general_gaussian(n_reflect * 2, 3.0, (n_reflect // 2), sym=True)[:n_reflect]


def set_rfft_input(tod, tdata, n_samp, n_buffer, n_reflect, apodize):
    # fft.py:173-188
    tdata[:] = 0
    tdata[n_buffer - n_reflect : n_buffer] = tod[n_reflect - 1 :: -1]
    tdata[n_buffer : n_buffer + n_samp] = tod[:]
    tdata[n_buffer + n_samp : n_buffer + n_samp + n_reflect] = tod[-1 : -(n_reflect + 1) : -1]
    tdata[n_buffer - n_reflect : n_buffer] *= apodize
    tdata[n_buffer + n_samp + n_reflect - 1 : n_buffer + n_samp - 1 : -1] *= apodize


def interpolate_rfft_kernel(kernel_freq, kern, freq, deconvolve):
    # fft.py:190-212
    kern_mag = np.absolute(kern)
    kern_ang = np.angle(kern)
    if deconvolve:
        kern_limit = 1.0e-5 * np.max(kern_mag)
        safe_mag = np.array(kern_mag)
        safe_mag[kern_mag < kern_limit] = kern_limit
    else:
        safe_mag = kern_mag
    mag_out = PchipInterpolator(kernel_freq, safe_mag, extrapolate=True)(freq)
    ang_out = PchipInterpolator(kernel_freq, kern_ang, extrapolate=True)(freq)
    return mag_out * np.exp(1j * ang_out)


def algorithm_numpy(data, rate, kernel_freq, kernels, deconvolve=False):
    """In-place convolution of every row of `data` (fft.py:252-350)."""
    n_tod, n_samp = data.shape
    n_fft = fft_length(n_samp)
    freq = np.fft.rfftfreq(n_fft, d=1.0 / rate)
    n_buffer = (n_fft - n_samp) // 2
    n_reflect = min(n_buffer, n_samp)
    common = None
    if kernels.ndim == 1:
        common = interpolate_rfft_kernel(kernel_freq, kernels, freq, deconvolve)
    apod = apodization(n_reflect)
    tdata = np.empty(n_fft)
    for itod in range(n_tod):
        set_rfft_input(data[itod], tdata, n_samp, n_buffer, n_reflect, apod)
        fdata = np.fft.rfft(tdata, norm="backward")
        krn = common if common is not None else interpolate_rfft_kernel(kernel_freq, kernels[itod], freq, deconvolve)
        if deconvolve:
            fdata /= krn
        else:
            fdata *= krn
        fdata.imag[-1] = 0
        fdata[0] = 0
        tdata[:] = np.fft.irfft(fdata, norm="backward")
        data[itod][:] = tdata[n_buffer : n_buffer + n_samp]


def extend_flags(flags, mask, buffer):
    # utils.py:1055-1113
    matching = np.array(flags & mask, dtype=bool)
    start_flags = np.where(matching[1:] > matching[:-1])[0] + 1
    end_flags = np.where(matching[1:] < matching[:-1])[0] + 1
    if len(start_flags) == 0 and len(end_flags) == 0:
        return
    regions = []
    if len(start_flags) > 0 and len(end_flags) > 0:
        if start_flags[0] < end_flags[0]:
            regions += list(zip(start_flags, end_flags))
            if len(start_flags) != len(end_flags):
                regions.append((start_flags[-1], len(matching)))
        else:
            regions.append((0, end_flags[0]))
            regions += list(zip(start_flags, end_flags[1:]))
            if len(start_flags) == len(end_flags):
                regions.append((start_flags[-1], len(matching)))
    elif len(start_flags) > 0:
        regions.append((start_flags[0], len(matching)))
    else:
        regions.append((0, end_flags[0]))
    for start, end in regions:
        fstart = max(start - buffer, 0)
        fend = end + buffer
        if fend >= len(matching):
            fend = len(matching) - 1
        flags[fstart:fend] = mask


def impulse_extent(n_tod, n_samp, rate, kernel_freq, kernels, deconvolve=False):
    """Spread (in samples) of an impulse through the convolution (fft.py:836-872)."""
    mid = n_samp // 2
    temp = np.zeros((n_tod, n_samp))
    temp[:, mid] = 100.0
    algorithm_numpy(temp, rate, kernel_freq, kernels, deconvolve)
    extend = np.zeros(n_tod, dtype=np.int32)
    for itod in range(n_tod):
        atemp = np.absolute(temp[itod])
        ipeak = np.argmax(atemp)
        apeak = atemp[ipeak]
        imin = ipeak
        while imin > 0 and atemp[imin] > 0.02 * apeak:
            imin -= 1
        imax = ipeak
        while imax < n_samp and atemp[imax] > 0.02 * apeak:
            imax += 1
        extend[itod] = imax - imin
        if extend[itod] == n_samp:
            raise RuntimeError("Impulse response spreads to all samples")
    return extend


def convolve(data, rate, flags=None, flag_mask=None, kernel_freq=None, kernels=None, deconvolve=False):
    """toast.fft.convolve(algorithm="numpy") on a 2-D array (fft.py:700-945)."""
    n_tod, n_samp = data.shape
    extend = np.zeros(n_tod, dtype=np.int32)
    if flags is not None:
        extend = impulse_extent(n_tod, n_samp, rate, kernel_freq, kernels, deconvolve)
    algorithm_numpy(data, rate, kernel_freq, kernels, deconvolve)
    if flags is not None:
        for itod in range(n_tod):
            ext = int(extend[itod])
            extend_flags(flags[itod], flag_mask, ext)
            flags[itod][:ext] |= flag_mask
            flags[itod][-ext:] |= flag_mask


def noise_filter_kernel(psd, net):
    """Inverse-noise kernel of one detector (ops/noise_filter.py:154-169)."""
    psd = np.array(psd, dtype=np.float64)
    net_sq = net**2
    psd_limit = 1.0e-3 * net_sq
    psd[psd < psd_limit] = psd_limit
    psd[:] = 1 / psd
    psd *= net_sq
    psd[0] = 0
    return psd


def r1d_forward(x):
    """FFTW r2hc layout of a batch of real transforms (toast_math_fft_fftw.cpp:26-128)."""
    x = np.atleast_2d(x)
    n = x.shape[1]
    f = np.fft.rfft(x, axis=1)
    out = np.empty_like(x)
    out[:, : n // 2 + 1] = f.real
    out[:, n // 2 + 1 :] = f.imag[:, (n + 1) // 2 - 1 : 0 : -1]
    return out


def r1d_backward(hc):
    hc = np.atleast_2d(hc)
    n = hc.shape[1]
    f = np.zeros((hc.shape[0], n // 2 + 1), dtype=np.complex128)
    f.real = hc[:, : n // 2 + 1]
    f.imag[:, 1 : (n + 1) // 2] = hc[:, n - 1 : n // 2 : -1]
    return np.fft.irfft(f, n=n, axis=1)
