"""FFT noise weighting on the GPU: the product-side mirror of ``toast.fft.convolve``
(reference: src/toast/fft.py:700-945, ``algorithm="numpy"`` semantics, :252-350).

Host logic only prepares small things -- the PCHIP piecewise cubics of |K| and arg K on the
kernel frequencies (SciPy, as the reference does at fft.py:190-212), the apodisation window
(fft.py:163-171) and the flag bookkeeping (fft.py:836-872, :935-945); the padding, the
batched rocFFT transforms, the kernel evaluation/multiply and the crop all run on the GPU
through ``toast_hip_fft_convolve`` (toast_amd/csrc/fft_filter.hip).
"""

import ctypes as C
import functools

import numpy as np
from scipy.interpolate import PchipInterpolator

from . import capi


def fft_length(n_samp):
    return int(capi.lib().toast_hip_fft_length(C.c_int64(int(n_samp))))


def implementation(n_samp=720000):
    """"fused-3pass" (toast_amd/csrc/fft_fused.hip) or "rocfft" for a timestream of this length."""
    return "fused-3pass" if capi.lib().toast_hip_fft_fused(C.c_int64(int(n_samp))) else "rocfft"


def select(rocfft_only):
    """Force the rocFFT pipeline for every length (True) or return to the automatic choice (False)."""
    capi.lib().toast_hip_fft_select(C.c_int(1 if rocfft_only else 0))


def set_points(rows=0, cols_fwd=0, cols_inv=0):
    """Points per thread (16 or 8, 0 = default) of the row pass / forward / inverse column pass of the
    fused kernels."""
    capi.lib().toast_hip_fft_points(C.c_int(int(rows)), C.c_int(int(cols_fwd)), C.c_int(int(cols_inv)))


def set_rows_mode(mode="reg"):
    """Row pass of the fused kernels: "reg" (default: tile in registers, 16 points per lane = a row in two waves, three
    workgroups per CU; fft_reg.hip), "reg32" (32 points per lane, one row per wave), "pair" (row pair in one 64 KB LDS tile)
    or "split" (one row per 32 KB LDS tile; experiment)."""
    capi.lib().toast_hip_fft_rows_split(C.c_int({"pair": 0, "lds": 0, "split": 1, "reg": 2, "reg32": 3}[mode]))


def set_cols_mode(mode="reg"):
    """Column passes of the fused kernels: "reg" (default: tile in registers for n_fft 2^22 and 2^23), "reg9" (for 2^21
    too) or "lds" (csrc/fft_fused.hip for every length)."""
    capi.lib().toast_hip_fft_cols_reg(C.c_int({"lds": 0, "reg": 1, "reg9": 2}[mode]))


def set_rows_split(split=True):
    """Older spelling: the "split" row pass (True) or the default one (False)."""
    set_rows_mode("split" if split else "reg")


def set_rows_n2(n2=2048):
    """Row length N2 of the four-step factorisation of the fused kernels: 2048 or 1024 (toast_hip_fft_rows_n2)."""
    capi.lib().toast_hip_fft_rows_n2(C.c_int(int(n2)))


def pipeline_bytes_per_sample(n_samp):
    """HBM bytes per timestream sample moved by the passes of the implementation in use."""
    fn = capi.lib().toast_hip_fft_pipeline_bytes
    fn.restype = C.c_double
    return float(fn(C.c_int64(int(n_samp))))


@functools.lru_cache(maxsize=8)
def apodization(n_reflect):
    """First half of ``general_gaussian(2 n_reflect, p=3, sig=n_reflect // 2)``
    (reference fft.py:163-171; formula of scipy.signal.windows.general_gaussian).  Cached per length
    (read-only array): the window of a cfg-3 timestream has 688 576 entries, 15 - 30 ms of exp()."""
    m = 2 * n_reflect
    n = np.arange(0, m) - (m - 1.0) / 2.0
    sig = n_reflect // 2
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.exp(-0.5 * np.abs(n / sig) ** (2 * 3.0))
    out = np.ascontiguousarray(w[:n_reflect])
    out.setflags(write=False)
    return out


def _pchip_coef(kernel_freq, values):
    """scipy PPoly coefficients [n_kernel, n_knot - 1, 4] of the PCHIP interpolant(s)."""
    values = np.atleast_2d(values)
    # one vectorised construction for all kernels (bit-identical to one interpolant per row)
    c = PchipInterpolator(kernel_freq, values.T, axis=0, extrapolate=True).c   # [4, n_knot - 1, n_kernel]
    return np.ascontiguousarray(np.transpose(c, (2, 1, 0)))


def kernel_coefficients(kernel_freq, kernels, deconvolve=False):
    """|K| and arg K as piecewise cubics (reference fft.py:190-212)."""
    kernels = np.asarray(kernels)
    mag = np.absolute(kernels)
    ang = np.angle(kernels)
    if deconvolve:
        m2 = np.atleast_2d(mag).copy()
        for row in m2:
            limit = 1.0e-5 * np.max(row)
            row[row < limit] = limit
        mag = m2
    mag_c = _pchip_coef(kernel_freq, mag)
    ang_c = None if not np.any(ang) else _pchip_coef(kernel_freq, ang)
    return mag_c, ang_c


def _p(a):
    return C.c_void_p(0) if a is None else C.c_void_p(a.ctypes.data)


def convolve_buffer(det_data, data_index, rate, kernel_freq, kernels, deconvolve=False, use_accel=False):
    """Convolve rows ``data_index`` of the 2-D float64 buffer ``det_data`` in place."""
    from .accel import ensure_assigned

    ensure_assigned()
    det_data = capi._buf(det_data, "det_data", np.float64, 2)
    di = capi._buf(np.ascontiguousarray(data_index, dtype=np.int32), "data_index", np.int32, 1)
    n_det, n_samp = di.size, det_data.shape[1]
    kernel_freq = np.ascontiguousarray(kernel_freq, dtype=np.float64)
    kernels = np.asarray(kernels)
    if kernels.ndim == 2 and kernels.shape[0] != n_det:
        raise RuntimeError("kernels should have one row per detector")
    mag_c, ang_c = kernel_coefficients(kernel_freq, kernels, deconvolve)
    n_fft = fft_length(n_samp)
    n_buffer = (n_fft - n_samp) // 2
    n_reflect = min(n_buffer, n_samp)
    apod = apodization(n_reflect)
    capi._check(capi.lib().toast_hip_fft_convolve(
        _p(det_data), C.c_int64(det_data.shape[0]), _p(di), C.c_int64(n_det), C.c_int64(n_samp),
        C.c_double(rate), _p(kernel_freq), C.c_int64(kernel_freq.size), _p(mag_c), _p(ang_c),
        C.c_int64(mag_c.shape[0]), C.c_int(bool(deconvolve)), _p(apod), C.c_int64(apod.size),
        C.c_int(bool(use_accel))))


def convolve_dev(d_det_data, data_index, n_samp, rate, kernel_freq, kernels, deconvolve=False, max_batch=0,
                 stream=0):
    """Device-pointer variant (``d_det_data`` = device address of a [rows, n_samp] f64 array)."""
    di = np.ascontiguousarray(data_index, dtype=np.int32)
    kernel_freq = np.ascontiguousarray(kernel_freq, dtype=np.float64)
    mag_c, ang_c = kernel_coefficients(kernel_freq, np.asarray(kernels), deconvolve)
    n_fft = fft_length(n_samp)
    n_buffer = (n_fft - n_samp) // 2
    n_reflect = min(n_buffer, n_samp)
    apod = apodization(n_reflect)
    capi._check(capi.lib().toast_hip_fft_convolve_dev(
        C.c_void_p(int(d_det_data)), _p(di), C.c_int64(di.size), C.c_int64(n_samp), C.c_double(rate),
        _p(kernel_freq), C.c_int64(kernel_freq.size), _p(mag_c), _p(ang_c), C.c_int64(mag_c.shape[0]),
        C.c_int(bool(deconvolve)), _p(apod), C.c_int64(apod.size), C.c_int64(max_batch),
        C.c_void_p(int(stream))))


def extend_flags(flags, mask, buffer):
    """Grow every flagged region by ``buffer`` samples on both sides
    (reference: src/toast/utils.py:1055-1113).  The reference loops over the regions and ASSIGNS the mask to
    [start - buffer, end + buffer) (end clipped to n - 1 when it reaches n); here the union of those ranges is
    built with a difference array, same result for any number of regions."""
    bad = (flags & mask) != 0
    if not bad.any() or bad.all():
        return        # a completely flagged array has no edge: the reference leaves it (and its other bits) alone
    edges = np.diff(np.concatenate([[0], bad.view(np.int8), [0]]))
    starts = np.flatnonzero(edges == 1)
    ends = np.flatnonzero(edges == -1)
    n = flags.size
    fstart = np.maximum(starts - buffer, 0)
    fend = ends + buffer
    fend = np.where(fend >= n, n - 1, fend)
    keep = fend > fstart                      # an empty slice assigns nothing
    delta = np.zeros(n + 1, dtype=np.int32)
    np.add.at(delta, fstart[keep], 1)
    np.add.at(delta, fend[keep], -1)
    flags[np.cumsum(delta[:n]) > 0] = mask


def extend_flags_buffer(flags, flag_index, mask, extents, edges=True, use_accel=False, or_row=None):
    """``extend_flags`` (and, with ``edges``, the flagging of the first and last ``extent`` samples that
    toast.fft.convolve does afterwards) for rows ``flag_index`` of the 2-D uint8 buffer ``flags``, each with its
    own extent, on the device (``toast_hip_fft_extend_flags``).  ``use_accel``: the buffer's registered device copy
    is updated instead of the host array.  ``or_row`` (uint8 [n_samp]) is OR-ed into every selected row first."""
    from .accel import ensure_assigned

    ensure_assigned()
    fl = capi._buf(flags, "flags", np.uint8, 2)
    fi = capi._buf(np.ascontiguousarray(flag_index, dtype=np.int32), "flag_index", np.int32, 1)
    ex = np.ascontiguousarray(extents, dtype=np.int32)
    if ex.shape != fi.shape:
        raise RuntimeError("extents should have one entry per flag row")
    orr = None
    if or_row is not None:
        orr = np.ascontiguousarray(or_row, dtype=np.uint8)
        if orr.shape != (fl.shape[1],):
            raise RuntimeError("or_row should have one entry per sample")
    capi._check(capi.lib().toast_hip_fft_extend_flags(
        _p(fl), C.c_int64(fl.shape[0]), _p(fi), C.c_int64(fi.size), C.c_int64(fl.shape[1]), C.c_uint8(int(mask)),
        _p(ex), C.c_int(bool(edges)), _p(orr) if orr is not None else None, C.c_int(bool(use_accel))))


def impulse_extent(atemp):
    """Width of the impulse response in one row of |convolved impulse| as the reference measures it
    (src/toast/fft.py:846-866): walk left and right from the peak while the response exceeds 2 % of it.  Same result
    as the reference's sample-by-sample loops, found with vectorised searches in windows that double in size."""
    n = atemp.shape[0]
    ipeak = int(np.argmax(atemp))
    thr = 0.02 * atemp[ipeak]
    # imin = largest j <= ipeak with atemp[j] <= thr, else 0
    imin, hi, width = 0, ipeak + 1, 1024
    while hi > 0:
        lo = max(hi - width, 0)
        below = np.flatnonzero(atemp[lo:hi] <= thr)
        if below.size:
            imin = lo + int(below[-1])
            break
        hi, width = lo, 2 * width
    # imax = smallest j >= ipeak with atemp[j] <= thr, else n
    imax, lo, width = n, ipeak, 1024
    while lo < n:
        hi = min(lo + width, n)
        below = np.flatnonzero(atemp[lo:hi] <= thr)
        if below.size:
            imax = lo + int(below[0])
            break
        lo, width = hi, 2 * width
    return imax - imin


def impulse_extents(n_tod, n_samp, rate, kernel_freq, kernels, deconvolve=False):
    """Width of every kernel's impulse response (reference src/toast/fft.py:836-872), measured on the device
    (``toast_hip_fft_impulse_extents``): impulses made, convolved and searched in HBM, only ``n_tod`` integers come
    back.  Same numbers as convolving ``temp[:, n_samp // 2] = 100`` on the host and ``impulse_extent`` per row."""
    from .accel import ensure_assigned

    ensure_assigned()
    kernel_freq = np.ascontiguousarray(kernel_freq, dtype=np.float64)
    kernels = np.asarray(kernels)
    if kernels.ndim == 2 and kernels.shape[0] != n_tod:
        raise RuntimeError("kernels should have one row per detector")
    if kernels.ndim == 2 and n_tod > 1:
        # the width is a function of the kernel alone: detectors sharing a noise model share one measurement
        uniq, inverse = np.unique(kernels, axis=0, return_inverse=True)
        if uniq.shape[0] < n_tod:
            return impulse_extents(uniq.shape[0], n_samp, rate, kernel_freq, uniq, deconvolve)[np.ravel(inverse)]
    mag_c, ang_c = kernel_coefficients(kernel_freq, kernels, deconvolve)
    n_fft = fft_length(n_samp)
    n_reflect = min((n_fft - n_samp) // 2, n_samp)
    apod = apodization(n_reflect)
    out = np.zeros(n_tod, dtype=np.int32)
    capi._check(capi.lib().toast_hip_fft_impulse_extents(
        C.c_int64(n_tod), C.c_int64(n_samp), C.c_double(rate), _p(kernel_freq), C.c_int64(kernel_freq.size), _p(mag_c),
        _p(ang_c), C.c_int64(mag_c.shape[0]), C.c_int(bool(deconvolve)), _p(apod), C.c_int64(apod.size), _p(out),
        C.c_void_p(0)))
    return out


def convolve(raw, rate, flags=None, flag_mask=None, kernel_freq=None, kernels=None, kernel_func=None,
             deconvolve=False, algorithm="numpy", use_accel=False):
    """Drop-in for ``toast.fft.convolve`` (2-D ``raw`` = one row per timestream, in place).

    ``algorithm`` "numpy"/"internal"/None all map to the rocFFT pipeline (they are the same
    mathematics in the reference: fft.py:296-350 vs :396-484); "nonuniform" and
    ``kernel_func`` are not part of the hot path and raise.
    """
    if kernel_func is not None:
        raise NotImplementedError("kernel_func kernels are evaluated on the host in the reference; not supported")
    if algorithm not in (None, "numpy", "internal"):
        raise RuntimeError(f"Unknown or unsupported algorithm '{algorithm}'")
    if kernel_freq is None or kernels is None:
        raise RuntimeError("Must specify explicit kernel values")
    if (flags is None) != (flag_mask is None):
        raise RuntimeError("Both flags and flag_mask must be specified or set to None")
    raw = np.asarray(raw)
    one_d = raw.ndim == 1
    data = raw.reshape(1, -1) if one_d else raw
    if data.ndim != 2 or data.dtype != np.float64 or not data.flags["C_CONTIGUOUS"]:
        raise RuntimeError("Only contiguous 1D and 2D float64 arrays are supported as inputs")
    n_tod, n_samp = data.shape
    idx = np.arange(n_tod, dtype=np.int32)
    extend = np.zeros(n_tod, dtype=np.int32)
    if flags is not None:
        # impulse response spread (fft.py:836-872), through the same GPU pipeline, measured on the device
        extend[:] = impulse_extents(n_tod, n_samp, rate, kernel_freq, kernels, deconvolve)
        if np.any(extend == n_samp):
            raise RuntimeError("Impulse response spreads to all samples")
    convolve_buffer(data, idx, rate, kernel_freq, kernels, deconvolve, use_accel=use_accel)
    if flags is not None:
        if isinstance(flags, np.ndarray) and flags.ndim == 2 and flags.dtype == np.uint8 and flags.flags["C_CONTIGUOUS"]:
            extend_flags_buffer(flags, idx, flag_mask, extend)
        else:
            for itod in range(n_tod):
                ext = int(extend[itod])
                extend_flags(flags[itod], flag_mask, ext)
                flags[itod][:ext] |= flag_mask
                flags[itod][-ext:] |= flag_mask


def _r1d(indata, direction):
    """The reference's r1d_forward / r1d_backward (src/toast/fft.py:26-117) on the plan store of
    the native module: copy into the plan's buffers, exec, copy out."""
    import toast_amd

    from .accel import ensure_assigned

    ensure_assigned()
    m = toast_amd.load_native()
    x = np.asarray(indata, dtype=np.float64)
    one = x.ndim == 1
    count, length = (1, x.shape[0]) if one else x.shape
    store = m.FFTPlanReal1DStore.get()
    plan = store.forward(length, count) if direction == "forward" else store.backward(length, count)
    src, dst = (plan.tdata, plan.fdata) if direction == "forward" else (plan.fdata, plan.tdata)
    if one:
        src(0)[:] = x
    else:
        for i in range(count):
            src(i)[:] = x[i]
    plan.exec()
    if one:
        return np.array(dst(0))
    out = np.zeros_like(x)
    for i in range(count):
        out[i] = dst(i)
    return out


def r1d_forward(indata):
    """Batched real FFT in FFTW half-complex layout (reference fft.py:26-68)."""
    return _r1d(indata, "forward")


def r1d_backward(indata):
    """Inverse of :func:`r1d_forward` (scaled by 1/length; reference fft.py:71-117)."""
    return _r1d(indata, "backward")
