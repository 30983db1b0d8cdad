"""Amplitude-domain noise prior of the Offset template, applied on the device.

The reference builds, per detector and view, a truncated real-space filter (the inverse baseline
covariance) and a preconditioner from the detector PSD, and applies both on the HOST every PCG
iteration with scipy (src/toast/templates/offset/offset.py:455-566 build, :884-960 `_add_prior`,
:963-1005 `_apply_precond`; ``use_accel`` raises NotImplementedError there).  Here the one-off
construction stays on the host -- it is O(n_amplitudes) work done once -- and the per-iteration
application runs in two HIP kernels on the resident amplitude vectors
(toast_hip_template_offset_convolve_dev / _banded_solve_dev, csrc/offset_prior.hip).
"""

import numpy as np
import scipy.linalg
import scipy.optimize

from ..accel import accel_data_create, accel_data_delete, accel_data_update_device, accel_device_ptr

_LOW_FREQ = 1.0e-10  # offset.py:549: |f| below this is treated as this frequency


def _log_interp(x, lfreq, lval):
    """exp(interp(log|x|)) with the reference's zero-frequency clamp (offset.py:547-561)."""
    ax = np.abs(np.asarray(x, dtype=np.float64))
    lx = np.log(np.where(ax < _LOW_FREQ, _LOW_FREQ, ax))
    return np.exp(np.interp(lx, lfreq, lval))


def _centre_and_cut(row, lim=1.0e-4):
    """Shift a circular real-space filter to its centre and keep the symmetric part above ``lim``
    of the zero-lag value (offset.py:563-571)."""
    half = row.size // 2
    above = np.nonzero(np.abs(row[:half]) > np.abs(row[0]) * lim)[0]
    cut = int(above[-1])
    cut += 1 - cut % 2  # odd half-width
    return np.roll(row, half)[half - cut:half + cut + 1]


def _correlated_psd(freq, psd):
    """PSD minus its white plateau, the plateau from a straight-line fit in log-log to the top
    20 % of the bins (offset.py:590-623; same scipy fit so the plateau agrees to rounding)."""
    n = psd.size
    first = int(0.8 * n)
    if n - first < 10:
        first = 0 if n < 10 else n - 10
    lx, ly = np.log(freq[first:]), np.log(psd[first:])

    def line(x, a, b, c):
        return a * (x - b) + c

    par, _ = scipy.optimize.curve_fit(line, lx, ly, p0=[0.0, lx[-1], ly[-1]])
    plateau = np.exp(line(lx, *par))[-1]
    floor = 1.0e-10 * np.amax(psd) - plateau
    return np.maximum(psd - plateau, floor)


def baseline_psd(psdfreq, psd, freq, step_time, m_max=5):
    """PSD of the baseline amplitudes: the detector's correlated PSD aliased by the boxcar of one
    step, P_a(f) = (1/T) sum_m P(f + m/T) sinc^2(pi T (f + m/T)), |m| < m_max
    (offset.py:625-711)."""
    lfreq = np.log(psdfreq)
    lpsd = np.log(_correlated_psd(psdfreq, psd))
    fbase = 1.0 / step_time
    total = None
    # same order of accumulation as the reference: m = 0, then +m, -m for m = 1 .. m_max - 1
    for m in [0] + [s * k for k in range(1, m_max) for s in (1, -1)]:
        x = np.pi * step_time * (freq + m * fbase)
        small = np.abs(x) < 1.0e-30
        sinc2 = np.where(small, 1.0, (np.sin(x) / np.where(small, 1.0, x)) ** 2)
        term = _log_interp(freq + m * fbase, lfreq, lpsd) * sinc2
        total = term if total is None else total + term
    return total * fbase


def prior_frequencies(obstime, step_time, rate):
    """1000 log-spaced frequencies covering the baseline spectrum (offset.py:204-223)."""
    lo = np.floor(np.log10(1.0 / obstime)) - 1
    hi = min(np.ceil(np.log10(1.0 / step_time)) + 2, np.log10(rate))
    return np.logspace(lo, hi, 1000)


def _filter_length(n_amp_view):
    n = 2
    while n < 2 * n_amp_view:
        n *= 2
    return n


class OffsetPrior:
    """Filters, preconditioners and their device tables for one initialised Offset template."""

    def __init__(self, name, precond_width, factor_on_device=None):
        from ..accel import accel_enabled

        self.name = name
        self.precond_width = int(precond_width)
        # banded preconditioner: factorise on the device (one kernel, the factors never exist on
        # the host) unless there is no device -- then scipy, as the reference does
        self.factor_on_device = accel_enabled() if factor_on_device is None else bool(factor_on_device)
        self.detnoise = []         # per segment
        self._offsetvar = None
        self.seg_start = None      # int64[n_seg + 1]
        self.filters = []          # per segment (host, for inspection / tests)
        self.precond = []          # per segment: Toeplitz row or lower banded Cholesky factor
        self._tables = {}
        self._on_device = False

    # ------------------------------------------------------------------ construction (host, once)
    def build(self, segments, offsetvar, step_time):
        """``segments``: list of dicts with keys first, n_amp, freq, psdfreq, psd, detnoise in
        amplitude order.  ``offsetvar``: the template's per-amplitude variances."""
        n_seg = len(segments)
        self._offsetvar = offsetvar
        self.seg_start = np.zeros(n_seg + 1, dtype=np.int64)
        psd_cache, filt_cache = {}, {}
        for iseg, seg in enumerate(segments):
            first, n_amp = int(seg["first"]), int(seg["n_amp"])
            self.seg_start[iseg] = first
            self.seg_start[iseg + 1] = first + n_amp
            pkey = (seg["freq"].tobytes(), seg["psdfreq"].tobytes(), seg["psd"].tobytes())
            if pkey not in psd_cache:
                opsd = baseline_psd(seg["psdfreq"], seg["psd"], seg["freq"], step_time)
                psd_cache[pkey] = (len(psd_cache), np.log(seg["freq"]), np.log(opsd))
            pid, lfreq, lopsd = psd_cache[pkey]
            flen = _filter_length(n_amp)
            fkey = (pid, flen)
            if fkey not in filt_cache:
                ffreq = np.fft.rfftfreq(flen, step_time)
                # the prior is the INVERSE amplitude covariance: 1 / PSD (offset.py:462-472)
                filt_cache[fkey] = _centre_and_cut(np.fft.irfft(_log_interp(ffreq, lfreq, -lopsd)))
            noisefilter = filt_cache[fkey]
            self.filters.append(noisefilter)
            detnoise = float(seg["detnoise"])
            self.detnoise.append(detnoise)
            if self.precond_width <= 1:
                tkey = (pid, flen, "toeplitz")
                if tkey not in filt_cache:
                    ffreq = np.fft.rfftfreq(flen, step_time)
                    filt_cache[tkey] = _centre_and_cut(np.fft.irfft(_log_interp(ffreq, lfreq, lopsd)))
                pre = filt_cache[tkey].copy()
                if detnoise != 0:
                    pre[pre.size // 2] += 1.0 / detnoise  # offset.py:505-507
                self.precond.append(pre)
            elif not self.factor_on_device:
                self.precond.append(self._banded_factor(noisefilter, offsetvar[first:first + n_amp], detnoise))
        return self

    def _banded_factor(self, noisefilter, var, detnoise):
        """Lower banded Cholesky factor of diag(1 / var) + Toeplitz(filter) with the reference's
        width doubling when the truncated matrix is not positive definite (offset.py:522-566)."""
        n_amp = var.size
        centre = noisefilter.size // 2
        width = self.precond_width
        while True:
            wband = min(width, centre)
            rows = max(wband, min(width, n_amp))
            ab = np.zeros((rows, n_amp), dtype=np.float64)
            if detnoise != 0:
                # a flagged amplitude has var = 0: the reference's 1 / 0 = inf makes scipy's
                # check_finite raise; here such an amplitude keeps the Toeplitz part only (its
                # result is zeroed after the solve, and any SPD preconditioner is valid)
                ab[0] = np.where(var > 0, 1.0 / np.where(var > 0, var, 1.0), 0.0)
            ab[:wband] += noisefilter[centre:centre + wband, None]
            try:
                return scipy.linalg.cholesky_banded(ab, overwrite_ab=True, lower=True, check_finite=True)
            except scipy.linalg.LinAlgError:
                if width < centre and width < n_amp:
                    width *= 2
                else:
                    raise RuntimeError(f"{self.name}: banded preconditioner is not positive definite at width {width}")

    # ------------------------------------------------------------------ device tables
    def _register(self, key, arr):
        arr = np.ascontiguousarray(arr)
        if arr.size == 0:
            arr = np.zeros(1, dtype=arr.dtype)
        self._tables[key] = arr
        accel_data_create(arr, f"{self.name}_prior_{key}", owner=self)
        accel_data_update_device(arr, f"{self.name}_prior_{key}")

    def _ptr(self, key):
        return accel_device_ptr(self._tables[key])

    @property
    def _max_seg(self):
        return int(np.max(np.diff(self.seg_start))) if self.seg_start.size > 1 else 0

    def to_device(self):
        if self._on_device:
            return
        n_seg = self.seg_start.size - 1
        self._register("seg_start", self.seg_start)

        def pack(rows):
            start = np.zeros(n_seg, dtype=np.int64)
            length = np.array([r.size for r in rows], dtype=np.int64)
            # identical rows (shared by every detector with the same PSD) are stored once
            pool, seen, cursor = [], {}, 0
            for i, r in enumerate(rows):
                k = id(r)
                if k not in seen:
                    seen[k] = cursor
                    pool.append(r)
                    cursor += r.size
                start[i] = seen[k]
            return start, length, np.concatenate(pool) if pool else np.zeros(1)

        fs, fl, fp = pack(self.filters)
        self._register("filt_start", fs)
        self._register("filt_len", fl)
        self._register("filters", fp)
        if self.precond_width <= 1:
            ps, pl, pp = pack(self.precond)
            self._register("pre_start", ps)
            self._register("pre_len", pl)
            self._register("pre_filters", pp)
        elif self.factor_on_device:
            self._factor_on_device()
        else:
            width = np.array([cb.shape[0] for cb in self.precond], dtype=np.int32)
            start, total = self._band_layout(width)
            fwd = np.zeros(max(total, 1), dtype=np.float64)
            bwd = np.zeros(max(total, 1), dtype=np.float64)
            for iseg, cb in enumerate(self.precond):
                w, n = cb.shape
                f = fwd[start[iseg]:start[iseg] + n * w].reshape(n, w)
                b = bwd[start[iseg]:start[iseg] + n * w].reshape(n, w)
                rdiag = 1.0 / cb[0]
                f[:, 0] = rdiag
                b[:, 0] = rdiag
                for k in range(1, min(w, n)):
                    f[:n - k, k] = cb[k, :n - k]  # L[i + k][i]: unknown i feeds i + k going forward
                    b[k:, k] = cb[k, :n - k]      # L[i][i - k]: unknown i feeds i - k going backward
            self.max_width = int(width.max()) if n_seg else 1
            self._register("band_width", width)
            self._register("band_start", start)
            self._register("forward", fwd)
            self._register("backward", bwd)
        self._on_device = True

    def _band_layout(self, width):
        n_amp = np.diff(self.seg_start)
        start = np.zeros(width.size, dtype=np.int64)
        if width.size > 1:
            start[1:] = np.cumsum(n_amp[:-1] * width[:-1])
        return start, int(np.sum(n_amp * width))

    def _create(self, key, arr):
        """Device buffer keyed by ``arr`` without an upload (``arr`` is never touched)."""
        self._tables[key] = arr
        accel_data_create(arr, f"{self.name}_prior_{key}", owner=self)

    def _drop(self, key):
        accel_data_delete(self._tables.pop(key), f"{self.name}_prior_{key}")

    def _factor_on_device(self):
        """Banded Cholesky factors of all segments in one launch
        (toast_hip_template_offset_banded_cholesky_dev), widening the band of the segments that
        are not positive definite the way the reference does (offset.py:546-566)."""
        from .. import capi
        from ..accel import accel_data_update_host, native

        n_seg = self.seg_start.size - 1
        n_amp = np.diff(self.seg_start)
        centre = np.array([f.size // 2 for f in self.filters], dtype=np.int64)
        try_width = np.full(n_seg, self.precond_width, dtype=np.int64)
        dscale = np.array([1.0 if d != 0 else 0.0 for d in self.detnoise], dtype=np.float64)
        self._register("diag_scale", dscale)
        self._register("offset_var", np.ascontiguousarray(self._offsetvar))
        status = np.zeros(max(n_seg, 1), dtype=np.int32)
        self._create("status", status)
        while True:
            wband = np.minimum(try_width, centre)
            width = np.maximum(wband, np.minimum(try_width, n_amp)).astype(np.int32)
            if int(width.max()) > 64:
                raise RuntimeError(f"{self.name}: banded preconditioner needs a band of {int(width.max())} > 64 "
                                   "baselines; use precond_width <= 1 (Toeplitz) for this noise model")
            # Toeplitz bands: filter[centre : centre + wband], stored once per distinct filter
            tstart = np.zeros(n_seg, dtype=np.int64)
            pool, seen, cursor = [], {}, 0
            for i, filt in enumerate(self.filters):
                k = (id(filt), int(wband[i]))
                if k not in seen:
                    seen[k] = cursor
                    pool.append(filt[centre[i]:centre[i] + wband[i]])
                    cursor += int(wband[i])
                tstart[i] = seen[k]
            start, total = self._band_layout(width)
            for key, arr in (("toep_start", tstart), ("toep_len", wband.astype(np.int32)),
                             ("toeplitz", np.concatenate(pool)), ("band_width", width), ("band_start", start)):
                self._register(key, arr)
            # the factor tables exist on the device only; the host arrays are untouched keys
            self._create("forward", np.empty(max(total, 1), dtype=np.float64))
            self._create("backward", np.empty(max(total, 1), dtype=np.float64))
            capi.dev.memset(self._ptr("backward"), 0, 8 * max(total, 1))
            capi.dev.offset_banded_cholesky(n_seg, self._ptr("seg_start"), self._ptr("band_width"), int(width.max()),
                                            self._ptr("band_start"), self._ptr("toep_start"), self._ptr("toep_len"),
                                            self._ptr("toeplitz"), self._ptr("diag_scale"), self._ptr("offset_var"),
                                            self._ptr("forward"), self._ptr("backward"), self._ptr("status"))
            native().accel_synchronize()
            accel_data_update_host(status, f"{self.name}_prior_status")
            failed = np.nonzero(status[:n_seg])[0]
            if failed.size == 0:
                break
            for i in failed:
                if try_width[i] < centre[i] and try_width[i] < n_amp[i]:
                    try_width[i] *= 2
                else:
                    raise RuntimeError(f"{self.name}: banded preconditioner of segment {i} is not positive "
                                       f"definite at the maximum width {int(try_width[i])}")
            for key in ("toep_start", "toep_len", "toeplitz", "band_width", "band_start", "forward", "backward"):
                self._drop(key)
        self.max_width = int(width.max())
        for key in ("toep_start", "toep_len", "toeplitz", "diag_scale", "offset_var", "status"):
            self._drop(key)

    def banded_factor(self, iseg):
        """Lower banded factor of one segment in scipy's layout, read back from the device tables
        (tests / inspection)."""
        from ..accel import accel_data_update_host

        if not self.factor_on_device:
            return self.precond[iseg]
        self.to_device()
        fwd = self._tables["forward"]
        accel_data_update_host(fwd, f"{self.name}_prior_forward")
        w = int(self._tables["band_width"][iseg])
        n = int(self.seg_start[iseg + 1] - self.seg_start[iseg])
        s0 = int(self._tables["band_start"][iseg])
        cb = fwd[s0:s0 + n * w].reshape(n, w).T.copy()
        cb[0] = 1.0 / cb[0]
        return cb

    def clear(self):
        for key, arr in self._tables.items():
            accel_data_delete(arr, f"{self.name}_prior_{key}")
        self._tables = {}
        self._on_device = False

    # ------------------------------------------------------------------ application (device)
    def add_prior(self, amps_in, amps_out):
        """amps_out += C_a^-1 amps_in, flagged amplitudes zeroed (offset.py:884-960)."""
        from .. import capi

        self.to_device()
        n_seg = self.seg_start.size - 1
        capi.dev.offset_convolve(int(self.seg_start[-1]), n_seg, self._ptr("seg_start"), self._max_seg,
                                 self._ptr("filt_start"), self._ptr("filt_len"), max(f.size for f in self.filters),
                                 self._ptr("filters"), accel_device_ptr(amps_in.buffer),
                                 accel_device_ptr(amps_in.local_flags), accel_device_ptr(amps_out.buffer), True)

    def apply_precond(self, amps_in, amps_out):
        """offset.py:963-1005: Toeplitz convolution (width <= 1) or banded Cholesky solve."""
        from .. import capi

        self.to_device()
        n_seg = self.seg_start.size - 1
        if self.precond_width <= 1:
            capi.dev.offset_convolve(int(self.seg_start[-1]), n_seg, self._ptr("seg_start"), self._max_seg,
                                     self._ptr("pre_start"), self._ptr("pre_len"), max(p.size for p in self.precond),
                                     self._ptr("pre_filters"), accel_device_ptr(amps_in.buffer),
                                     accel_device_ptr(amps_in.local_flags), accel_device_ptr(amps_out.buffer), False)
        else:
            capi.dev.offset_banded_solve(n_seg, self._ptr("seg_start"), self._ptr("band_width"), self.max_width,
                                         self._ptr("band_start"), self._ptr("forward"), self._ptr("backward"),
                                         accel_device_ptr(amps_in.buffer), accel_device_ptr(amps_in.local_flags),
                                         accel_device_ptr(amps_out.buffer))
