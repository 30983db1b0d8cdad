"""NoiseFilter: convolve each detector timestream with N_tt'^-1 = NET^2 / PSD(f) using the
rocFFT pipeline (reference: src/toast/ops/noise_filter.py:20-205 -> toast.fft.convolve)."""

import numpy as np

from .. import fft as hipfft
from ..accel import accel_enabled
from ..data import defaults
from ..traits import Float, Int, Unicode
from .operator import Operator


def _fit_window(n_psd):
    """First bin and polynomial degree of the high-frequency fit (noise_model.py:133-143)."""
    offset = int(0.8 * n_psd)
    deg = 2
    if n_psd - offset < 10:
        deg = 1
        offset = 0 if n_psd < 10 else n_psd - 10
    return offset, deg


def estimate_net_reference(freqs, data):
    """The reference's ``estimate_net`` (src/toast/ops/noise_model.py:108-170), statement by statement: a parabola
    a (x - b)^2 + c (a line when fewer than 10 bins are left) fitted in log-log space to the last 20 % of the
    spectrum with scipy's iterative ``curve_fit`` from the reference's starting point, NET = sqrt(exp(fit(last bin))).
    The fit stops at scipy's convergence tolerance (~1e-8), so only the same calls reproduce the reference's
    number: this is what NoiseFilter uses by default."""
    from scipy.optimize import curve_fit

    def quad_func(x, a, b, c):
        return a * (x - b) ** 2 + c

    def lin_func(x, a, b, c):
        return a * (x - b) + c

    data = np.asarray(data)
    freqs = np.asarray(freqs)
    n_psd = len(data)
    offset = int(0.8 * n_psd)
    try_quad = True
    if n_psd - offset < 10:
        try_quad = False
        offset = 0 if n_psd < 10 else n_psd - 10
    ffreq = np.log(freqs[offset:])
    fdata = np.log(data[offset:])
    if try_quad:
        try:
            params, _ = curve_fit(quad_func, ffreq, fdata, p0=[1.0, ffreq[-1], fdata[-1]])
            return np.sqrt(np.exp(quad_func(ffreq, params[0], params[1], params[2]))[-1])
        except RuntimeError:
            pass
    params, _ = curve_fit(lin_func, ffreq, fdata, p0=[0.0, ffreq[-1], fdata[-1]])
    return np.sqrt(np.exp(lin_func(ffreq, params[0], params[1], params[2]))[-1])


def estimate_net_stack(freqs, psds, method="reference"):
    """White-noise level of a stack of PSDs on one frequency grid.

    ``method="reference"``: ``estimate_net_reference`` per distinct PSD (identical rows are fitted once; ~0.2 ms
    each).  ``method="closed_form"``: the same least-squares problem solved exactly for all detectors at once
    (np.polyfit with the PSDs as columns) -- it is the minimum the reference's iteration converges to, and agrees with
    it to that iteration's tolerance (measured 2e-9 .. 5e-8 relative; tests/test_fft_oracle.py), at 1/1000 of the
    time."""
    psds = np.atleast_2d(np.asarray(psds, dtype=np.float64))
    if method == "reference":
        freqs = np.asarray(freqs, dtype=np.float64)
        out = np.empty(psds.shape[0])
        seen = {}
        for i, row in enumerate(psds):
            key = row.tobytes()
            if key not in seen:
                seen[key] = float(estimate_net_reference(freqs, row))
            out[i] = seen[key]
        return out
    if method != "closed_form":
        raise RuntimeError(f"unknown NET estimator '{method}'")
    offset, deg = _fit_window(psds.shape[1])
    x = np.log(np.asarray(freqs, dtype=np.float64)[offset:])
    y = np.log(psds[:, offset:])
    if x.size < deg + 1:
        return np.sqrt(np.exp(np.mean(y, axis=1)))
    coef = np.polyfit(x - x[-1], y.T, deg)
    return np.sqrt(np.exp(coef[-1]))


def estimate_net(freqs, data, method="reference"):
    """``estimate_net_stack`` for one PSD."""
    return float(estimate_net_stack(freqs, np.asarray(data, dtype=np.float64)[None, :], method=method)[0])


class NoiseFilter(Operator):
    """Apply the inverse noise covariance to the signal in the Fourier domain."""

    API = Int(0, help="Internal interface version for this operator")
    times = Unicode(defaults.times, help="Observation shared key for timestamps")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    det_mask = Int(defaults.det_mask_invalid, help="Bit mask value for per-detector flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for flags to use")
    det_flag_mask = Int(defaults.det_mask_invalid, help="Bit mask value for detector sample flagging")
    shared_flags = Unicode(defaults.shared_flags, allow_none=True, help="Observation shared key for telescope flags to use")
    shared_flag_mask = Int(defaults.shared_mask_invalid, help="Bit mask value for optional shared flagging")
    noise_model = Unicode(defaults.noise_model, help="Observation key containing the noise model")
    white_noise_min = Float(None, allow_none=True, help="Minimum frequency of the white noise plateau [Hz]")
    white_noise_max = Float(None, allow_none=True, help="Maximum frequency of the white noise plateau [Hz]")
    debug = Unicode(None, allow_none=True, help="Path to directory for generating debug plots (not produced here)")
    upload_parts = Int(4, help="Row blocks in which a host-resident timestream buffer of 256 MB or more is uploaded, "
                       "the transform of one block overlapping the upload of the next (1: one blocking upload)")
    net_fit = Unicode("reference", help="White-noise estimate when no plateau range is given: 'reference' = the "
                      "reference's iterative curve_fit per detector (its exact numbers), 'closed_form' = the same least "
                      "squares solved exactly for all detectors at once (agrees to ~1e-8, 1000x faster)")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        if self.white_noise_max is not None and self.white_noise_min is None:
            raise RuntimeError("You must set both of the min / max values or neither of them")
        for obs in data.obs:
            dets = obs.select_local_detectors(detectors, flagmask=self.det_mask)
            if len(dets) == 0:
                continue
            rate = obs.telescope.focalplane.sample_rate
            dd = obs.detdata[self.det_data]
            on_dev = dd.accel_in_use()
            made_resident = False
            upload_bounds = None
            if not on_dev and accel_enabled():
                # one page-locked upload instead of per-call pageable staging; the filtered
                # timestream stays resident for the map-maker (lazy host coherence)
                if not dd.accel_exists():
                    dd.accel_create(self.det_data)
                rows = dd.buffer.shape[0]
                n_parts = min(self.upload_parts, rows) if dd.buffer.nbytes >= (256 << 20) else 1
                if n_parts > 1:
                    # Large buffer: the upload is enqueued in row blocks on the library's upload stream and this
                    # thread moves on -- the kernels below are built while the data crosses PCIe, and the transform
                    # of block k runs while block k + 1 is still on its way (toast_hip_accel_update_device_parts).
                    from .. import capi

                    upload_bounds = np.linspace(0, rows, n_parts + 1).astype(np.int64)
                    capi.accel_update_device_parts(dd.buffer, upload_bounds[1:] * (dd.buffer.nbytes // rows), self.det_data)
                    dd.accel_used(True)
                else:
                    dd.accel_update_device()
                on_dev = made_resident = True
            flags = None
            flag_mask = None
            shflg = None
            flags_in_flight = False
            if self.det_flags is not None:
                flags = obs.detdata[self.det_flags]
                if upload_bounds is not None and not flags.accel_in_use() and flags.buffer.nbytes >= (16 << 20):
                    # the detector flags follow the timestream on the upload stream: they cross PCIe while the
                    # transforms run, instead of queueing behind them on the compute stream
                    from .. import capi

                    if not flags.accel_exists():
                        flags.accel_create(self.det_flags)
                    capi.accel_update_device_parts(flags.buffer, np.array([flags.buffer.nbytes]), self.det_flags)
                    flags.accel_used(True)
                    flags_in_flight = True
                if self.shared_flags is not None:
                    # (the flag VALUE times the mask, as the reference writes it: noise_filter.py:121-124); OR-ed into
                    # the detector flags by the same device pass that extends them (nothing reads them in between)
                    shflg = (self.det_flag_mask * np.array(
                        obs.shared[self.shared_flags].data & self.shared_flag_mask, dtype=np.uint8)).astype(np.uint8)
                flag_mask = self.det_flag_mask
            # N_tt'^-1 kernels (noise_filter.py:130-171), all detectors at once
            nse = obs[self.noise_model]
            kern_freq = np.asarray(nse.freq(dets[0]), dtype=np.float64)
            for d in dets[1:]:
                freq = np.asarray(nse.freq(d), dtype=np.float64)
                if freq.shape != kern_freq.shape or not np.allclose(kern_freq, freq):
                    raise RuntimeError("All detectors in the noise model must have the same frequency binning")
            psds = np.array([np.asarray(nse.psd(d), dtype=np.float64) for d in dets])
            if self.white_noise_max is None:
                net = estimate_net_stack(kern_freq, psds, method=self.net_fit)
            else:
                plateau = np.logical_and(kern_freq > self.white_noise_min, kern_freq < self.white_noise_max)
                net = np.sqrt(np.mean(psds[:, plateau], axis=1))
            net_sq = net**2
            kernels = net_sq[:, None] / np.maximum(psds, 1.0e-3 * net_sq[:, None])
            kernels[:, 0] = 0
            idx = dd.indices(dets)
            extend = np.zeros(len(dets), dtype=np.int32)
            n_samp = dd.shape[1]
            if flags is not None:
                # impulse response spread (fft.py:836-872) through the same GPU pipeline, measured on the device
                extend[:] = hipfft.impulse_extents(len(dets), n_samp, rate, kern_freq, kernels)
                if np.any(extend == n_samp):
                    raise RuntimeError("Impulse response spreads to all samples")
            if upload_bounds is not None:
                from .. import capi

                for part in range(len(upload_bounds) - 1):
                    sel = np.flatnonzero((idx >= upload_bounds[part]) & (idx < upload_bounds[part + 1]))
                    if sel.size == 0:
                        continue
                    capi.accel_update_device_wait(dd.buffer, part)      # the stream waits, the host does not
                    hipfft.convolve_buffer(dd.arg(True), idx[sel], rate, kern_freq, kernels[sel], use_accel=True)
                capi.accel_update_device_finish(dd.buffer)
            else:
                hipfft.convolve_buffer(dd.arg(on_dev), idx, rate, kern_freq, kernels, use_accel=on_dev)
            if made_resident and not getattr(data, "lazy_host", False):
                dd.accel_update_host()
                dd.accel_delete()
            if flags is not None:
                # shared flags + extend_flags + first / last samples (fft.py:935-945) for all detectors in one device
                # pass over the resident detector flags (uploaded once; the map-maker reads them there)
                fdata = flags
                if accel_enabled():
                    if flags_in_flight:
                        from .. import capi

                        capi.accel_update_device_wait(fdata.buffer, 0)       # the stream waits, the host does not
                    elif not fdata.accel_in_use():
                        if not fdata.accel_exists():
                            fdata.accel_create(self.det_flags)
                        fdata.accel_update_device()
                    hipfft.extend_flags_buffer(fdata.buffer, fdata.indices(dets), flag_mask, extend, or_row=shflg,
                                               use_accel=True)
                    if flags_in_flight:
                        capi.accel_update_device_finish(fdata.buffer)
                    if not getattr(data, "lazy_host", False):
                        fdata.accel_update_host()
                        fdata.accel_delete()
                else:
                    hipfft.extend_flags_buffer(fdata.data, fdata.indices(dets), flag_mask, extend, or_row=shflg)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = {"meta": [self.noise_model], "shared": [], "detdata": [self.det_data], "intervals": []}
        if self.shared_flags is not None:
            req["shared"].append(self.shared_flags)
        if self.det_flags is not None:
            req["detdata"].append(self.det_flags)
        return req

    def _provides(self):
        return {"detdata": [self.det_data]}
