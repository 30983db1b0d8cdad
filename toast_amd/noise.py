"""Noise models: only what the map-maker reads -- ``freq``, ``psd``, ``detector_weight``
(reference: src/toast/noise.py:213-277, src/toast/noise_sim.py:20-143)."""

import numpy as np


class Noise:
    """Generic PSD-based model (reference: src/toast/noise.py; no mixing matrix: one PSD per
    detector)."""

    def __init__(self, detectors, freqs, psds, rate=None):
        self.detectors = list(detectors)
        self._freqs = {d: np.asarray(freqs[d], dtype=np.float64) for d in self.detectors}
        self._psds = {d: np.asarray(psds[d], dtype=np.float64) for d in self.detectors}
        self._rate = rate
        self._weights = {}

    def freq(self, det):
        return self._freqs[det]

    def psd(self, det):
        return self._psds[det]

    def rate(self, det):
        return 2.0 * self._freqs[det][-1] if self._rate is None else self._rate

    def detector_weight(self, det):
        """1 / (white-noise variance * rate), the white level measured as the reference does
        (src/toast/noise.py:216-262): the median PSD over [0.45, 0.5] x rate, or over
        [0.2, 0.4] x rate when the end of the spectrum lies below half of its middle
        ([0.225, 0.275] x rate), i.e. a transfer-function roll-off; 0 for a detector whose mid-band
        PSD is zero (flagged in the noise model)."""
        if det not in self._weights:
            f, p, rate = self._freqs[det], self._psds[det], self.rate(det)

            def median_between(lo, hi):
                first = np.searchsorted(f, rate * lo, side="left")
                last = np.searchsorted(f, rate * hi, side="right")
                if first == last:
                    first = max(0, first - 1)
                    last = min(f.size - 1, last + 1)
                return np.median(p[first:last])

            noisevar_mid = median_between(0.225, 0.275)
            if noisevar_mid == 0:
                self._weights[det] = 0.0
            else:
                noisevar_end = median_between(0.45, 0.50)
                noisevar = median_between(0.2, 0.4) if noisevar_end / noisevar_mid < 0.5 else noisevar_end
                self._weights[det] = 1.0 / noisevar / rate
        return self._weights[det]


class AnalyticNoise(Noise):
    """``psd = NET^2 (f^alpha + fknee^alpha) / (f^alpha + fmin^alpha)`` on a log frequency grid
    (reference: src/toast/noise_sim.py:88-112); ``detector_weight = 1 / (NET^2 rate)``
    (noise_sim.py:137-143)."""

    def __init__(self, rate, fmin, detectors, fknee, alpha, NET):
        self._rate_d = {d: float(rate[d]) for d in detectors}
        self._net = {d: float(NET[d]) for d in detectors}
        freqs, psds = {}, {}
        for d in detectors:
            r = self._rate_d[d]
            nyq = r / 2.0
            tempfreq = []
            cur = 1.0e-9
            while cur < nyq:  # noise_sim.py:88-96: 1e-9 * 1.4^k ... Nyquist
                tempfreq.append(cur)
                cur *= 1.4
            tempfreq.append(nyq)
            f = np.array(tempfreq, dtype=np.float64)
            a = float(alpha[d])
            fk, fm = float(fknee[d]), float(fmin[d])
            if fk > 0.0 and fk < fm:
                raise RuntimeError("If knee frequency is non-zero, it must be greater than f_min")
            if fk > 0.0:
                ktemp = np.power(fk, a)
                mtemp = np.power(fm, a)
                temp = np.power(f, a)
                psds[d] = (temp + ktemp) / (temp + mtemp) * self._net[d] ** 2
            else:
                psds[d] = np.ones_like(f) * self._net[d] ** 2   # white (noise_sim.py:109-111)
            freqs[d] = f
        super().__init__(detectors, freqs, psds)

    def rate(self, det):
        return self._rate_d[det]

    def NET(self, det):
        return self._net[det]

    def detector_weight(self, det):
        if self._net[det] == 0:
            return 0.0   # noise_sim.py:138-140
        return 1.0 / (self._net[det] ** 2) / self._rate_d[det]
