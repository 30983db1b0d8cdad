"""Noise models: only what the map-maker reads -- ``freq``, ``psd``, ``detector_weight``
(reference: src/toast/noise.py:213-277, src/toast/noise_sim.py:20-143)."""

import numpy as np


class Noise:
    """Generic PSD-based model.  ``detector_weight`` = 1 / (white level * rate) with the white
    level = median PSD over [0.45, 0.5] x rate (reference: src/toast/noise.py:213-277)."""

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
        if det not in self._weights:
            f, p, rate = self._freqs[det], self._psds[det], self.rate(det)
            first = np.searchsorted(f, rate * 0.45, side="left")
            last = np.searchsorted(f, rate * 0.50, side="right")
            if first == last:
                first = max(0, first - 1)
                last = min(f.size, last + 1)
            noisevar = np.median(p[first:last])
            self._weights[det] = 1.0 / (noisevar * rate)
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
            ktemp = np.power(float(fknee[d]), a)
            mtemp = np.power(float(fmin[d]), a)
            temp = np.power(f, a)
            psds[d] = (temp + ktemp) / (temp + mtemp) * self._net[d] ** 2
            freqs[d] = f
        super().__init__(detectors, freqs, psds)

    def rate(self, det):
        return self._rate_d[det]

    def NET(self, det):
        return self._net[det]

    def detector_weight(self, det):
        return 1.0 / (self._net[det] ** 2 * self._rate_d[det])
