"""Noise models: only what the map-maker reads -- ``freq``, ``psd``, ``detector_weight``
(reference: src/toast/noise.py:213-277, src/toast/noise_sim.py:20-143)."""

import os

import numpy as np


def name_UID(name, int64=False):
    """Reproducible integer of a name (reference src/toast/utils.py:635-652: the low bits of the md5 digest read as
    a little-endian integer)."""
    import hashlib

    ind = int.from_bytes(hashlib.md5(name.encode("utf-8")).digest(), byteorder="little")
    return np.uint64(ind & 0x7FFFFFFFFFFFFFFF) if int64 else np.uint32(ind & 0x7FFFFFFF)


class Noise:
    """PSD-based noise model of the local detectors of one observation (reference: src/toast/noise.py:17-277; arrays
    are plain float64 in Hz and K^2 s -- no astropy here).  ``psds`` / ``freqs`` are keyed by PSD name; a detector's
    noise is the ``mixmatrix``-weighted combination of PSDs (default: one PSD per detector under its own name)."""

    def __init__(self, detectors=(), freqs=None, psds=None, mixmatrix=None, indices=None, detweights=None, rate=None):
        freqs = {} if freqs is None else freqs
        psds = {} if psds is None else psds
        self._dets = list(sorted(detectors))
        if mixmatrix is None:
            self._keys = self._dets
            self._keys_for_dets = {x: [x] for x in self._dets}
            self._dets_for_keys = {x: [x] for x in self._dets}
            self._mixmatrix = {d: {d: 1.0} for d in self._dets}
        else:
            self._mixmatrix = {det: dict(mixmatrix[det]) for det in self._dets}
            keys = set()
            self._keys_for_dets, self._dets_for_keys = {}, {}
            for det in self._dets:
                self._keys_for_dets[det] = []
                for key, weight in self._mixmatrix[det].items():
                    keys.add(key)
                    self._dets_for_keys.setdefault(key, [])
                    if weight != 0:
                        self._keys_for_dets[det].append(key)
                        self._dets_for_keys[key].append(det)
            self._keys = list(sorted(keys))
        self._indices = {x: name_UID(x) for x in self._keys} if indices is None else dict(indices)
        self._freqs, self._psds, self._rates = {}, {}, {}
        for key in self._keys:
            f = np.array(freqs[key], dtype=np.float64)
            p = np.array(psds[key], dtype=np.float64)
            if p.shape[0] != f.shape[0]:
                raise ValueError("PSD length must match the number of frequencies")
            self._freqs[key], self._psds[key] = f, p
            # last frequency point should be Nyquist (noise.py:105); `rate` overrides (not in the reference)
            self._rates[key] = 2.0 * f[-1] if rate is None else float(rate)
        self._detweights = None if detweights is None else dict(detweights)

    #: reproduce the reference's first-call result of ``detector_weight`` (see ``_detector_weight``); off by
    #: default, TOAST_HIP_NOISE_REFERENCE_QUIRK=1 switches it on at import
    reference_first_call_quirk = os.environ.get("TOAST_HIP_NOISE_REFERENCE_QUIRK", "0") == "1"

    detectors = property(lambda self: self._dets)
    keys = property(lambda self: self._keys)
    mixing_matrix = property(lambda self: self._mixmatrix)

    def multiply_ntt(self, key, data):
        raise NotImplementedError("multiply_ntt not yet implemented")

    def multiply_invntt(self, key, data):
        raise NotImplementedError("multiply_invntt not yet implemented")

    def weight(self, det, key):
        """Mixing weight of PSD ``key`` in detector ``det`` (0 when absent)."""
        return self._mixmatrix[det].get(key, 0)

    def all_keys_for_dets(self, dets):
        keys = set()
        for det in dets:
            keys.update(self._keys_for_dets[det])
        return list(sorted(keys))

    def index(self, key):
        return self._indices[key]

    def freq(self, key):
        return self._freqs[key]

    def rate(self, key):
        return self._rates[key]

    def psd(self, key):
        return self._psds[key]

    def _detector_weight(self, det):
        """1 / (white-noise variance * rate) of every PSD, combined with the mixing matrix (noise.py:217-265): the
        white level is the median PSD over [0.45, 0.5] x rate, or over [0.2, 0.4] x rate when the end of the spectrum
        lies below half of its middle ([0.225, 0.275] x rate), i.e. a transfer-function roll-off; 0 for a PSD whose
        mid-band is zero (flagged in the noise model)."""
        if self._detweights is None:
            self._detweights = {d: 0.0 for d in self._dets}
            for k in self._keys:
                f, p, rate = self._freqs[k], self._psds[k], self._rates[k]

                def median_between(lo, hi):
                    first = np.searchsorted(f, rate * lo, side="left")
                    last = np.searchsorted(f, rate * hi, side="right")
                    if first == last:
                        first = max(0, first - 1)
                        last = min(f.size - 1, last + 1)
                    return np.median(p[first:last])

                noisevar_mid = median_between(0.225, 0.275)
                if noisevar_mid == 0:
                    invvar = 0.0
                else:
                    noisevar_end = median_between(0.45, 0.50)
                    noisevar = median_between(0.2, 0.4) if noisevar_end / noisevar_mid < 0.5 else noisevar_end
                    invvar = 1.0 / noisevar / rate
                for kdet in self._dets_for_keys[k]:
                    self._detweights[kdet] += self._mixmatrix[kdet][k] * invvar
                    last = kdet
            if Noise.reference_first_call_quirk and self._keys:
                # the reference's loop variable is also called `det` (noise.py:262-265): the call that computes the
                # table returns the weight of the LAST detector of the last key instead of the one asked for
                return self._detweights[last]
        return self._detweights[det]

    def detector_weight(self, det):
        return self._detector_weight(det)


class AnalyticNoise(Noise):
    """``psd = NET^2 (f^alpha + fknee^alpha) / (f^alpha + fmin^alpha)`` on a log frequency grid
    (reference: src/toast/noise_sim.py:88-112); ``detector_weight = 1 / (NET^2 rate)``
    (noise_sim.py:137-143)."""

    def __init__(self, detectors=(), rate=None, fmin=None, fknee=None, alpha=None, NET=None, indices=None):
        detectors = list(detectors)
        self._rate_d = {d: float(rate[d]) for d in detectors}
        self._net = {d: float(NET[d]) for d in detectors}
        self._fmin = {d: float(fmin[d]) for d in detectors}
        self._fknee = {d: float(fknee[d]) for d in detectors}
        self._alpha = {d: float(alpha[d]) for d in detectors}
        for d in detectors:
            if self._alpha[d] < 0.0:
                raise RuntimeError("alpha exponents should be positive in this formalism")
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
            a = self._alpha[d]
            fk, fm = self._fknee[d], self._fmin[d]
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
        super().__init__(detectors=detectors, freqs=freqs, psds=psds, indices=indices)

    def rate(self, det):
        return self._rate_d[det]

    def fmin(self, det):
        return self._fmin[det]

    def fknee(self, det):
        return self._fknee[det]

    def alpha(self, det):
        return self._alpha[det]

    def NET(self, det):
        return self._net[det]

    def _detector_weight(self, det):
        if self._net[det] == 0:
            return 0.0   # noise_sim.py:138-140
        return 1.0 / (self._net[det] ** 2) / self._rate_d[det]
