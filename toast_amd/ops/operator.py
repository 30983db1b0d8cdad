"""Operator base class (reference: src/toast/ops/operator.py:11-315)."""

from ..traits import Bool, TraitConfig


class Operator(TraitConfig):
    """An operator works on a ``Data`` object: ``exec`` (possibly many times, per detector
    set), then ``finalize``; ``apply`` = both.  ``requires()`` / ``provides()`` name the
    objects it reads / writes so a ``Pipeline`` can stage them on the accelerator."""

    timing = Bool(False, help="If True, print timing of exec()")
    timing_total = Bool(False, help="If True, print timing of finalize()")

    def _exec(self, data, detectors=None, **kwargs):
        raise NotImplementedError("Fell through to Operator base class")

    def exec(self, data, detectors=None, **kwargs):
        if self.enabled:
            self._exec(data, detectors=detectors, **kwargs)

    def _finalize(self, data, **kwargs):
        raise NotImplementedError("Fell through to Operator base class")

    def finalize(self, data, **kwargs):
        if self.enabled:
            return self._finalize(data, **kwargs)
        return None

    def apply(self, data, detectors=None, **kwargs):
        self.exec(data, detectors=detectors, **kwargs)
        return self.finalize(data, **kwargs)

    def _requires(self):
        raise NotImplementedError("Fell through to Operator base class")

    def requires(self):
        """Dict of lists under the keys global / meta / detdata / shared / intervals
        (operator.py:209-232)."""
        return self._with_all_keys(self._requires())

    def _provides(self):
        raise NotImplementedError("Fell through to Operator base class")

    def provides(self):
        return self._with_all_keys(self._provides())

    @staticmethod
    def _with_all_keys(d):
        out = {k: [] for k in ("global", "meta", "detdata", "shared", "intervals")}
        for k, v in d.items():
            out[k] = [x for x in v if x is not None]
        return out
