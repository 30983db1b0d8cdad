"""Pipeline: run a list of operators over detector sets and stage their data on the
accelerator (reference: src/toast/ops/pipeline.py:22-389)."""

import os

from ..accel import accel_enabled
from ..traits import Bool, ImplementationType, Int, List
from .operator import Operator


class _SetDict(dict):
    KEYS = ("global", "meta", "detdata", "shared", "intervals")

    def __init__(self, other=None):
        super().__init__({k: set() for k in self.KEYS})
        if other:
            for k, v in other.items():
                self[k] = set(v)

    def __isub__(self, other):
        for k in self.KEYS:
            self[k] -= set(other.get(k, ()))
        return self

    def __ior__(self, other):
        for k in self.KEYS:
            self[k] |= set(other.get(k, ()))
        return self

    def __iand__(self, other):
        for k in self.KEYS:
            self[k] &= set(other.get(k, ()))
        return self

    def is_empty(self):
        return all(len(v) == 0 for v in self.values())


def uncached_detector_sets():
    """Detector sets for passes that recompute pointing into scratch buffers: the reference's
    ["SINGLE"] on the host, ["BATCH"] on the accelerator (one-detector launches cannot fill
    256 CUs)."""
    return ["BATCH"] if accel_enabled() else ["SINGLE"]


class Pipeline(Operator):
    """Chain of operators.  ``detector_sets`` = ["ALL"] (one pass over all detectors),
    ["SINGLE"] (one pass per detector: pointing is recomputed into recycled one-detector
    buffers) or explicit lists.  With an accelerator and all (or, ``use_hybrid``, some)
    operators supporting it, required objects are created/updated on the device before each
    operator and outputs are copied back and freed at finalize (pipeline.py:208-303)."""

    API = Int(0, help="Internal interface version for this operator")
    operators = List([], help="List of Operator instances to run.")
    detector_sets = List(["ALL"], help="List of detector sets: 'ALL', 'SINGLE', 'BATCH' or lists of names")
    batch_size = Int(64, help="Detectors per pass for detector_sets=['BATCH'] (not a reference trait)")
    use_hybrid = Bool(True, help="Should the pipeline be allowed to use the GPU when it has some cpu-only operators.")

    def _validate_operators(self, ops):
        for op in ops:
            if not isinstance(op, Operator):
                raise RuntimeError("operators must be a list of Operator instances or None")
        return ops

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._staged_data = None
        self._unstaged_data = None

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        if len(self.operators) == 0:
            return
        self._staged_data = None
        self._unstaged_data = None
        pipe_accel = self._pipe_accel(use_accel)
        if pipe_accel:
            self._staged_data = _SetDict()
            self._unstaged_data = _SetDict()
            self._protect = set()
            if hasattr(data, "_protected"):
                data._protected.append(self._protect)
        det_mask = None
        for op in self.operators:
            if op.has_trait("det_mask"):
                if det_mask is None:
                    det_mask = op.det_mask
                elif op.det_mask != det_mask:
                    raise RuntimeError(
                        "All operators in a Pipeline which use a det_mask must have the same mask value "
                        f"({op.det_mask} != {det_mask}) in ops {[x.name for x in self.operators]}"
                    )
        if det_mask is None:
            det_mask = 0
        sets = self.detector_sets
        if len(sets) == 1 and sets[0] == "ALL":
            for op in self.operators:
                self._exec_operator(op, data, detectors, pipe_accel)
        elif len(sets) == 1 and sets[0] == "SINGLE":
            all_local = data.all_local_detectors(selection=detectors, flagmask=det_mask)
            if len(all_local) == 0:
                all_local = [None]
            for det in all_local:
                dets = [] if det is None else [det]
                for op in self.operators:
                    self._exec_operator(op, data, dets, pipe_accel)
        elif len(sets) == 1 and sets[0] == "BATCH":
            # SINGLE semantics (scratch pointing buffers recycled between passes) with enough
            # detectors per pass to fill the GPU: 64 detectors x 64 B/sample of pointing scratch
            all_local = data.all_local_detectors(selection=detectors, flagmask=det_mask)
            n = max(1, int(os.environ.get("TOAST_HIP_POINTING_BATCH", self.batch_size)))
            for i in range(0, max(len(all_local), 1), n):
                dets = all_local[i:i + n]
                for op in self.operators:
                    self._exec_operator(op, data, dets, pipe_accel)
        else:
            check = None if detectors is None else set(detectors)
            for det_set in sets:
                selected = list(det_set) if check is None else [d for d in det_set if d in check]
                if len(selected) == 0:
                    continue
                for op in self.operators:
                    self._exec_operator(op, data, selected, pipe_accel)

    def _exec_operator(self, op, data, detectors, pipe_accel):
        """Run one operator, moving data to / from the device first (pipeline.py:208-263)."""
        run_accel = bool(pipe_accel and op.supports_accel())
        if self._staged_data is not None:
            requires = _SetDict(op.requires())
            if run_accel:
                requires -= self._staged_data
                self._protect |= requires["detdata"] | set(op.provides().get("detdata", ()))
                self._protect |= {"global:" + k for k in set(requires["global"]) | set(op.provides().get("global", ()))}
                data.accel_create(requires)
                data.accel_update_device(requires)
                self._unstaged_data -= requires
                self._staged_data |= requires
                self._staged_data |= op.provides()
                self._unstaged_data -= op.provides()
            else:
                requires &= self._staged_data
                data.accel_update_host(requires)
                self._staged_data -= requires
                self._unstaged_data |= requires
                self._unstaged_data |= op.provides()
                run_accel = None
        op.exec(data, detectors=detectors, use_accel=run_accel)

    def _finalize(self, data, use_accel=None, **kwargs):
        pipe_accel = self._pipe_accel(use_accel)
        result = []
        for op in self.operators:
            use_accel_op = bool(pipe_accel and op.supports_accel())
            result.append(op.finalize(data, use_accel=use_accel_op, **kwargs))
        if self._staged_data is not None:
            # The reference copies back only the pruned provides() (pipeline.py:280-295), which
            # leaves intermediate products that stay allocated on the host (e.g. cached pixels /
            # weights with full_pointing=True) stale there once the device copies are deleted.
            # We bring back everything any operator provided.
            provides = _SetDict()
            for op in self.operators:
                provides |= op.provides()
            provides &= self._staged_data
            if getattr(data, "lazy_host", False):
                # detector data (the bulk) stays resident and device-current: the host copy is
                # refreshed on access (DetectorData.data) or on eviction.  Small objects follow
                # the reference: outputs copied back, device copies freed.
                provides["detdata"] = set()
                self._staged_data["detdata"] = set()
                # maps too (PixelData.data / .raw copy back on access): the next operator of the map-making chain
                # (covariance inversion, covariance_apply, the solver) finds them where the kernels left them
                from ..pixels import PixelData

                lazy = {k for k in self._staged_data.get("global", ()) if isinstance(data._internal.get(k), PixelData)}
                if "global" in provides:
                    provides["global"] = set(provides["global"]) - lazy
                if "global" in self._staged_data:
                    self._staged_data["global"] = set(self._staged_data["global"]) - lazy
            if getattr(self, "_protect", None) is not None and hasattr(data, "_protected"):
                data._protected[:] = [s for s in data._protected if s is not self._protect]
                self._protect = None
            data.accel_update_host(provides)
            data.accel_delete(self._staged_data)
            self._staged_data = None
            self._unstaged_data = None
        return result

    def _pipe_accel(self, use_accel):
        if (use_accel is None) and accel_enabled():
            return self._supports_accel_partial() if self.use_hybrid else self._supports_accel()
        return bool(use_accel)

    def _requires(self):
        req = _SetDict()
        for op in reversed(self.operators):
            req -= op.provides()
            req |= op.requires()
        return {k: list(v) for k, v in req.items()}

    def _provides(self):
        prov = _SetDict()
        for op in self.operators:
            prov -= op.requires()
            prov |= op.provides()
        return {k: list(v) for k, v in prov.items()}

    def _implementations(self):
        impls = {ImplementationType.DEFAULT, ImplementationType.COMPILED}
        for op in self.operators:
            impls.intersection_update(op.implementations())
        return list(impls)

    def _supports_accel(self):
        return all(op.supports_accel() for op in self.operators)

    def _supports_accel_partial(self):
        return any(op.supports_accel() for op in self.operators)

    def __str__(self):
        return f"Pipeline{[op.__class__.__qualname__ for op in self.operators]}"
