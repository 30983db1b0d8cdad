"""``Combine``: arithmetic on two detdata objects, ``result = first (op) second`` (reference:
src/toast/ops/arithmetic.py:14-185).  The reference runs per-detector NumPy expressions on the
host; add / subtract of float64 buffers that are current on the device run as one whole-buffer
kernel there (toast_hip_vec_axpby_dev), everything else follows the reference on the host."""

import numpy as np

from ..traits import Int, Unicode
from .operator import Operator


class Combine(Operator):
    API = Int(0, help="Internal interface version for this operator")
    op = Unicode(None, allow_none=True,
                 help="Operation on the timestreams: 'subtract', 'add', 'multiply', or 'divide'")
    first = Unicode(None, allow_none=True, help="The first detdata object")
    second = Unicode(None, allow_none=True, help="The second detdata object")
    result = Unicode(None, allow_none=True, help="The resulting detdata object")

    def _validate_op(self, val):
        if val is not None and val not in ("add", "subtract", "multiply", "divide"):
            raise RuntimeError("op must be one of the 4 allowed strings")
        return val

    def _exec(self, data, detectors=None, **kwargs):
        for name in ("first", "second", "result", "op"):
            if getattr(self, name) is None:
                raise RuntimeError(f"The {name} trait must be set before calling exec")
        for ob in data.obs:
            local_dets = ob.select_local_detectors(detectors)
            if len(local_dets) == 0 or self.first not in ob.detdata or self.second not in ob.detdata:
                continue
            fd, sd = ob.detdata[self.first], ob.detdata[self.second]
            if fd.units != sd.units:
                raise NotImplementedError("unit conversion between detdata objects is outside the hot path")
            dets = sorted(set(fd.detectors) & set(sd.detectors))
            if self.result not in (self.first, self.second):
                ob.detdata.ensure(self.result, sample_shape=fd.detector_shape[1:], dtype=fd.dtype,
                                  detectors=fd.detectors, create_units=fd.units)
            rd = ob.detdata[self.result]
            if self._exec_device(fd, sd, rd, dets):
                continue
            for d in dets:
                a, b = fd[d], sd[d]
                if self.op == "add":
                    rd[d] = a + b
                elif self.op == "subtract":
                    rd[d] = a - b
                elif self.op == "multiply":
                    rd[d] = a * b
                else:
                    rd[d] = a / b

    def _exec_device(self, fd, sd, rd, dets):
        """``result = first +- second`` on the device for the common detectors, one launch per run
        of rows that are adjacent in all three buffers; False when it does not apply."""
        if self.op not in ("add", "subtract") or fd.dtype != np.float64 or sd.dtype != np.float64:
            return False
        if not (fd.accel_in_use() and sd.accel_in_use()) or len(dets) == 0:
            return False
        if fd.detector_shape != sd.detector_shape or rd.detector_shape != fd.detector_shape:
            return False
        from .. import capi
        from ..accel import accel_device_ptr

        if rd is not fd and rd is not sd:
            if not rd.accel_exists():
                rd.accel_create(self.result)
            elif not rd.accel_in_use():
                rd.accel_update_device()   # rows of other detectors keep their values
        row = int(np.prod(fd.detector_shape))
        sign = 1.0 if self.op == "add" else -1.0
        fp, sp, rp = accel_device_ptr(fd.buffer), accel_device_ptr(sd.buffer), accel_device_ptr(rd.buffer)
        rows = sorted(zip(fd.indices(dets).tolist(), sd.indices(dets).tolist(), rd.indices(dets).tolist()))
        i = 0
        while i < len(rows):
            j = i
            while j + 1 < len(rows) and all(rows[j + 1][k] == rows[j][k] + 1 for k in range(3)):
                j += 1
            n = (j - i + 1) * row
            f0, s0, r0 = (rows[i][k] * row * 8 for k in range(3))
            if rd is fd:
                capi.dev.vec_axpby(n, sign, sp + s0, 1.0, fp + f0)          # first = first +- second
            elif rd is sd:
                capi.dev.vec_axpby(n, 1.0, fp + f0, sign, sp + s0)          # second = first +- second
            else:
                capi.dev.copy(rp + r0, fp + f0, n * 8)
                capi.dev.vec_axpby(n, sign, sp + s0, 1.0, rp + r0)
            i = j + 1
        rd.accel_used(True)
        return True

    def _finalize(self, data, **kwargs):
        return None

    def _supports_accel(self):
        # the host branch reads through DetectorData's lazy host coherence, so the operator can
        # sit in a device pipeline without forcing its inputs off the device beforehand
        return True

    def _requires(self):
        req = {"detdata": [self.first, self.second]}
        if self.result is not None and self.result not in (self.first, self.second):
            req["detdata"].append(self.result)
        return req

    def _provides(self):
        prov = {"detdata": []}
        if self.result not in (self.first, self.second):
            prov["detdata"].append(self.result)
        return prov
