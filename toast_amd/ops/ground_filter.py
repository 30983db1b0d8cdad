"""GroundFilter: fit and subtract ground-synchronous templates (Legendre trend in time, Legendre
or binned functions of azimuth, optionally separate for left- and right-going scans) from every
detector timestream (reference: src/toast/ops/groundfilter.py:57-537).

The reference builds the templates with libtoast's `legendre`, then loops over detectors on the
host: `bin_proj`, `bin_invcov`, a dense solve, `add_templates`.  Here the templates of an
observation live in HBM once and all detectors pass through three launches
(toast_hip_legendre_templates_dev / _template_select_dev, toast_hip_template_fit_dev,
toast_hip_template_subtract_dev: csrc/ground_filter.hip); only the n_template x n_template
systems are solved on the host (NumPy, stacked over detectors -- the reference's own
np.linalg calls).
"""

import re

import numpy as np

from ..accel import (
    accel_data_create,
    accel_data_delete,
    accel_data_update_device,
    accel_data_update_host,
    accel_device_ptr,
    accel_enabled,
    native,
)
from ..data import defaults
from ..traits import Bool, Float, Int, Unicode
from .operator import Operator


class GroundFilter(Operator):
    """Operator that applies ground template filtering to azimuthal scans."""

    API = Int(0, help="Internal interface version for this operator")
    det_data = Unicode(defaults.det_data, help="Observation detdata key")
    pattern = Unicode(".*", allow_none=True, help="Regex pattern to match against detector names")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    det_mask = Int(defaults.det_mask_invalid, help="Bit mask value for per-detector flagging")
    shared_flags = Unicode(defaults.shared_flags, allow_none=True, help="Observation shared key for telescope flags")
    shared_flag_mask = Int(defaults.shared_mask_invalid, help="Bit mask value for optional shared flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for flags to use")
    det_flag_mask = Int(defaults.det_mask_invalid, help="Bit mask value for detector sample flagging")
    ground_flag_mask = Int(defaults.det_mask_invalid, help="Bit mask to use when adding flags based on filter failures")
    azimuth = Unicode(defaults.azimuth, allow_none=True, help="Observation shared key for Azimuth")
    boresight_azel = Unicode(defaults.boresight_azel, allow_none=True, help="Observation shared key for boresight Az/El")
    trend_order = Int(5, allow_none=True, help="Order of a Legendre polynomial to fit along with the ground template")
    filter_order = Int(5, allow_none=True, help="Order of a Legendre polynomial to fit as a function of azimuth")
    bin_width = Float(None, allow_none=True, help="Azimuthal bin width of ground filter [rad]")
    detrend = Bool(False, help="Subtract the fitted trend along with the ground template")
    split_template = Bool(False, help="Apply a different template for left and right scans")
    leftright_interval = Unicode(defaults.throw_leftright_interval, help="Intervals for left-to-right scans")
    rightleft_interval = Unicode(defaults.throw_rightleft_interval, help="Intervals for right-to-left scans")

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.nsingular = 0
        self.ngood = 0
        self.rcondsum = 0.0

    # ------------------------------------------------------------------ templates (device)
    def _azimuth(self, obs):
        if self.azimuth is not None and self.azimuth in obs.shared:
            return np.array(obs.shared[self.azimuth].data, dtype=np.float64)
        if self.boresight_azel is not None and self.boresight_azel in obs.shared:
            # az = 2 pi - phi of the boresight direction (groundfilter.py:285-289)
            q = np.asarray(obs.shared[self.boresight_azel].data, dtype=np.float64)
            x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
            dx = 2 * (x * z + w * y)
            dy = 2 * (y * z - w * x)
            phi = np.arctan2(dy, dx)
            phi[phi < 0] += 2 * np.pi        # qa.to_iso_angles: phi in [0, 2 pi) (toast_math_qarray.cpp:1184-1189)
            return 2 * np.pi - phi
        raise RuntimeError("Failed to get boresight azimuth from TOD.  Perhaps it is not ground TOD?")

    def build_templates(self, obs):
        """Template hierarchy of one observation, built in HBM (groundfilter.py:259-331).  Returns
        the [n_template, n_samp] host array that keys the device buffer (the host copy is not
        filled; ``accel_data_update_host`` fetches it when somebody wants to look)."""
        from .. import capi

        D = capi.dev
        n = obs.n_local_samples
        az = self._azimuth(obs)
        # phase maps azimuth to [-1, 1], across the zero meridian if needed (groundfilter.py:296-312)
        azmin, azmax = np.amin(az), np.amax(az)
        while azmin < 0:
            azmin += 2 * np.pi
            azmax += 2 * np.pi
        if azmax - azmin > 2 * np.pi:
            azmin, azmax = 0, 2 * np.pi
            az %= 2 * np.pi
        phase = (az - azmin) / (azmax - azmin) * 2 - 1
        x = np.arange(n) / n * 2 - 1
        n_trend = self.trend_order if self.trend_order is not None else 0
        n_poly = self.filter_order + 1 if self.filter_order is not None else 0
        bins = np.zeros(0, dtype=np.int64)
        ibin = None
        if self.bin_width is not None:
            ibin = (az // self.bin_width).astype(np.int32)
            bins, counts = np.unique(ibin, return_counts=True)
            if self.filter_order is not None:
                # one bin is dropped: the others are relative to it, which breaks the degeneracy
                # with the polynomial templates (groundfilter.py:243-249, with the hit counts the
                # reference forgot to request from np.unique)
                keep = np.ones(bins.size, dtype=bool)
                keep[np.argmax(counts)] = False
                bins = bins[keep]
        mult = 2 if self.split_template else 1
        nt = n_trend + mult * (n_poly + bins.size)
        if nt == 0:
            raise RuntimeError("GroundFilter: no templates requested")
        templates = np.empty((nt, n), dtype=np.float64)
        scratch = {"x": x, "phase": phase}
        direction = None
        if self.split_template:
            direction = np.zeros(n, dtype=np.int32)
            for ival in obs.intervals[self.leftright_interval]:
                direction[ival.first:ival.last] = 1
            for ival in obs.intervals[self.rightleft_interval]:
                direction[ival.first:ival.last] = 2
            scratch["direction"] = direction
            scratch["row"] = np.empty(max(n_poly, 1) * n, dtype=np.float64)
        if ibin is not None:
            scratch["ibin"] = ibin
        name = f"{self.name}_templates"
        accel_data_create(templates, name, owner=self)
        for key, arr in scratch.items():
            accel_data_create(arr, f"{self.name}_{key}", owner=self)
            if key != "row":
                accel_data_update_device(arr, f"{self.name}_{key}")
        t_ptr = accel_device_ptr(templates)
        row_ptr = lambda r: t_ptr + 8 * n * r  # noqa: E731
        if n_trend > 0:
            # the offset is not part of the trend: it belongs to the ground template
            D.legendre_templates(accel_device_ptr(x), n, 1, n_trend + 1, row_ptr(0))
        cursor = n_trend
        if n_poly > 0:
            if not self.split_template:
                D.legendre_templates(accel_device_ptr(phase), n, 0, n_poly, row_ptr(cursor))
                cursor += n_poly
            else:
                tmp = accel_device_ptr(scratch["row"])
                D.legendre_templates(accel_device_ptr(phase), n, 0, n_poly, tmp)
                for r in range(n_poly):
                    for value in (1, 2):  # without the left-right samples, without the right-left ones
                        D.template_select(tmp + 8 * n * r, accel_device_ptr(direction), value, False, n, row_ptr(cursor))
                        cursor += 1
        for b in bins:
            if not self.split_template:
                D.template_select(0, accel_device_ptr(ibin), int(b), True, n, row_ptr(cursor))
                cursor += 1
            else:
                tmp = accel_device_ptr(scratch["row"])
                D.template_select(0, accel_device_ptr(ibin), int(b), True, n, tmp)
                for value in (1, 2):
                    D.template_select(tmp, accel_device_ptr(direction), value, False, n, row_ptr(cursor))
                    cursor += 1
        assert cursor == nt
        native().accel_synchronize()
        for key, arr in scratch.items():
            accel_data_delete(arr, f"{self.name}_{key}")
        return templates

    # ------------------------------------------------------------------ fit (host, tiny systems)
    def solve(self, proj, gram_common, gram_flagged, n_good):
        """coeff[d] = cov_d proj_d with cov_d the inverse (pseudo-inverse when rcond <= 1e-6) of
        invcov_d = gram_common - gram_flagged[d] (groundfilter.py:334-381); None rows where a
        detector has no good sample."""
        invcov = gram_common[None, :, :] - gram_flagged
        coeff = np.zeros_like(proj)
        ok = n_good > 0
        sel = np.nonzero(ok)[0]
        if sel.size > 0:
            # stacked over detectors: the same np.linalg calls as the reference makes one by one
            with np.errstate(divide="ignore"):
                rcond = 1 / np.linalg.cond(invcov[sel])
            self.rcondsum += float(np.sum(rcond))
            regular = rcond > 1e-6
            self.ngood += int(np.count_nonzero(regular))
            self.nsingular += int(np.count_nonzero(~regular))
            if np.any(regular):
                cov = np.linalg.inv(invcov[sel[regular]])
                coeff[sel[regular]] = np.einsum("dij,dj->di", cov, proj[sel[regular]])
            for d in sel[~regular]:
                cov = np.linalg.pinv(invcov[d], rcond=1e-12, hermitian=True)
                coeff[d] = np.dot(cov, proj[d])
        return coeff, ok

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        from .. import capi

        if not accel_enabled():
            raise RuntimeError("GroundFilter needs the HIP library and an assigned device (no host path)")
        D = capi.dev
        self.nsingular, self.ngood, self.rcondsum = 0, 0, 0.0
        pat = re.compile(self.pattern if self.pattern is not None else ".*")
        for obs in data.obs:
            dets = [d for d in obs.select_local_detectors(detectors, flagmask=self.det_mask) if pat.match(d) is not None]
            if len(dets) == 0:
                continue
            n = obs.n_local_samples
            templates = self.build_templates(obs)
            nt = templates.shape[0]
            dd = obs.detdata[self.det_data]
            made_resident = False
            if not dd.accel_in_use():
                if not dd.accel_exists():
                    dd.accel_create(self.det_data)
                dd.accel_update_device()
                made_resident = True
            f_ptr, f_idx, fd = 0, None, None
            if self.det_flags is not None:
                fd = obs.detdata[self.det_flags]
                if not fd.accel_in_use():
                    if not fd.accel_exists():
                        fd.accel_create(self.det_flags)
                    fd.accel_update_device()
                f_ptr, f_idx = accel_device_ptr(fd.buffer), fd.indices(dets)
            s_ptr = 0
            common_good = np.ones(n, dtype=bool)
            if self.shared_flags is not None:
                sf = obs.shared[self.shared_flags]
                if not sf.accel_in_use():
                    if not sf.accel_exists():
                        sf.accel_create(self.shared_flags)
                    sf.accel_update_device()
                s_ptr = accel_device_ptr(sf.data)
                common_good = (sf.data & self.shared_flag_mask) == 0
            n_det = len(dets)
            proj = np.zeros((n_det, nt))
            gram_common = np.zeros((nt, nt))
            gram_flagged = np.zeros((n_det, nt, nt))
            n_flagged = np.zeros(n_det, dtype=np.int64)
            outs = {"proj": proj, "gram": gram_common, "dgram": gram_flagged, "nflag": n_flagged}
            for key, arr in outs.items():
                accel_data_create(arr, f"{self.name}_{key}", owner=self)
            D.template_fit(accel_device_ptr(templates), nt, n, dd.indices(dets), accel_device_ptr(dd.buffer), f_idx, f_ptr,
                           self.det_flag_mask, s_ptr, self.shared_flag_mask, accel_device_ptr(proj),
                           accel_device_ptr(gram_common), accel_device_ptr(gram_flagged), accel_device_ptr(n_flagged))
            native().accel_synchronize()
            for key, arr in outs.items():
                accel_data_update_host(arr, f"{self.name}_{key}")
                accel_data_delete(arr, f"{self.name}_{key}")
            # good samples per detector: the commonly good ones minus those only this detector flags
            n_good = int(np.count_nonzero(common_good)) - n_flagged
            coeff, ok = self.solve(proj, gram_common, gram_flagged, n_good)
            for i, det in enumerate(dets):
                if not ok[i]:
                    # all samples flagged: mark the detector (groundfilter.py:474-480)
                    cur = obs.local_detector_flags.get(det, 0)
                    obs.update_local_detector_flags({det: cur | self.ground_flag_mask})
            fit_dets = [det for i, det in enumerate(dets) if ok[i]]
            if len(fit_dets) > 0:
                offset = 0 if self.detrend else (self.trend_order or 0)
                cf = np.ascontiguousarray(coeff[ok])
                accel_data_create(cf, f"{self.name}_coeff", owner=self)
                accel_data_update_device(cf, f"{self.name}_coeff")
                D.template_subtract(accel_device_ptr(templates), nt, offset, n, dd.indices(fit_dets),
                                    accel_device_ptr(dd.buffer), accel_device_ptr(cf))
                native().accel_synchronize()
                accel_data_delete(cf, f"{self.name}_coeff")
            self.coefficients = {det: coeff[i] for i, det in enumerate(dets) if ok[i]}
            accel_data_delete(templates, f"{self.name}_templates")
            if made_resident and not getattr(data, "lazy_host", False):
                dd.accel_update_host()
                dd.accel_delete()
            else:
                dd.accel_used(True)
        comm = data.comm
        if comm.comm_world is not None:
            self.nsingular = int(comm.allreduce_scalar(self.nsingular, op="sum"))
            self.ngood = int(comm.allreduce_scalar(self.ngood, op="sum"))
            self.rcondsum = float(comm.allreduce_scalar(self.rcondsum, op="sum"))

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = {"shared": [], "detdata": [self.det_data], "intervals": []}
        for key in (self.shared_flags, self.azimuth, self.boresight_azel):
            if key is not None:
                req["shared"].append(key)
        if self.det_flags is not None:
            req["detdata"].append(self.det_flags)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        return {"detdata": [self.det_data]}
