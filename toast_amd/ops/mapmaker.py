"""MapMaker: solve for template amplitudes (destriping) and bin the cleaned signal.

Reference: src/toast/ops/mapmaker.py:37-790 (MapMaker) and
src/toast/ops/mapmaker_templates.py:423-1125 (SolveAmplitudes, ApplyAmplitudes).  The
orchestration is reproduced for the configuration used on the hot path: covariance + hits
from the same pointing, RHS, PCG on the Offset amplitudes, template subtraction, final BinMap.
"""

import numpy as np

from ..data import defaults
from ..pixels import PixelData
from ..traits import Bool, Float, Instance, Int, Unicode
from .arithmetic import Combine
from .mapmaker_ops import Copy, CovarianceAndHits, Delete, ScanMask
from .mapmaker_solve import SolverLHS, SolverRHS, solve
from .operator import Operator
from .pipeline import Pipeline, uncached_detector_sets


class ApplyAmplitudes(Operator):
    """``out = det_data (-|+|*|/) M a`` (mapmaker_templates.py:1128-1320, subtract only)."""

    API = Int(0, help="Internal interface version for this operator")
    op = Unicode("subtract", help="Operation on the timestreams: 'subtract' or 'add'")
    amplitudes = Unicode(None, allow_none=True, help="Data key for template amplitudes")
    template_matrix = Instance(klass=Operator, help="This must be an instance of a template matrix operator")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    output = Unicode(None, allow_none=True, help="Observation detdata key for output (default: in place)")

    def _exec(self, data, detectors=None, **kwargs):
        if self.op not in ("subtract", "add", "multiply", "divide"):
            raise RuntimeError("op must be one of 'subtract', 'add', 'multiply', 'divide'")
        temp = f"{self.name}_temp"
        if self.output is not None:
            # the input is copied first, then overwritten (mapmaker_templates.py:1231-1236)
            Pipeline(operators=[Copy(detdata=[(self.det_data, self.output)])]).apply(data, detectors=detectors)
        tm = self.template_matrix.duplicate()
        tm.amplitudes = self.amplitudes
        tm.transpose = False
        tm.det_data = temp
        combine = Combine(op=self.op, first=self.det_data, second=temp,
                          result=self.det_data if self.output is None else self.output)
        # The reference projects and combines one detector at a time (Pipeline SINGLE,
        # mapmaker_templates.py:1252-1260); here all detectors go through one batched template
        # kernel and one device Combine, the projected timestreams never leave the GPU.
        Pipeline(operators=[tm, combine]).apply(data, detectors=detectors)
        Delete(detdata=[temp]).apply(data)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        return {"global": [self.amplitudes], "detdata": [self.det_data]}

    def _provides(self):
        return {"detdata": [self.det_data if self.output is None else self.output]}


class MapMaker(Operator):
    """Generalised destriping map-maker.

    Products (``<name>_hits``, ``_cov``, ``_rcond``, ``_map``, ``_amplitudes``) are stored in
    ``data``.  With no templates this reduces to CovarianceAndHits + BinMap."""

    API = Int(0, help="Internal interface version for this operator")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    convergence = Float(1.0e-12, help="Relative convergence limit")
    iter_min = Int(3, help="Minimum number of iterations")
    iter_max = Int(100, help="Maximum number of iterations")
    solve_rcond_threshold = Float(1.0e-8, help="When solving, minimum value for inverse pixel condition number cut.")
    map_rcond_threshold = Float(1.0e-8, help="For final map, minimum value for inverse pixel condition number cut.")
    binning = Instance(klass=Operator, help="Binning operator used for solving template amplitudes")
    template_matrix = Instance(klass=Operator, help="This must be an instance of a template matrix operator")
    map_binning = Instance(klass=Operator, help="Binning operator for final map making (default: solver binning)")
    keep_solver_products = Bool(False, help="If True, keep the map domain solver products in data")
    keep_final_products = Bool(True, help="If True, keep the map domain products in data after write")
    save_cleaned = Bool(False, help="If True, save the template-subtracted detector timestreams")
    overwrite_cleaned = Bool(False, help="If True and save_cleaned is True, overwrite the input data")
    reset_pix_dist = Bool(False, help="Clear any existing pixel distribution.")
    fused_lhs = Bool(True, help="Let SolverLHS use the fused device-resident kernels when it can "
                                "(not a reference trait; False = the reference operator sequence)")

    def _exec(self, data, detectors=None, **kwargs):
        for trait in ("binning",):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        import time as _time

        from ..accel import native as _native

        def _lap(label, t_start):
            _native().accel_synchronize()
            self.timing_log[label] = self.timing_log.get(label, 0.0) + (_time.time() - t_start)
            return _time.time()

        binning = self.binning
        map_binning = self.map_binning if self.map_binning is not None else binning
        self.history = []
        self.timing_log = {}
        _t = _time.time()
        if self.reset_pix_dist and binning.pixel_dist in data:
            del data[binning.pixel_dist]
        hits_name, cov_name, rcond_name = f"{self.name}_hits", f"{self.name}_cov", f"{self.name}_rcond"
        map_name, amp_name = f"{self.name}_map", f"{self.name}_amplitudes"
        # Solver flags: one uint8 per sample combining detector and shared flags, so that the
        # binning and the templates cut exactly the same samples while solving
        # (SolveAmplitudes._prepare_flagging, mapmaker_templates.py:764-810).
        solver_flags = f"{self.name}_solve_flags"
        saved = (binning.det_flags, binning.det_flag_mask)
        tm = self.template_matrix
        use_templates = tm is not None and len(tm.templates) > 0
        if use_templates:
            from ..accel import accel_enabled as _acc_on

            for ob in data.obs:
                if _acc_on():
                    self._solver_flags_device(ob, solver_flags, binning)
                    continue
                ob.detdata.ensure(solver_flags, dtype=np.uint8, detectors=ob.local_detectors)
                sf = ob.detdata[solver_flags]
                sf.data[:] = 0
                if binning.det_flags is not None:
                    src = ob.detdata[binning.det_flags]
                    for d in ob.local_detectors:
                        sf[d][(src[d] & binning.det_flag_mask) != 0] = 1
                if binning.shared_flags is not None:
                    shared = ob.shared[binning.shared_flags]
                    if shared.accel_in_use():
                        shared.accel_update_host()
                    bad = (shared.data & binning.shared_flag_mask) != 0
                    sf.data[:, bad] = 1
                if binning.pixel_pointing.view is not None:
                    outside = np.ones(ob.n_local_samples, dtype=bool)
                    for iv in ob.intervals[binning.pixel_pointing.view]:
                        outside[iv.first:iv.last] = False
                    sf.data[:, outside] = 1
            binning.det_flags, binning.det_flag_mask = solver_flags, 1
            tm.det_flags, tm.det_flag_mask = solver_flags, 1
        # Cached pointing is written once and read by every later phase: keep it on the device
        # (288 GB HBM) instead of the copy-back / delete / re-upload of each Pipeline.
        pinned = None
        from ..accel import accel_enabled as _accel_enabled

        if binning.full_pointing and _accel_enabled():
            pinned = {"detdata": [binning.pixel_pointing.pixels, binning.stokes_weights.weights]}
            data.accel_pin(pinned)
        # covariance + hits with the solver flags
        cov_op = CovarianceAndHits(
            pixel_dist=binning.pixel_dist, covariance=cov_name, hits=hits_name, rcond=rcond_name,
            det_mask=binning.det_mask, det_flags=binning.det_flags, det_flag_mask=binning.det_flag_mask,
            shared_flags=binning.shared_flags, shared_flag_mask=binning.shared_flag_mask,
            pixel_pointing=binning.pixel_pointing, stokes_weights=binning.stokes_weights,
            noise_model=binning.noise_model, rcond_threshold=self.solve_rcond_threshold,
            sync_type=binning.sync_type, save_pointing=binning.full_pointing)
        cov_op.apply(data, detectors=detectors)
        _t = _lap("covariance_and_hits", _t)
        binning.covariance = cov_name
        map_binning.covariance = cov_name
        if use_templates:
            # samples in pixels that fail the rcond cut must not constrain the templates
            # (SolveAmplitudes._get_rcond_mask, mapmaker_templates.py:895-939)
            mask_name = f"{self.name}_rcond_mask"
            data[mask_name] = PixelData(data[binning.pixel_dist], np.uint8, n_value=1)
            data[mask_name].data[data[rcond_name].data < self.solve_rcond_threshold] = 1
            scanner = ScanMask(det_flags=solver_flags, det_flags_value=1, det_mask=binning.det_mask,
                               pixels=binning.pixel_pointing.pixels, view=binning.pixel_pointing.view,
                               mask_key=mask_name)
            scan_pipe = Pipeline(detector_sets=["ALL"] if binning.full_pointing else uncached_detector_sets(),
                                 operators=[binning.pixel_pointing, scanner])
            scan_pipe.apply(data, detectors=detectors)
            del data[mask_name]
        cleaned = self.det_data
        if use_templates:
            tm.reset()
            tm.amplitudes = f"{self.name}_rhs"
            solver_bin = f"{self.name}_solve_bin"
            binning.binned = solver_bin
            rhs = SolverRHS(name=f"{self.name}_rhs", det_data=self.det_data, binning=binning, template_matrix=tm)
            if f"{self.name}_rhs" in data:
                del data[f"{self.name}_rhs"]
            rhs.apply(data, detectors=detectors)
            _t = _lap("rhs", _t)
            lhs = SolverLHS(name=f"{self.name}_lhs", binning=binning, template_matrix=tm, fused=self.fused_lhs)
            if amp_name in data:
                del data[amp_name]
            self.iteration_seconds = []
            self.history = solve(data, detectors, lhs, f"{self.name}_rhs", amp_name, convergence=self.convergence,
                                 n_iter_min=self.iter_min, n_iter_max=self.iter_max,
                                 iteration_seconds=self.iteration_seconds)
            _t = _lap("pcg_iterations", _t)
            for ob in data.obs:
                if lhs.det_temp in ob.detdata:
                    del ob.detdata[lhs.det_temp]
            if not self.keep_solver_products:
                for key in (solver_bin, f"{self.name}_rhs"):
                    if key in data:
                        if hasattr(data[key], "clear"):
                            data[key].clear()
                        del data[key]
            # cleaned timestreams = d - M a
            if self.save_cleaned and not self.overwrite_cleaned:
                cleaned = f"{self.name}_cleaned"
            elif not self.overwrite_cleaned:
                cleaned = f"{self.name}_temp_cleaned"
            ApplyAmplitudes(op="subtract", amplitudes=amp_name, template_matrix=tm, det_data=self.det_data,
                            output=None if cleaned == self.det_data else cleaned).apply(data, detectors=detectors)
        if use_templates:
            binning.det_flags, binning.det_flag_mask = saved
            Delete(detdata=[solver_flags]).apply(data)
        map_binning.binned = map_name
        map_binning.det_data = cleaned
        _t = _lap("apply_amplitudes", _t)
        map_binning.apply(data, detectors=detectors)
        _t = _lap("final_binning", _t)
        if cleaned.endswith("_temp_cleaned"):
            Delete(detdata=[cleaned]).apply(data)
        if pinned is not None:
            data.accel_unpin(pinned)
            _t = _lap("unpin_pointing", _t)

    @staticmethod
    def _solver_flags_device(ob, solver_flags, binning):
        """The same combination on the device (toast_hip_combine_flags_dev): nothing but the
        uint8 inputs cross PCIe, and those only if they are not resident yet."""
        from .. import capi
        from ..accel import accel_device_ptr

        dets = ob.local_detectors
        ob.detdata.ensure(solver_flags, dtype=np.uint8, detectors=dets, accel=True)
        sf = ob.detdata[solver_flags]
        n_samp = ob.n_local_samples
        f_ptr, f_n, f_idx = 0, 0, np.zeros(len(dets), np.int32)
        if binning.det_flags is not None:
            src = ob.detdata[binning.det_flags]
            if not src.accel_in_use():
                if not src.accel_exists():
                    src.accel_create(binning.det_flags)
                src.accel_update_device()
            f_ptr, f_n, f_idx = accel_device_ptr(src.buffer), n_samp, src.indices(dets)
        s_ptr, s_n = 0, 0
        if binning.shared_flags is not None:
            shared = ob.shared[binning.shared_flags]
            if not shared.accel_in_use():
                if not shared.accel_exists():
                    shared.accel_create(binning.shared_flags)
                shared.accel_update_device()
            s_ptr, s_n = accel_device_ptr(shared.data), n_samp
        view = binning.pixel_pointing.view
        capi.dev.combine_flags(accel_device_ptr(sf.buffer), sf.indices(dets), f_ptr, f_n, f_idx,
                               binning.det_flag_mask, s_ptr, s_n, binning.shared_flag_mask, n_samp,
                               ob.intervals[view].data, n_out_rows=len(sf.detectors),
                               outside_value=1 if view is not None else 0)
        sf.accel_used(True)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.binning.requires()
        req["detdata"].append(self.det_data)
        return req

    def _provides(self):
        return {"global": [f"{self.name}_map", f"{self.name}_hits", f"{self.name}_cov", f"{self.name}_rcond"]}
