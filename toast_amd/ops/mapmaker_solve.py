"""Template regression: TemplateMatrix, SolverRHS, SolverLHS and the PCG ``solve`` loop.

Reference: src/toast/ops/mapmaker_templates.py:32-420 (TemplateMatrix),
src/toast/ops/mapmaker_solve.py:27-229 (SolverRHS), :232-521 (SolverLHS), :524-755 (solve).

    RHS:  b  = M^T N^-1 Z d          LHS:  a' = M^T N^-1 Z M a (+ prior)
    Z = I - P (P^T N^-1 P)^-1 P^T N^-1
"""

import numpy as np

from ..data import defaults
from ..templates import AmplitudesMap
from ..traits import Bool, Float, ImplementationType, Instance, Int, List, Unicode
from .mapmaker_ops import BinMap, Copy, Delete, NoiseWeight, ScanMap
from .operator import Operator
from .pipeline import Pipeline


class TemplateMatrix(Operator):
    """Projects amplitudes to timestreams (``transpose=False``: tod = M a, after zeroing) or
    accumulates timestreams into amplitudes (``transpose=True``: a += M^T tod)."""

    API = Int(0, help="Internal interface version for this operator")
    templates = List([], help="This should be a list of Template-derived objects")
    amplitudes = Unicode(None, allow_none=True, help="Data key for template amplitudes")
    transpose = Bool(False, help="If True, apply the transpose.")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    det_data = Unicode(None, allow_none=True, help="Observation detdata key for the timestream data")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired units of detector data")
    det_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for per-detector flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for solver flags to use")
    det_flag_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for solver flags")

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self._initialized = False

    def duplicate(self):
        """Same templates, independent traits (mapmaker_templates.py:181-196)."""
        ret = TemplateMatrix(API=self.API, templates=list(self.templates), amplitudes=self.amplitudes,
                             transpose=self.transpose, view=self.view, det_data=self.det_data,
                             det_data_units=self.det_data_units, det_mask=self.det_mask, det_flags=self.det_flags,
                             det_flag_mask=self.det_flag_mask)
        ret._initialized = self._initialized
        return ret

    def reset(self):
        self._initialized = False
        for tmpl in self.templates:
            if hasattr(tmpl, "clear"):
                tmpl.clear()

    def reset_templates(self):
        self.reset()

    def initialize(self, data):
        if not self._initialized:
            for tmpl in self.templates:
                tmpl.view = self.view
                tmpl.det_data_units = self.det_data_units
                tmpl.det_mask = self.det_mask
                tmpl.det_flags = self.det_flags
                tmpl.det_flag_mask = self.det_flag_mask
                # the data trait triggers template initialisation
                tmpl.det_data = self.det_data
                tmpl.data = data
            self._initialized = True

    def apply_precond(self, amps_in, amps_out, use_accel=None, **kwargs):
        if not self._initialized:
            raise RuntimeError("You must call exec() once before applying preconditioners")
        for tmpl in self.templates:
            if tmpl.enabled:
                tmpl.apply_precond(amps_in[tmpl.name], amps_out[tmpl.name], use_accel=use_accel, **kwargs)

    def add_prior(self, amps_in, amps_out, use_accel=None, **kwargs):
        if not self._initialized:
            raise RuntimeError("You must call exec() once before adding the prior")
        for tmpl in self.templates:
            if tmpl.enabled:
                tmpl.add_prior(amps_in[tmpl.name], amps_out[tmpl.name], use_accel=use_accel, **kwargs)

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        if self.det_data is None:
            raise RuntimeError("You must set the det_data trait before calling exec()")
        if self.amplitudes is None:
            raise RuntimeError("You must set the amplitudes trait before calling exec()")
        if len(self.templates) == 0:
            raise RuntimeError("No templates in use")
        for tmpl in self.templates:
            tmpl.det_data = self.det_data
        self.initialize(data)
        all_dets = data.all_local_detectors(selection=detectors, flagmask=self.det_mask)
        if self.transpose:
            if self.amplitudes not in data:
                data[self.amplitudes] = AmplitudesMap()
                for tmpl in self.templates:
                    if tmpl.enabled:
                        data[self.amplitudes][tmpl.name] = tmpl.zeros()
            for d in all_dets:
                for tmpl in self.templates:
                    if tmpl.enabled:
                        tmpl.project_signal(d, data[self.amplitudes][tmpl.name], use_accel=use_accel, **kwargs)
        else:
            if self.amplitudes not in data:
                raise RuntimeError(f"Template amplitudes '{self.amplitudes}' do not exist in data")
            for ob in data.obs:
                dets = ob.select_local_detectors(selection=detectors, flagmask=self.det_mask)
                exists = ob.detdata.ensure(self.det_data, detectors=dets, accel=use_accel,
                                           create_units=self.det_data_units)
                if exists:
                    ob.detdata[self.det_data].reset(dets=dets)
            for d in all_dets:
                for tmpl in self.templates:
                    if tmpl.enabled:
                        tmpl.add_to_signal(d, data[self.amplitudes][tmpl.name], use_accel=use_accel, **kwargs)

    def _finalize(self, data, use_accel=None, **kwargs):
        if self.transpose and self.amplitudes in data:
            # make the accumulated amplitudes current on the host for the PCG algebra
            data[self.amplitudes].accel_update_host()
            for v in data[self.amplitudes].values():
                v.sync()

    def _requires(self):
        req = {"global": [], "meta": [], "shared": [], "detdata": [], "intervals": []}
        if self.view is not None:
            req["intervals"].append(self.view)
        if self.transpose:
            req["detdata"].append(self.det_data)
            if self.det_flags is not None:
                req["detdata"].append(self.det_flags)
        else:
            req["global"].append(self.amplitudes)
        return req

    def _provides(self):
        prov = {"global": [], "detdata": []}
        if self.transpose:
            prov["global"].append(self.amplitudes)
        else:
            prov["detdata"].append(self.det_data)
        return prov

    def _implementations(self):
        return [ImplementationType.DEFAULT, ImplementationType.COMPILED]

    def _supports_accel(self):
        return all(t.supports_accel() for t in self.templates)


class SolverRHS(Operator):
    """Right-hand side ``b = M^T N^-1 Z d`` (mapmaker_solve.py:27-229)."""

    API = Int(0, help="Internal interface version for this operator")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired units of detector data")
    binning = Instance(klass=Operator, help="Binning operator for solving")
    template_matrix = Instance(klass=Operator, help="This must be an instance of a template matrix operator")

    def _exec(self, data, detectors=None, **kwargs):
        for trait in ("det_data", "binning", "template_matrix"):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        self.binning.det_data = self.det_data
        self.binning.det_data_units = self.det_data_units
        self.binning.apply(data, detectors=detectors)
        det_temp = "temp_RHS"
        pixels = self.binning.pixel_pointing
        weights = self.binning.stokes_weights
        copy_det = Copy(detdata=[(self.det_data, det_temp)])
        scan_map = ScanMap(pixels=pixels.pixels, weights=weights.weights, view=pixels.view,
                           map_key=self.binning.binned, det_data=det_temp, det_data_units=self.det_data_units,
                           det_mask=self.binning.det_mask, det_flag_mask=self.binning.det_flag_mask, subtract=True)
        noise_weight = NoiseWeight(noise_model=self.binning.noise_model, det_data=det_temp,
                                   det_mask=self.binning.det_mask, det_flag_mask=self.binning.det_flag_mask,
                                   view=pixels.view)
        self.template_matrix.transpose = True
        self.template_matrix.det_data = det_temp
        self.template_matrix.det_data_units = self.det_data_units
        if self.binning.full_pointing:
            ops = [copy_det, scan_map, noise_weight, self.template_matrix]
            proj_pipe = Pipeline(detector_sets=["ALL"], operators=ops)
        else:
            ops = [copy_det, pixels, weights, scan_map, noise_weight, self.template_matrix]
            proj_pipe = Pipeline(detector_sets=["SINGLE"], operators=ops)
        proj_pipe.apply(data, detectors=detectors)
        Delete(detdata=[det_temp]).apply(data)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.binning.requires()
        req["detdata"].append(self.det_data)
        return req

    def _provides(self):
        return {"global": [self.template_matrix.amplitudes]}


class SolverLHS(Operator):
    """Left-hand side ``a' = M^T N^-1 Z M a + M_p a`` for the current proposal
    (mapmaker_solve.py:232-521).  This is the per-iteration hot loop."""

    API = Int(0, help="Internal interface version for this operator")
    det_temp = Unicode("temp_LHS", help="Observation detdata key for temporary timestream data")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired units of detector data")
    binning = Instance(klass=Operator, help="Binning operator for solving")
    template_matrix = Instance(klass=Operator, help="This must be an instance of a template matrix operator")
    out = Unicode(None, allow_none=True, help="Output Data key for resulting amplitudes")

    def _zero_temp(self, data):
        for ob in data.obs:
            if self.det_temp in ob.detdata:
                ob.detdata[self.det_temp].reset()

    def _exec(self, data, detectors=None, **kwargs):
        for trait in ("binning", "template_matrix", "out"):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        self._zero_temp(data)
        pixels = self.binning.pixel_pointing
        weights = self.binning.stokes_weights
        self.template_matrix.transpose = False
        self.template_matrix.det_data = self.det_temp
        self.template_matrix.det_data_units = self.det_data_units
        self.binning.det_data = self.det_temp
        self.binning.det_data_units = self.det_data_units
        self.binning.pre_process = self.template_matrix
        self.binning.apply(data, detectors=detectors)
        self.binning.pre_process = None
        if self.out in data:
            data[self.out].reset()
        self.template_matrix.add_prior(data[self.template_matrix.amplitudes], data[self.out])
        scan_map = ScanMap(pixels=pixels.pixels, weights=weights.weights, view=pixels.view,
                           map_key=self.binning.binned, det_data=self.det_temp, det_data_units=self.det_data_units,
                           det_mask=self.binning.det_mask, det_flag_mask=self.binning.det_flag_mask, subtract=True)
        noise_weight = NoiseWeight(noise_model=self.binning.noise_model, det_data=self.det_temp,
                                   det_mask=self.binning.det_mask, det_flag_mask=self.binning.det_flag_mask,
                                   view=pixels.view)
        template_transpose = self.template_matrix.duplicate()
        template_transpose.amplitudes = self.out
        template_transpose.transpose = True
        self._zero_temp(data)
        if self.binning.full_pointing:
            ops = [self.template_matrix, scan_map, noise_weight, template_transpose]
            proj_pipe = Pipeline(detector_sets=["ALL"], operators=ops)
        else:
            ops = [self.template_matrix, pixels, weights, scan_map, noise_weight, template_transpose]
            proj_pipe = Pipeline(detector_sets=["SINGLE"], operators=ops)
        proj_pipe.apply(data, detectors=detectors)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.binning.requires()
        req["global"].append(self.template_matrix.amplitudes)
        return req

    def _provides(self):
        return {"global": [self.out]}


def solve(data, detectors, lhs_op, rhs_key, result_key, convergence=1.0e-12, n_iter_min=3, n_iter_max=100,
          log=None):
    """Preconditioned conjugate gradient for the template amplitudes
    (mapmaker_solve.py:524-755; same recurrence, same convergence / stall tests).

    Returns the list of relative residuals, one per iteration."""
    if rhs_key not in data:
        raise RuntimeError(f"rhs_key '{rhs_key}' does not exist in data")
    rhs = data[rhs_key]
    if not isinstance(rhs, AmplitudesMap):
        raise RuntimeError("rhs_key does not point to an AmplitudesMap")
    if result_key not in data:
        data[result_key] = rhs.duplicate()
        data[result_key].reset()
    result = data[result_key]
    if not isinstance(result, AmplitudesMap):
        raise RuntimeError("result_key does not point to an AmplitudesMap")

    lhs_out_key = f"{lhs_op.name}_out"
    if lhs_out_key in data:
        data[lhs_out_key].clear()
        del data[lhs_out_key]
    data[lhs_out_key] = rhs.duplicate()
    lhs_out = data[lhs_out_key]
    proposal_key = f"{lhs_op.name}_in"
    if proposal_key in data:
        data[proposal_key].clear()
        del data[proposal_key]
    data[proposal_key] = rhs.duplicate()
    data[proposal_key].reset()
    proposal = data[proposal_key]
    temp = rhs.duplicate()
    temp.reset()

    # residual of the starting guess
    lhs_op.template_matrix.amplitudes = result_key
    lhs_op.out = lhs_out_key
    lhs_op.apply(data, detectors=detectors)
    residual = rhs.duplicate()
    residual -= lhs_out
    precond = rhs.duplicate()
    precond.reset()
    lhs_op.template_matrix.apply_precond(residual, precond)
    for k, v in proposal.items():
        v._host()
        v.local[:] = precond[k].local
    lhs_op.template_matrix.amplitudes = proposal_key

    sqsum = rhs.dot(rhs)
    sqsum_init = sqsum
    sqsum_best = sqsum
    last_best = sqsum
    delta = proposal.dot(residual)
    history = []
    for it in range(n_iter_max):
        if not np.isfinite(sqsum):
            raise RuntimeError("Residual is not finite")
        lhs_op.apply(data, detectors=detectors)
        alpha = delta / proposal.dot(lhs_out)
        temp.reset()
        for k, v in temp.items():
            v.local[:] = proposal[k].local
        temp *= alpha
        result += temp
        temp.reset()
        for k, v in temp.items():
            v.local[:] = lhs_out[k].local
        temp *= alpha
        residual -= temp
        sqsum = residual.dot(residual)
        relative = sqsum / sqsum_init if sqsum_init != 0 else 0.0
        history.append(relative)
        if log is not None:
            log(f"MapMaker iteration {it:4d}, relative residual = {relative:0.6e}")
        if relative < convergence or sqsum < 1e-30:
            break
        sqsum_best = min(sqsum, sqsum_best)
        if it % 10 == 0 and it >= n_iter_min:
            if last_best < sqsum_best * 2:
                break
            last_best = sqsum_best
        lhs_op.template_matrix.apply_precond(residual, precond)
        delta_last = delta
        delta = precond.dot(residual)
        beta = delta / delta_last
        proposal *= beta
        proposal += precond
    temp.clear()
    proposal.clear()
    del data[proposal_key]
    lhs_out.clear()
    del data[lhs_out_key]
    return history
