"""MapMaker: solve for template amplitudes (destriping) and bin the cleaned signal.

Reference: src/toast/ops/mapmaker.py:37-790 (MapMaker) and
src/toast/ops/mapmaker_templates.py:423-1125 (SolveAmplitudes, ApplyAmplitudes).  The
orchestration is reproduced for the configuration used on the hot path: covariance + hits
from the same pointing, RHS, PCG on the Offset amplitudes, template subtraction, final BinMap.
"""

import os as _os
import sys as _sys

import numpy as np

from ..data import defaults
from ..pixels import PixelData
from ..traits import Bool, Float, Instance, Int, Unicode
from .arithmetic import Combine
from .mapmaker_ops import Copy, CovarianceAndHits, Delete, ScanMask
from .mapmaker_solve import SolverLHS, SolverRHS, solve
from .operator import Operator
from .pipeline import Pipeline, uncached_detector_sets


def _pin_for_mapmaking(data, binning):
    """Keep what every Pipeline of a map-making run reads on the device between them: the cached pointing (written once,
    read by every later phase) and the shared inputs of the pointing operators (boresight, shared flags, HWP angle: a few
    tens of MB that each Pipeline would otherwise upload again and free).  Only names that nobody holds yet are taken, so
    that nested calls (MapMaker -> SolveAmplitudes) release what they took and nothing else.  Returns the dict to hand to
    ``_unpin_for_mapmaking``."""
    from ..accel import accel_enabled

    if not accel_enabled():
        return None
    pixels, weights = binning.pixel_pointing, binning.stokes_weights
    want = {"detdata": [], "shared": []}
    if binning.full_pointing:
        want["detdata"] = [pixels.pixels, weights.weights]
    for op, trait in ((pixels.detector_pointing, "boresight"), (pixels.detector_pointing, "shared_flags"),
                      (weights, "hwp_angle"), (binning, "shared_flags")):
        key = getattr(op, trait, None)
        if key is not None and key not in want["shared"]:
            want["shared"].append(key)
    taken = {k: [n for n in v if n not in data._pinned[k]] for k, v in want.items()}
    data.accel_pin(taken)
    return taken


def _unpin_for_mapmaking(data, taken):
    if taken is None:
        return
    data.accel_unpin({"detdata": taken["detdata"]})
    data.accel_unpin({"shared": taken["shared"]}, update_host=False)   # inputs: never modified on the device


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
        result = self.det_data if self.output is None else self.output
        n_enabled = len([t for t in tm.templates if t.enabled])
        if self.op in ("subtract", "add") and n_enabled == 1 and self.amplitudes in data:
            # One template: det -/+ M a = det + M (-/+ a) exactly (the projection is linear and a sign change is exact),
            # so the template is added straight into the result with signed amplitudes -- no second buffer the size of
            # the timestreams (5.9 GB at cfg-3: an allocation the driver may have to clear first, 120 ms) and no
            # separate combine pass.  Several templates keep the reference's order of operations below.
            signed = f"{self.name}_signed_amplitudes"
            data[signed] = data[self.amplitudes].duplicate()
            try:
                if self.op == "subtract":
                    data[signed] *= -1.0
                tm.amplitudes, tm.det_data, tm.accumulate = signed, result, True
                Pipeline(operators=[tm]).apply(data, detectors=detectors)
            finally:
                data[signed].clear()
                del data[signed]
            return
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


class SolveAmplitudes(Operator):
    """Solve for template amplitudes (reference: src/toast/ops/mapmaker_templates.py:407-1155):

        a = (M^T N^-1 Z M + M_p)^-1 M^T N^-1 Z d,     Z = I - P (P^T N^-1 P)^-1 P^T N^-1

    Stages, as in the reference: solver flags (bit 1: detector | shared flags, bit 2: pixel
    mask, bit 4: poorly conditioned pixels), solver covariance / hits / rcond, right-hand side,
    PCG.  Products ``<name>_solve_{hits,cov,rcond,rcond_mask,rhs,bin,flags}`` are removed at the
    end unless ``keep_solver_products``; the solution is ``data[self.amplitudes]``
    (default ``<name>_solve_amplitudes``)."""

    API = Int(0, help="Internal interface version for this operator")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    amplitudes = Unicode(None, allow_none=True, help="Data key for output amplitudes")
    convergence = Float(1.0e-12, help="Relative convergence limit")
    iter_min = Int(3, help="Minimum number of iterations")
    iter_max = Int(100, help="Maximum number of iterations")
    solve_rcond_threshold = Float(1.0e-8, help="When solving, minimum value for inverse pixel condition number cut.")
    map_rcond_threshold = Float(1.0e-8, help="For final map, minimum value for inverse pixel condition number cut "
                                "(declared like the reference's; SolveAmplitudes itself does not bin a final map).")
    mask = Unicode(None, allow_none=True, help="Data key for pixel mask to use in solving.  "
                                               "First bit of pixel values is tested")
    binning = Instance(klass=Operator, help="Binning operator used for solving template amplitudes")
    template_matrix = Instance(klass=Operator, help="This must be an instance of a template matrix operator")
    keep_solver_products = Bool(False, help="If True, keep the map domain solver products in data")
    mc_mode = Bool(False, help="If True, re-use solver flags, sparse covariances, etc")
    mc_index = Int(None, allow_none=True, help="The Monte-Carlo index")
    reset_pix_dist = Bool(False, help="Clear any existing pixel distribution.")
    fused_lhs = Bool(True, help="Let SolverLHS use the fused device-resident kernels when it can "
                                "(not a reference trait; False = the reference operator sequence)")

    def _names(self):
        n = self.name
        # Monte-Carlo realisations share flags / covariance and get their own rhs / bin /
        # amplitudes (mapmaker_templates.py:612-626)
        root = f"{n}_{self.mc_index:05d}" if (self.mc_mode and self.mc_index is not None) else n
        return dict(flags=f"{n}_solve_flags", hits=f"{n}_solve_hits", cov=f"{n}_solve_cov", rcond=f"{n}_solve_rcond",
                    rcond_mask=f"{n}_solve_rcond_mask", rhs=f"{root}_solve_rhs", bin=f"{root}_solve_bin",
                    amplitudes=f"{root}_solve_amplitudes")

    def _exec(self, data, detectors=None, **kwargs):
        import time as _time

        from ..accel import accel_enabled, native

        for trait in ("binning", "template_matrix"):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        binning, tm = self.binning, self.template_matrix
        if len([t for t in tm.templates if t.enabled]) == 0:
            raise RuntimeError("No enabled templates: nothing to solve for")
        self.history, self.iteration_seconds = [], []
        self.timing_log = {}

        def lap(label, t0):
            # (a phase boundary is a host-side mark: the device is waited for only when somebody wants to READ the phase
            # times -- TOAST_HIP_TRACE / TOAST_HIP_PHASE_SYNC=1 --, otherwise the host goes on into the next phase's own
            # work, e.g. the templates' bookkeeping, while the device is still busy with this one)
            if accel_enabled() and (_os.environ.get("TOAST_HIP_PHASE_SYNC", "0") == "1"
                                    or _os.environ.get("TOAST_HIP_TRACE", "0") not in ("", "0")):
                native().accel_synchronize()
            self.timing_log[label] = self.timing_log.get(label, 0.0) + (_time.time() - t0)
            if _os.environ.get("TOAST_HIP_TRACE", "0") not in ("", "0"):
                print(f"[toast_hip] phase         {label:30s} {1e3 * (_time.time() - t0):10.2f} ms", file=_sys.stderr, flush=True)
            return _time.time()

        t0 = _time.time()
        nm = self._names()
        if self.amplitudes is None:
            self.amplitudes = nm["amplitudes"]
        # state of the binning / template operators that the solve borrows (:596-611)
        saved = dict(det_flags=binning.det_flags, det_flag_mask=binning.det_flag_mask, binned=binning.binned,
                     covariance=binning.covariance, shared_flags=binning.shared_flags,
                     shared_flag_mask=binning.shared_flag_mask, tm_flags=tm.det_flags, tm_flag_mask=tm.det_flag_mask,
                     tm_det_data=tm.det_data, tm_amplitudes=tm.amplitudes)
        pixels, weights = binning.pixel_pointing, binning.stokes_weights
        pixels.detector_pointing.det_mask = binning.det_mask
        if self.reset_pix_dist:
            for key in (nm["hits"], nm["cov"], nm["rcond_mask"], nm["rcond"], nm["rhs"], nm["bin"], binning.pixel_dist):
                if key in data:
                    del data[key]
            Delete(detdata=[pixels.pixels, weights.weights, pixels.detector_pointing.quats, nm["flags"]]).apply(data)
        # Cached pointing is written once and read by every later phase: keep it on the device
        pinned = _pin_for_mapmaking(data, binning)
        rhs_map_ready = False
        self.rhs_map_with_covariance = None
        if self.mc_mode:
            # re-use the flags and the covariance of an earlier realisation (:702-712, :852-864)
            for ob in data.obs:
                dets = ob.select_local_detectors(detectors, flagmask=binning.det_mask)
                if nm["flags"] not in ob.detdata or not set(dets) <= set(ob.detdata[nm["flags"]].detectors):
                    raise RuntimeError(f"In MC mode, solver flags missing for observation {ob.name}")
            if binning.pixel_dist not in data:
                raise RuntimeError(f"MC mode, pixel distribution '{binning.pixel_dist}' does not exist")
            if nm["cov"] not in data:
                raise RuntimeError(f"MC mode, covariance '{nm['cov']}' does not exist")
        else:
            # -- solver flags (:698-810)
            for ob in data.obs:
                if accel_enabled() and getattr(data, "lazy_host", False):
                    # built on the device and left there; with eager copies (lazy_host False: Pipelines free the
                    # device copies of their inputs without copying them back) the flags are built on the host
                    MapMaker._solver_flags_device(ob, nm["flags"], binning, detectors)
                else:
                    self._solver_flags_host(ob, nm["flags"], binning, detectors)
            scanner = ScanMask(det_flags=nm["flags"], det_mask=binning.det_mask, pixels=pixels.pixels, view=pixels.view)
            scan_pipe = Pipeline(detector_sets=["ALL"] if binning.full_pointing else uncached_detector_sets(),
                                 operators=[pixels, scanner])
            if self.mask is not None:
                scanner.det_flags_value, scanner.mask_key = 2, self.mask
                scan_pipe.apply(data, detectors=detectors)
            # -- solver covariance, hits, condition numbers (:846-900)
            # The right-hand side's first step, A^T N^-1 d (:941-1000), reads the same pixels, weights and flags: with
            # cached pointing on the accelerator it rides along with the covariance (one sweep of 42 B per
            # detector-sample instead of 33 + 41; TOAST_HIP_FUSED_COV_RHS=0 keeps the two apart).  The right-hand side
            # would see the solver flags WITH the condition-number cut, which is only known after this pass: the two
            # differ in samples of pixels whose covariance the cut zeroes -- the binned map is zero there either way.
            rhs_rides = self._rhs_map_rides_along(data, binning, detectors)
            if rhs_rides and nm["bin"] in data:
                del data[nm["bin"]]
            cah = CovarianceAndHits(
                pixel_dist=binning.pixel_dist, covariance=nm["cov"], hits=nm["hits"], rcond=nm["rcond"],
                det_mask=binning.det_mask, det_flags=nm["flags"], det_flag_mask=255, shared_flags=None,
                pixel_pointing=pixels, stokes_weights=weights, noise_model=binning.noise_model,
                rcond_threshold=self.solve_rcond_threshold, sync_type=binning.sync_type,
                # (MapMaker asks for the inverse covariance too when the final binning's products are going to be these
                #  very arrays: _share_with_final below)
                inverse_covariance=getattr(self, "_share_invcov", None),
                signal=self.det_data if rhs_rides else None, signal_map=nm["bin"] if rhs_rides else None,
                save_pointing=binning.full_pointing, det_data_units=binning.det_data_units)
            cah.apply(data, detectors=detectors)
            rhs_map_ready = bool(rhs_rides and cah.signal_in_one_sweep is not None and nm["bin"] in data)
            # how the right-hand side's map was accumulated: None = by the right-hand side itself; a tuple = with the
            # covariance, per observation True when one kernel did it, False when the library ran the separate sweeps
            self.rhs_map_with_covariance = cah.signal_in_one_sweep if rhs_map_ready else None
            t0 = lap("covariance_and_hits", t0)
            # -- samples in poorly conditioned pixels must not constrain the templates (:902-939)
            data[nm["rcond_mask"]] = PixelData(data[binning.pixel_dist], np.uint8, n_value=1)
            rc, mask = data[nm["rcond"]], data[nm["rcond_mask"]]
            if rc.accel_in_use():
                # the condition numbers were computed on the device: threshold them there
                from .. import capi
                from ..accel import accel_device_ptr

                mask.accel_create(nm["rcond_mask"], zero_out=True)
                mask.accel_used(True)
                capi.dev.threshold_mask(rc.buffer.size, accel_device_ptr(rc.buffer), self.solve_rcond_threshold, 1,
                                        accel_device_ptr(mask.buffer))
            else:
                mask.data[rc.data < self.solve_rcond_threshold] = 1
            scanner.det_flags_value, scanner.mask_key = 4, nm["rcond_mask"]
            scan_pipe.apply(data, detectors=detectors)
        # -- right-hand side (:941-1000): the binning and the templates see the solver flags only
        binning.det_flags, binning.det_flag_mask = nm["flags"], 255
        binning.shared_flags = None
        tm.det_flags, tm.det_flag_mask = nm["flags"], 255
        binning.covariance, binning.binned = nm["cov"], nm["bin"]
        try:
            tm.reset()
            tm.amplitudes = nm["rhs"]
            if nm["rhs"] in data:
                del data[nm["rhs"]]
            if not self.mc_mode and rhs_map_ready:
                binning._zmap_is_accumulated = True      # (BinMap consumes the mark in its next call)
            SolverRHS(name=f"{self.name}_rhs", det_data=self.det_data, binning=binning,
                      template_matrix=tm, fused=self.fused_lhs).apply(data, detectors=detectors)
            binning.__dict__.pop("_zmap_is_accumulated", None)
            t0 = lap("rhs", t0)
            # -- PCG (:1002-1060)
            lhs = SolverLHS(name=f"{self.name}_lhs", binning=binning, template_matrix=tm, fused=self.fused_lhs)
            if self.amplitudes in data:
                del data[self.amplitudes]
            self.history = solve(data, detectors, lhs, nm["rhs"], self.amplitudes, convergence=self.convergence,
                                 n_iter_min=self.iter_min, n_iter_max=self.iter_max,
                                 iteration_seconds=self.iteration_seconds)
            # how the left-hand side was applied, per observation: "packed" | "fused" | "fused-otf", or ("sequence",)
            self.lhs_route = tuple(getattr(lhs, "last_route", ()))
            self.lhs_pack_bytes = tuple(getattr(lhs, "last_pack_bytes", ()))
            t0 = lap("pcg_iterations", t0)
            for ob in data.obs:
                if lhs.det_temp in ob.detdata:
                    del ob.detdata[lhs.det_temp]
        finally:
            # -- restore the borrowed operators (:1062-1100)
            binning.det_flags, binning.det_flag_mask = saved["det_flags"], saved["det_flag_mask"]
            binning.shared_flags, binning.shared_flag_mask = saved["shared_flags"], saved["shared_flag_mask"]
            binning.binned, binning.covariance = saved["binned"], saved["covariance"]
            tm.det_flags, tm.det_flag_mask = saved["tm_flags"], saved["tm_flag_mask"]
        if not self.keep_solver_products and not self.mc_mode:
            for key in (nm["hits"], nm["cov"], nm["rcond"], nm["rcond_mask"], nm["rhs"], nm["bin"]):
                if key in getattr(self, "_share_keep", ()):
                    continue          # MapMaker takes it over as a final product
                if key in data:
                    if hasattr(data[key], "clear") and not isinstance(data[key], PixelData):
                        data[key].clear()
                    del data[key]
            Delete(detdata=[nm["flags"]]).apply(data)
        _unpin_for_mapmaking(data, pinned)

    def _rhs_map_rides_along(self, data, binning, detectors):
        """May the covariance pass accumulate the right-hand side's noise-weighted map as well?  Cached pointing on the
        accelerator, float64 timestreams of the same detectors, nothing between the timestream and the binning, nobody
        who wants the noise-weighted map itself (it differs in the pixels the condition-number cut removes)."""
        from ..accel import accel_enabled

        if _os.environ.get("TOAST_HIP_FUSED_COV_RHS", "1") == "0" or self.mc_mode:
            return False
        if not (accel_enabled() and getattr(data, "lazy_host", False) and binning.full_pointing):
            return False
        if binning.pre_process is not None or binning.noiseweighted is not None or self.det_data is None:
            return False
        if not (binning.stokes_weights.supports_accel() and binning.pixel_pointing.supports_accel()):
            return False
        for ob in data.obs:
            dets = ob.select_local_detectors(detectors, flagmask=binning.det_mask)
            if len(dets) == 0:
                continue
            if self.det_data not in ob.detdata or ob.detdata[self.det_data].dtype != np.float64:
                return False
            if not set(dets) <= set(ob.detdata[self.det_data].detectors):
                return False
        return True

    @staticmethod
    def _solver_flags_host(ob, solver_flags, binning, detectors=None):
        dets = ob.select_local_detectors(detectors, flagmask=binning.det_mask)
        ob.detdata.ensure(solver_flags, dtype=np.uint8, detectors=dets)
        if len(dets) == 0:
            return
        sf = ob.detdata[solver_flags]
        view = binning.pixel_pointing.view
        for iv in ob.intervals[view]:
            start = np.zeros(iv.last - iv.first, dtype=np.uint8)
            if binning.shared_flags is not None:
                start[:] = (ob.shared[binning.shared_flags].data[iv.first:iv.last] & binning.shared_flag_mask) != 0
            for d in dets:
                sf[d, iv.first:iv.last] = start
                if binning.det_flags is not None:
                    sf[d, iv.first:iv.last] |= (
                        (ob.detdata[binning.det_flags][d, iv.first:iv.last] & binning.det_flag_mask) != 0
                    ).astype(np.uint8)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.binning.requires()
        req["detdata"].append(self.det_data)
        if self.mask is not None:
            req["global"].append(self.mask)
        return req

    def _provides(self):
        return {"global": [self.amplitudes if self.amplitudes is not None else f"{self.name}_solve_amplitudes"]}


class MapMaker(Operator):
    """Generalised destriping map-maker (reference: src/toast/ops/mapmaker.py:27-811):
    SolveAmplitudes -> final covariance / hits / rcond with the map binning -> template-cleaned
    timestreams (ApplyAmplitudes) -> binned map.

    Products ``<name>_hits``, ``_cov``, ``_invcov``, ``_rcond``, ``_map``, ``_noiseweighted_map``
    (and ``_binmap`` with ``write_binmap``, ``_cleaned`` with ``save_cleaned``) stay in ``data``;
    the solved amplitudes ``<name>_solve_amplitudes`` only with ``keep_solver_products``.  With no
    templates this reduces to CovarianceAndHits + BinMap."""

    API = Int(0, help="Internal interface version for this operator")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    convergence = Float(1.0e-12, help="Relative convergence limit")
    iter_min = Int(3, help="Minimum number of iterations")
    iter_max = Int(100, help="Maximum number of iterations")
    solve_rcond_threshold = Float(1.0e-8, help="When solving, minimum value for inverse pixel condition number cut.")
    map_rcond_threshold = Float(1.0e-8, help="For final map, minimum value for inverse pixel condition number cut.")
    mask = Unicode(None, allow_none=True, help="Data key for pixel mask to use in solving.  "
                                               "First bit of pixel values is tested")
    binning = Instance(klass=Operator, help="Binning operator used for solving template amplitudes")
    template_matrix = Instance(klass=Operator, help="This must be an instance of a template matrix operator")
    map_binning = Instance(klass=Operator, help="Binning operator for final map making (default: solver binning)")
    write_binmap = Bool(False, help="If True, also bin a map of the input (undestriped) signal: <name>_binmap")
    mc_mode = Bool(False, help="If True, re-use solver flags, sparse covariances, etc")
    mc_index = Int(None, allow_none=True, help="The Monte-Carlo index")
    keep_solver_products = Bool(False, help="If True, keep the map domain solver products in data")
    keep_final_products = Bool(True, help="If True, keep the map domain products in data after write")
    save_cleaned = Bool(False, help="If True, save the template-subtracted detector timestreams")
    overwrite_cleaned = Bool(False, help="If True and save_cleaned is True, overwrite the input data")
    reset_pix_dist = Bool(False, help="Clear any existing pixel distribution.")
    fused_lhs = Bool(True, help="Let SolverLHS use the fused device-resident kernels when it can "
                                "(not a reference trait; False = the reference operator sequence)")
    pattern = Unicode(None, allow_none=True, help="Regex pattern to match against detector names. "
                                                  "Only these are mapped.")

    focalplane_key = Unicode(None, allow_none=True, help="Focalplane key for split mapmaking.")

    def _exec(self, data, detectors=None, **kwargs):
        # Split map-making: one complete run per value of a focalplane column, products named
        # <name>_<value> (mapmaker.py:728-790).
        map_binning = self.map_binning if self.map_binning is not None and self.map_binning.enabled else self.binning
        splits = data.all_detector_groups(column=self.focalplane_key, selection=detectors, flagmask=map_binning.det_mask)
        if len(splits) == 0:
            return  # no valid detectors, no mapmaking
        for split_key, split_dets in splits.items():
            if split_key == "ALL":
                self._exec_pattern(data, detectors=detectors, **kwargs)
                continue
            import re

            saved = (self.name, self.reset_pix_dist)
            self.name = f"{saved[0]}_" + re.sub(r"\s", "", str(split_key))
            self.reset_pix_dist = True
            try:
                self._exec_pattern(data, detectors=split_dets, **kwargs)
            finally:
                self.name, self.reset_pix_dist = saved

    def _exec_pattern(self, data, detectors=None, **kwargs):
        if self.pattern is None:
            return self._run(data, detectors=detectors, **kwargs)
        # Detectors that do not match are flagged invalid for the duration of the run
        # (mapmaker.py:315-333) and restored afterwards (:658-662).
        import re

        det_pat = re.compile(self.pattern)
        saved = []
        for ob in data.obs:
            saved.append(dict(ob.local_detector_flags))
            ob.update_local_detector_flags({d: ob.local_detector_flags.get(d, 0) | defaults.det_mask_invalid
                                            for d in ob.local_detectors if det_pat.match(d) is None})
        try:
            return self._run(data, detectors=detectors, **kwargs)
        finally:
            for ob, flags in zip(data.obs, saved):
                ob.local_detector_flags = flags

    def _run(self, data, detectors=None, **kwargs):
        import time as _time

        from ..accel import accel_enabled, native
        from .pointing import BuildPixelDistribution

        if self.binning is None:
            raise RuntimeError("You must set the 'binning' trait before calling exec()")
        n = self.name
        # per-realisation products carry the Monte-Carlo index (mapmaker.py:296-313)
        root = f"{n}_{self.mc_index:05d}" if (self.mc_mode and self.mc_index is not None) else n
        hits_name, cov_name, invcov_name, rcond_name = f"{n}_hits", f"{n}_cov", f"{n}_invcov", f"{n}_rcond"
        clean_name, binmap_name = f"{n}_cleaned", f"{root}_binmap"
        map_name, nw_name = f"{root}_map", f"{root}_noiseweighted_map"
        self.history, self.iteration_seconds, self.timing_log = [], [], {}

        def lap(label, t0):
            # (a phase boundary is a host-side mark: the device is waited for only when somebody wants to READ the phase
            # times -- TOAST_HIP_TRACE / TOAST_HIP_PHASE_SYNC=1 --, otherwise the host goes on into the next phase's own
            # work, e.g. the templates' bookkeeping, while the device is still busy with this one)
            if accel_enabled() and (_os.environ.get("TOAST_HIP_PHASE_SYNC", "0") == "1"
                                    or _os.environ.get("TOAST_HIP_TRACE", "0") not in ("", "0")):
                native().accel_synchronize()
            self.timing_log[label] = self.timing_log.get(label, 0.0) + (_time.time() - t0)
            if _os.environ.get("TOAST_HIP_TRACE", "0") not in ("", "0"):
                print(f"[toast_hip] phase         {label:30s} {1e3 * (_time.time() - t0):10.2f} ms", file=_sys.stderr, flush=True)
            return _time.time()

        t0 = _time.time()
        tm = self.template_matrix
        use_templates = tm is not None and len([t for t in tm.templates if t.enabled]) > 0
        final_binning = self.map_binning if (self.map_binning is not None and self.map_binning.enabled) else self.binning
        pinned = _pin_for_mapmaking(data, final_binning)
        # -- fit templates (mapmaker.py:338-379)
        amplitudes = None
        if use_templates:
            solver = SolveAmplitudes(name=n, det_data=self.det_data, convergence=self.convergence,
                                     iter_min=self.iter_min, iter_max=self.iter_max,
                                     solve_rcond_threshold=self.solve_rcond_threshold, mask=self.mask,
                                     binning=self.binning, template_matrix=tm,
                                     keep_solver_products=self.keep_solver_products, mc_mode=self.mc_mode,
                                     mc_index=self.mc_index, reset_pix_dist=self.reset_pix_dist,
                                     fused_lhs=self.fused_lhs)
            # The final binning's hits / covariance / rcond are the SOLVER's when both are built from the same samples: the
            # solver flags' first bit is exactly the final binning's own flag test (same operator, its flags and masks
            # restored after the solve), no pixel mask adds a second bit before the solver covariance is accumulated, and
            # the two condition-number cuts agree.  The reference accumulates them twice (mapmaker_templates.py:843-893,
            # mapmaker.py:438-494); here the second sweep over the pointing (33 B per detector-sample, 6.4 ms at cfg-3)
            # is skipped and the arrays change their names.  TOAST_HIP_SHARE_SOLVER_COV=0: accumulate twice.
            share = (not self.mc_mode and self.mask is None and final_binning is self.binning
                     and self.map_rcond_threshold == self.solve_rcond_threshold and not self.reset_pix_dist
                     and _os.environ.get("TOAST_HIP_SHARE_SOLVER_COV", "1") != "0")
            snm = solver._names()
            if share:
                solver._share_invcov = f"{n}_solve_invcov"
                solver._share_keep = (snm["hits"], snm["cov"], snm["rcond"])
            solver.apply(data, detectors=detectors)
            amplitudes = solver.amplitudes
            self.history, self.iteration_seconds = solver.history, solver.iteration_seconds
            self.lhs_route = getattr(solver, "lhs_route", ())
            self.rhs_map_with_covariance = getattr(solver, "rhs_map_with_covariance", None)
            self.lhs_pack_bytes = getattr(solver, "lhs_pack_bytes", ())
            self.timing_log.update(solver.timing_log)
            t0 = _time.time()
        # -- final binning set-up (:381-436)
        map_binning = self.map_binning if (self.map_binning is not None and self.map_binning.enabled) else self.binning
        map_binning.pre_process = None
        map_binning.covariance = cov_name
        if self.reset_pix_dist:
            for key in (hits_name, cov_name, invcov_name, rcond_name, binmap_name, map_name, nw_name,
                        map_binning.pixel_dist):
                if key is not None and key in data:
                    del data[key]
            Delete(detdata=[clean_name]).apply(data)
        if map_binning.pixel_dist not in data:
            BuildPixelDistribution(pixel_dist=map_binning.pixel_dist, pixel_pointing=map_binning.pixel_pointing,
                                   save_pointing=map_binning.full_pointing).apply(data, detectors=detectors)
        # -- final covariance, hits, rcond with the map binning's own flags (:438-479); an MC
        #    realisation re-uses the existing one
        shared = False
        if use_templates and share and all(k in data for k in (snm["hits"], snm["cov"], snm["rcond"], f"{n}_solve_invcov")):
            for key in (hits_name, cov_name, invcov_name, rcond_name):
                if key in data:
                    del data[key]
            # (moved, not deleted: Data.__delitem__ would drop the device copy with the name)
            take = (lambda k: data[k].duplicate()) if self.keep_solver_products else (lambda k: data._internal.pop(k))
            data[hits_name], data[cov_name], data[rcond_name] = take(snm["hits"]), take(snm["cov"]), take(snm["rcond"])
            data[invcov_name] = data._internal.pop(f"{n}_solve_invcov")
            shared = True
        self.shared_solver_covariance = shared
        if not shared and not (self.mc_mode and cov_name in data):
            CovarianceAndHits(
                pixel_dist=map_binning.pixel_dist, covariance=cov_name, inverse_covariance=invcov_name, hits=hits_name,
                rcond=rcond_name, det_mask=map_binning.det_mask, det_flags=map_binning.det_flags,
                det_flag_mask=map_binning.det_flag_mask, det_data_units=map_binning.det_data_units,
                shared_flags=map_binning.shared_flags, shared_flag_mask=map_binning.shared_flag_mask,
                pixel_pointing=map_binning.pixel_pointing, stokes_weights=map_binning.stokes_weights,
                noise_model=map_binning.noise_model, rcond_threshold=self.map_rcond_threshold,
                sync_type=map_binning.sync_type, save_pointing=map_binning.full_pointing).apply(data, detectors=detectors)
        t0 = lap("final_covariance", t0)
        # -- undestriped map (:481-513)
        if self.write_binmap:
            map_binning.det_data, map_binning.binned, map_binning.noiseweighted = self.det_data, binmap_name, None
            map_binning.apply(data, detectors=detectors)
        # -- template-cleaned timestreams (:515-559)
        out_cleaned = self.det_data
        fused_final = use_templates and self._fused_final_binning(data, detectors, map_binning, tm, amplitudes)
        self.fused_final_binning = bool(fused_final)
        if fused_final:
            # nobody asked for the cleaned timestreams: d - M a is formed inside the accumulate kernel (41 B per
            # detector-sample instead of copy + add_to_signal + build_noise_weighted = 73 B)
            map_binning._clean = fused_final
            if not self.keep_solver_products:
                self._drop_amplitudes_after = amplitudes
        elif use_templates:
            out_cleaned = clean_name
            output = clean_name
            if self.save_cleaned and self.overwrite_cleaned:
                output, out_cleaned = None, self.det_data
            ApplyAmplitudes(op="subtract", det_data=self.det_data, amplitudes=amplitudes, template_matrix=tm,
                            output=output).apply(data, detectors=detectors)
            if not self.keep_solver_products:
                data[amplitudes].clear()
                del data[amplitudes]
            t0 = lap("apply_amplitudes", t0)
        # -- destriped map (:561-590)
        map_binning.det_data = out_cleaned
        map_binning.noiseweighted = nw_name if self.keep_final_products else None
        map_binning.binned = map_name
        try:
            map_binning.apply(data, detectors=detectors)
        finally:
            map_binning._clean = None
        if getattr(self, "_drop_amplitudes_after", None) is not None:
            key, self._drop_amplitudes_after = self._drop_amplitudes_after, None
            if key in data:
                data[key].clear()
                del data[key]
        t0 = lap("final_binning", t0)
        if use_templates and not self.save_cleaned and out_cleaned == clean_name:
            Delete(detdata=[clean_name]).apply(data)
        _unpin_for_mapmaking(data, pinned)

    def _fused_final_binning(self, data, detectors, map_binning, tm, amplitudes):
        """(template, its amplitudes) when the last two steps -- subtract the templates, bin the result -- can run as ONE
        sweep (toast_hip_offset_clean_accumulate_dev), else None: a single Offset template, the cleaned timestreams not
        asked for, cached IQU pointing on the device, every observation of a shape the kernel takes.
        TOAST_HIP_FUSED_FINAL=0 keeps the two operators."""
        from ..accel import accel_enabled
        from ..templates import Offset

        if self.save_cleaned or not accel_enabled() or _os.environ.get("TOAST_HIP_FUSED_FINAL", "1") == "0":
            return None
        from .. import capi

        if capi.get_deterministic() or amplitudes not in data:
            return None
        map_binning.det_data = self.det_data           # (what the one-sweep form reads; _on_the_fly looks at its dtype)
        # BinMap either evaluates the pointing inside the accumulate kernel (full_pointing=False, nothing cached) or runs
        # its Pipeline [pixels, weights, accumulate] over cached / freshly expanded pointing, all detectors or group by
        # group: both accumulate operators have the one-sweep form (k_otf_accumulate<.., SIG = 2, ..> / k_offset_accumulate_v2<E, true>)
        cached = not ((not map_binning.full_pointing) and map_binning._on_the_fly(data, detectors, True))
        tmpls = [t for t in tm.templates if t.enabled]
        if len(tmpls) != 1 or not isinstance(tmpls[0], Offset):
            return None
        tmpl = tmpls[0]
        pixels_op, weights_op = map_binning.pixel_pointing, map_binning.stokes_weights
        if weights_op.mode not in (("IQU",) if cached else ("I", "IQU")) or tmpl.name not in data[amplitudes]:
            return None
        if tmpl.use_noise_prior and pixels_op.view is not None:
            return None          # (with a noise prior the baselines ignore the view, offset.py:135-140)
        for iob, ob in enumerate(data.obs):
            dets = ob.select_local_detectors(detectors, flagmask=map_binning.det_mask)
            if len(dets) == 0:
                continue
            if not set(dets) <= set(tmpl._obs_dets.get(iob, ())):
                return None
            if cached and ob.n_local_samples % 2 != 0:
                return None      # (the two-samples-per-lane kernel)
            if self.det_data not in ob.detdata or not set(dets) <= set(ob.detdata[self.det_data].detectors):
                return None
            if ob.detdata[self.det_data].dtype != np.float64:
                return None
        return (tmpl, data[amplitudes][tmpl.name])

    @staticmethod
    def _solver_flags_device(ob, solver_flags, binning, detectors=None):
        """The same combination on the device (toast_hip_combine_flags_dev): nothing but the
        uint8 inputs cross PCIe, and those only if they are not resident yet."""
        from .. import capi
        from ..accel import accel_device_ptr

        dets = ob.select_local_detectors(detectors, flagmask=binning.det_mask)
        ob.detdata.ensure(solver_flags, dtype=np.uint8, detectors=dets, accel=True)
        if len(dets) == 0:
            return
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
                               ob.intervals[view].data, n_out_rows=len(sf.detectors), outside_value=0)
        sf.accel_used(True)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.binning.requires()
        req["detdata"].append(self.det_data)
        return req

    def _provides(self):
        return {"global": [f"{self.name}_map", f"{self.name}_hits", f"{self.name}_cov", f"{self.name}_rcond"]}
