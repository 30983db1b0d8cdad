"""Template regression: TemplateMatrix, SolverRHS, SolverLHS and the PCG ``solve`` loop.

Reference: src/toast/ops/mapmaker_templates.py:32-420 (TemplateMatrix),
src/toast/ops/mapmaker_solve.py:27-229 (SolverRHS), :232-521 (SolverLHS), :524-755 (solve).

    RHS:  b  = M^T N^-1 Z d          LHS:  a' = M^T N^-1 Z M a (+ prior)
    Z = I - P (P^T N^-1 P)^-1 P^T N^-1
"""

import numpy as np

from ..data import defaults
from ..templates import AmplitudesMap
from ..traits import Bool, ImplementationType, Instance, Int, List, Unicode
from .mapmaker_ops import Copy, Delete, NoiseWeight, ScanMap
from .operator import Operator
from .pipeline import Pipeline, uncached_detector_sets


class TemplateMatrix(Operator):
    """Projects amplitudes to timestreams (``transpose=False``: tod = M a, after zeroing) or
    accumulates timestreams into amplitudes (``transpose=True``: a += M^T tod)."""

    API = Int(0, help="Internal interface version for this operator")
    templates = List([], help="This should be a list of Template-derived objects")
    amplitudes = Unicode(None, allow_none=True, help="Data key for template amplitudes")
    transpose = Bool(False, help="If True, apply the transpose.")
    accumulate = Bool(False, help="transpose=False: add M a to the existing timestreams instead of zeroing them first "
                      "(not a reference trait; lets ApplyAmplitudes work without a second timestream buffer)")
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
            for tmpl in self.templates:
                if not tmpl.enabled:
                    continue
                if use_accel and hasattr(tmpl, "project_signal_multi"):
                    tmpl.project_signal_multi(all_dets, data[self.amplitudes][tmpl.name], **kwargs)
                else:
                    for d in all_dets:
                        tmpl.project_signal(d, data[self.amplitudes][tmpl.name], use_accel=use_accel, **kwargs)
        else:
            if self.amplitudes not in data:
                raise RuntimeError(f"Template amplitudes '{self.amplitudes}' do not exist in data")
            for ob in data.obs:
                dets = ob.select_local_detectors(selection=detectors, flagmask=self.det_mask)
                exists = ob.detdata.ensure(self.det_data, detectors=dets, accel=use_accel,
                                           create_units=self.det_data_units)
                if exists and not self.accumulate:
                    ob.detdata[self.det_data].reset(dets=dets)
            for tmpl in self.templates:
                if not tmpl.enabled:
                    continue
                if use_accel and hasattr(tmpl, "add_to_signal_multi"):
                    tmpl.add_to_signal_multi(all_dets, data[self.amplitudes][tmpl.name], **kwargs)
                else:
                    for d in all_dets:
                        tmpl.add_to_signal(d, data[self.amplitudes][tmpl.name], use_accel=use_accel, **kwargs)

    def _finalize(self, data, use_accel=None, **kwargs):
        if self.transpose and self.amplitudes in data:
            # make the accumulated amplitudes current on the host for the PCG algebra -- unless the host side is brought up
            # to date on access anyway (Data.lazy_host; Amplitudes.local): the solver's algebra runs on the device
            if not getattr(data, "lazy_host", False):
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
            if self.accumulate:
                req["detdata"].append(self.det_data)   # added to, not overwritten
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
    det_data = Unicode(None, allow_none=True, help="Observation detdata key for the timestream data")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired units of detector data")
    binning = Instance(klass=Operator, help="Binning operator for solving")
    template_matrix = Instance(klass=Operator, help="This must be an instance of a template matrix operator")
    fused = Bool(True, help="With one Offset template on the accelerator (pointing cached, compact-cached or on the fly): "
                            "project the timestreams in one pass that leaves them untouched instead of copy / scan / "
                            "weight / project "
                            "(not a reference trait; False = the reference operator sequence)")

    def _fused_lhs(self, data):
        """The SolverLHS whose launch context the fused tail borrows, or None when the sequence has to run."""
        if not self.fused:
            return None
        lhs = SolverLHS(name=f"{self.name}_ctx", binning=self.binning, template_matrix=self.template_matrix,
                        out=self.template_matrix.amplitudes, fused=True, packed_pointing=False)
        if not lhs._can_fuse(data):
            return None
        tmpl = [t for t in self.template_matrix.templates if t.enabled][0]
        if tmpl.use_noise_prior:
            return None     # (the prior's baselines ignore the view; keep the reference sequence there)
        for ob in data.obs:
            if self.det_data in ob.detdata and ob.detdata[self.det_data].dtype != np.float64:
                return None
        return lhs

    def _exec_fused(self, data, detectors, lhs):
        """b = M^T N^-1 (d - A C A^T N^-1 d): the binning pass, then ONE pass over the cached pointing that reads d
        (toast_hip_offset_scan_project_signal_dev) -- no copy of the timestreams, no scan_map / noise_weight /
        project_signal passes over it."""
        from .. import capi
        from ..accel import accel_device_ptr
        from ..templates import AmplitudesMap

        tm = self.template_matrix
        tm.transpose = True
        tm.det_data = self.det_data
        tm.det_data_units = self.det_data_units
        for t in tm.templates:
            t.det_data = self.det_data
        tm.initialize(data)
        tmpl = [t for t in tm.templates if t.enabled][0]
        if tm.amplitudes not in data:
            data[tm.amplitudes] = AmplitudesMap()
            data[tm.amplitudes][tmpl.name] = tmpl.zeros()
        # The context holds raw device pointers: collect until one whole pass made no allocation (an allocation that
        # fails evicts lazily retained buffers, possibly one whose pointer was already taken; see SolverLHS).
        for _attempt in range(4):
            gen0 = capi.accel_generation()
            # the binned map of the pass above is an INPUT here (the left-hand side only uses that buffer as scratch)
            SolverLHS._resident(data[self.binning.binned], self.binning.binned)
            for ob in data.obs:
                if self.det_data in ob.detdata:
                    SolverLHS._resident(ob.detdata[self.det_data], self.det_data)
            ctx = lhs._fused_prepare(data, detectors)
            if capi.accel_generation() == gen0:
                break
        else:
            raise RuntimeError("SolverRHS: device buffers keep being evicted while the right-hand side is prepared "
                               "(device memory too small for its working set)")
        D = capi.dev
        # (like TemplateMatrix(transpose): amplitudes that already exist are added to, not reset)
        for ps in ctx["passes"]:
            ob = data.obs[ps["iob"]]
            dd = ob.detdata[self.det_data]
            if ctx["on_the_fly"]:
                D.otf_offset_scan_project_signal(ps["pt"], ps["step"], ps["ao"], ps["nav"], dd.indices(ps["dets"]),
                                                 accel_device_ptr(dd.buffer), ctx["out_ptr"], ctx["in_flags_ptr"],
                                                 ctx["g2l_ptr"], ctx["zmap_ptr"], ctx["nps"], ps["pf_idx"],
                                                 ps["pf_ptr"], ps["pf_n"], ctx["tmpl_flag_mask"], ps["detw"],
                                                 ps["n_samp"], ps["ivl"])
                continue
            D.offset_scan_project_signal(ps["step"], ps["ao"], ps["nav"], dd.indices(ps["dets"]),
                                         accel_device_ptr(dd.buffer), ctx["out_ptr"], ctx["in_flags_ptr"],
                                         ctx["g2l_ptr"], ctx["zmap_ptr"], ctx["nps"], ctx["nnz"], ps["pi"], ps["pp"],
                                         ps["wi"], ps["wp"], ps["pf_idx"], ps["pf_ptr"], ctx["tmpl_flag_mask"],
                                         ps["detw"], ps["n_samp"], ps["ivl"])
        ctx["amps_out"].accel_used(True)
        tm._finalize(data)

    def _exec(self, data, detectors=None, **kwargs):
        for trait in ("det_data", "binning", "template_matrix"):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        self.binning.det_data = self.det_data
        self.binning.det_data_units = self.det_data_units
        self.binning.apply(data, detectors=detectors)
        lhs = self._fused_lhs(data)
        if lhs is not None:
            self._exec_fused(data, detectors, lhs)
            return
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
            proj_pipe = Pipeline(detector_sets=uncached_detector_sets(), operators=ops)
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
    fused = Bool(True, help="Use the fused device-resident kernels when the template is an Offset template "
                            "(pointing cached, compact-cached or evaluated on the fly)")
    packed_pointing = Bool(True, help="With cached IQU pointing: let the fused kernels sweep a packed copy of it "
                                      "(20 instead of 33 bytes per detector-sample: local pixel offset and flag bits in "
                                      "one word, the Q / U weights, the intensity weight once per detector), built on "
                                      "the device at the first iteration and released at the end of the solve")

    # -- fused path ---------------------------------------------------------------------------
    def _can_fuse(self, data):
        from ..accel import accel_enabled
        from ..templates import Offset

        if not (self.fused and accel_enabled()):
            return False
        from .. import capi

        if capi.get_deterministic():
            # debug mode: only build_noise_weighted has an order-deterministic form (deterministic.hip)
            return False
        if not self.binning.full_pointing:
            # uncached pointing: only through the on-the-fly kernels
            from .pointing import otf_supported

            if not (getattr(self.binning, "on_the_fly", False)
                    and otf_supported(self.binning.pixel_pointing, self.binning.stokes_weights)):
                return False
        tmpls = [t for t in self.template_matrix.templates if t.enabled]
        if len(tmpls) != 1 or not isinstance(tmpls[0], Offset):
            return False
        if tmpls[0].use_noise_prior and self.binning.pixel_pointing.view is not None:
            # with a noise prior the baselines ignore the view (offset.py:135-140); the fused
            # kernels take the baseline boundaries from the pointing view
            return False
        if self.binning.stokes_weights.mode not in ("I", "IQU"):
            return False
        return True

    @staticmethod
    def _resident(obj, name):
        """Make the device copy of an AcceleratorObject current and leave it there."""
        if getattr(obj, "buffer", None) is not None and obj.buffer.size == 0:
            return obj   # an observation without valid detectors: nothing to hold
        if not obj.accel_exists():
            obj.accel_create(name)
        if not obj.accel_in_use():
            obj.accel_update_device()
        return obj

    def __del__(self):
        try:
            self.release_packed()
        except Exception:     # noqa: BLE001 -- interpreter shutdown: the library may be gone already
            pass

    def release_packed(self):
        """Give the packed pointing cache back (and forget the recorded launches that point into it)."""
        packed = self.__dict__.pop("_packed", None)
        if packed:
            from .. import capi

            self.__dict__.pop("_fused_plan", None)
            for _key, pk in packed.values():
                self._free_pack(pk)

    @staticmethod
    def _room_for_pack(nbytes):
        """The packed cache is an extra: it is built only while at least a fifth of the device stays free afterwards (a
        data set of many observations keeps the on-the-fly / unpacked sweeps for those that do not fit)."""
        from .. import capi

        free, total = capi.accel_mem_info()
        return free - nbytes >= total // 5

    @staticmethod
    def _free_pack(pk):
        from .. import capi

        for ptr, nbytes in pk["blocks"]:
            capi.device_release(ptr, nbytes)     # (kept by the manager for the next solve's copy of the same size)

    def _pack_pass(self, c, ps):
        """The packed copy of one observation's cached pointing (toast_hip_offset_pack_pointing_dev), or None when it
        cannot be had: switched off, rows of an odd length, an intensity weight that varies along a row, no memory."""
        import os

        from .. import capi

        if not self.packed_pointing or os.environ.get("TOAST_HIP_PACKED_POINTING", "1") == "0":
            return None
        if not getattr(self, "keep_on_device", False):
            # The pack is a SNAPSHOT of pixels, weights and flags, keyed by their device addresses: it lives inside a
            # solve (solve() sets keep_on_device and gives the pack back in its finally block).  A stand-alone apply()
            # sweeps the live arrays -- an in-place edit followed by accel_update_device keeps the address.
            return None
        n_det, n_samp = len(ps["dets"]), int(ps["n_samp"])
        if c["nnz"] != 3 or n_samp % 2 != 0 or n_det == 0:
            return None
        # One packed copy per observation for the whole solve: the launch plan is rebuilt when the amplitude vectors
        # change (starting guess, then the proposal), the pointing behind it is the same -- same arrays at the same device
        # addresses, same flags and masks, same views.
        ident = (tuple(ps["dets"]), n_samp, c["g2l_ptr"], c["nps"], ps["pp"], tuple(int(i) for i in ps["pi"]), ps["wp"],
                 tuple(int(i) for i in ps["wi"]), ps["f_ptr"], ps["f_ns"], c["det_flag_mask"], ps["s_ptr"], ps["s_n"],
                 c["shared_flag_mask"], ps["pf_ptr"], ps["pf_n"], c["tmpl_flag_mask"],
                 np.ascontiguousarray(ps["ivl"]).tobytes())
        packed = self.__dict__.setdefault("_packed", {})
        have = packed.get(ps["iob"])
        if have is not None:
            if have[0] == ident:
                return have[1]
            del packed[ps["iob"]]
            self._free_pack(have[1])
        if not self._room_for_pack(20 * n_det * n_samp):
            return None
        mine = []
        try:
            for nbytes in (4 * n_det * n_samp, 16 * n_det * n_samp, 8 * n_det):
                nbytes = max(nbytes, 16)
                mine.append((capi.device_malloc(nbytes, -2), nbytes))
        except RuntimeError:
            self._free_pack(dict(blocks=mine))
            return None
        key_ptr, qu_ptr, cal_ptr = (m[0] for m in mine)
        # the pair weight sums are written by the same sweep (toast_hip_offset_pack_pointing_onepass_dev): their block
        # is taken first and given back when the pairs turn out not to qualify
        corr_ptr, corr_bytes = 0, max(8 * ((n_det + 1) // 2) * n_samp, 16)
        if os.environ.get("TOAST_HIP_PACKED_PAIR_WEIGHTS", "1") != "0" and os.environ.get("TOAST_HIP_PACK_ONEPASS", "1") != "0":
            try:
                corr_ptr = capi.device_malloc(corr_bytes, -2)
            except RuntimeError:
                corr_ptr = 0
        tried_onepass = bool(corr_ptr)
        if corr_ptr:
            ok, pair, pair_weights = capi.dev.offset_pack_pointing_onepass(
                c["g2l_ptr"], c["nps"], ps["pi"], ps["pp"], ps["wi"], ps["wp"], ps["f_idx"], ps["f_ptr"], ps["f_ns"],
                c["det_flag_mask"], ps["s_ptr"], ps["s_n"], c["shared_flag_mask"], ps["pf_idx"], ps["pf_ptr"], ps["pf_n"],
                c["tmpl_flag_mask"], n_samp, ps["ivl"], key_ptr, qu_ptr, cal_ptr, corr_ptr)
            if ok and pair_weights:
                mine.append((corr_ptr, corr_bytes))
            else:
                capi.device_release(corr_ptr, corr_bytes)
                corr_ptr = 0
        else:
            ok, pair = capi.dev.offset_pack_pointing(
                c["g2l_ptr"], c["nps"], ps["pi"], ps["pp"], ps["wi"], ps["wp"], ps["f_idx"], ps["f_ptr"], ps["f_ns"],
                c["det_flag_mask"], ps["s_ptr"], ps["s_n"], c["shared_flag_mask"], ps["pf_idx"], ps["pf_ptr"], ps["pf_n"],
                c["tmpl_flag_mask"], n_samp, ps["ivl"], key_ptr, qu_ptr, cal_ptr)
        if not ok:
            self._free_pack(dict(blocks=mine))
            return None
        pk = dict(key=key_ptr, qu=qu_ptr, cal=cal_ptr, pair=pair, corr=corr_ptr, blocks=mine)
        if not tried_onepass:
            self._pack_pair_weights(pk, n_det, n_samp, ps["ivl"])
        packed[ps["iob"]] = (ident, pk)
        return pk

    @staticmethod
    def _pack_pair_weights(pk, n_det, n_samp, ivl):
        """With pair words: the partner's Q / U weights as float sums q_a + q_b, u_a + u_b when that is exact for every
        sample in view (orthogonal pairs of equal calibration and efficiency; toast_hip_offset_pack_pair_weights_dev) --
        14 instead of 18 B per detector-sample in both sweeps.  Anything else leaves the pack as it is."""
        import os

        from .. import capi

        if not pk["pair"] or os.environ.get("TOAST_HIP_PACKED_PAIR_WEIGHTS", "1") == "0":
            return
        nbytes = max(8 * ((n_det + 1) // 2) * n_samp, 16)
        try:
            ptr = capi.device_malloc(nbytes, -2)
        except RuntimeError:
            return
        if capi.dev.offset_pack_pair_weights(pk["qu"], ptr, n_det, n_samp, ivl):
            pk["corr"] = ptr
            pk["blocks"] = list(pk["blocks"]) + [(ptr, nbytes)]
        else:
            capi.device_release(ptr, nbytes)

    def _pack_pass_otf(self, c, ps, ob, pixels_op, weights_op, n_submap):
        """The packed cache of an observation whose pointing is NOT cached (full_pointing=False, packed_cache=True):
        pixels and weights of a batch of detectors are expanded from the boresight into a bounded temporary
        (toast_hip_otf_pixels_healpix_dev / _stokes_weights_dev), packed into the batch's rows, and the temporary goes to
        the next batch; co-pointing pairs are merged at the end.  None when it cannot be had (see _pack_pass)."""
        import os

        from .. import capi
        from .pointing import otf_descriptor

        if os.environ.get("TOAST_HIP_PACKED_POINTING", "1") == "0" or not getattr(self, "keep_on_device", False):
            return None            # (outside a solve: see _pack_pass)
        dets = ps["dets"]
        n_det, n_samp = len(dets), int(ps["n_samp"])
        if c["nnz"] != 3 or n_samp % 2 != 0 or n_det == 0:
            return None
        ident = ("otf", tuple(dets), n_samp, c["g2l_ptr"], c["nps"], id(ob), pixels_op.nside, pixels_op.nest,
                 ps["f_ptr"], ps["f_ns"], c["det_flag_mask"], ps["s_ptr"], ps["s_n"], c["shared_flag_mask"],
                 ps["pf_ptr"], ps["pf_n"], c["tmpl_flag_mask"], np.ascontiguousarray(ps["ivl"]).tobytes())
        packed = self.__dict__.setdefault("_packed", {})
        have = packed.get(ps["iob"])
        if have is not None:
            if have[0] == ident:
                return have[1]
            del packed[ps["iob"]]
            self._free_pack(have[1])
        D = capi.dev
        batch = max(2, int((4 << 30) // (32 * n_samp)) // 2 * 2)       # <= 4 GB of expanded pointing at a time, whole pairs
        batch = min(batch, n_det + (n_det & 1))
        if not self._room_for_pack(20 * n_det * n_samp + 32 * batch * n_samp):
            return None
        mine, temps = [], []
        try:
            for nbytes in (4 * n_det * n_samp, 16 * n_det * n_samp, 8 * n_det):
                nbytes = max(nbytes, 16)
                mine.append((capi.device_malloc(nbytes, -2), nbytes))
            for nbytes in (8 * batch * n_samp, 24 * batch * n_samp, max(int(n_submap), 16)):
                temps.append((capi.device_malloc(nbytes, -2), nbytes))
        except RuntimeError:
            self._free_pack(dict(blocks=mine + temps))
            return None
        (key_ptr, _), (qu_ptr, _), (cal_ptr, _) = mine
        (pix_ptr, _), (w_ptr, _), (hsub_ptr, _) = temps
        # the pair words and the pair weight sums straight from each batch's expanded pointing (one sweep per batch,
        # toast_hip_offset_pack_pointing_onepass_dev; batches are whole pairs) -- or, when a batch's pairs do not share their
        # pixels, everything again the old way: plain words per batch, then pair check / merge / weights over all rows
        corr_ptr, corr_bytes = 0, max(8 * ((n_det + 1) // 2) * n_samp, 16)
        if (os.environ.get("TOAST_HIP_PACKED_PAIR_WEIGHTS", "1") != "0" and os.environ.get("TOAST_HIP_PACK_ONEPASS", "1") != "0"
                and self._room_for_pack(20 * n_det * n_samp + 32 * batch * n_samp + corr_bytes)):
            try:
                corr_ptr = capi.device_malloc(corr_bytes, -2)
            except RuntimeError:
                corr_ptr = 0

        def sweep(onepass):
            """-> (ok, every batch in pair words, every batch's sums usable)"""
            pairs, sums = True, True
            for b0 in range(0, n_det, batch):
                bd = dets[b0:b0 + batch]
                rows = np.arange(len(bd), dtype=np.int32)
                pt = otf_descriptor(ob, bd, pixels_op, weights_op)
                D.otf_pixels_healpix(pt, rows, pix_ptr, n_samp, ps["ivl"], hsub_ptr, n_submap, c["nps"])
                D.otf_stokes_weights(pt, rows, w_ptr, n_samp, ps["ivl"])
                pf_idx = None if ps["pf_idx"] is None else ps["pf_idx"][b0:b0 + batch]
                args = (c["g2l_ptr"], c["nps"], rows, pix_ptr, rows, w_ptr, ps["f_idx"][b0:b0 + batch], ps["f_ptr"], ps["f_ns"],
                        c["det_flag_mask"], ps["s_ptr"], ps["s_n"], c["shared_flag_mask"], pf_idx, ps["pf_ptr"], ps["pf_n"],
                        c["tmpl_flag_mask"], n_samp, ps["ivl"], key_ptr + 4 * b0 * n_samp, qu_ptr + 16 * b0 * n_samp,
                        cal_ptr + 8 * b0)
                if onepass:
                    good, pair_b, sums_b = D.offset_pack_pointing_onepass(*args, corr_ptr + 8 * (b0 // 2) * n_samp)
                    pairs, sums = pairs and pair_b, sums and sums_b
                    if good and not pair_b:
                        return True, False, False
                else:
                    good, _ = D.offset_pack_pointing(*args, pair_words=False)
                if not good:
                    return False, False, False
            return True, pairs, sums

        ok, tried_onepass = True, False
        try:
            pair, sums = False, False
            if corr_ptr:
                tried_onepass = True
                ok, pair, sums = sweep(True)
            if ok and not pair:
                tried_onepass = False
                ok, _, _ = sweep(False)
                pair = bool(ok and D.offset_pack_pairs(key_ptr, n_det, n_samp, ps["ivl"]))
        finally:
            self._free_pack(dict(blocks=temps))
        if corr_ptr and not (ok and tried_onepass and sums):
            capi.device_release(corr_ptr, corr_bytes)
            corr_ptr = 0
        if not ok:
            self._free_pack(dict(blocks=mine))
            return None
        if corr_ptr:
            mine.append((corr_ptr, corr_bytes))
        pk = dict(key=key_ptr, qu=qu_ptr, cal=cal_ptr, pair=pair, corr=corr_ptr, blocks=mine)
        if not tried_onepass:
            self._pack_pair_weights(pk, n_det, n_samp, ps["ivl"])
        packed[ps["iob"]] = (ident, pk)
        return pk

    def _fused_prepare(self, data, detectors):
        """Everything of the fused left-hand side that is not a kernel launch: residency of the
        operands, device pointers, per-observation index arrays.  Returns the launch context."""
        from ..accel import accel_device_ptr
        from ..pixels import PixelData
        from .pointing import compact_pixel_cache, otf_descriptor

        binning, tm = self.binning, self.template_matrix
        pixels_op, weights_op = binning.pixel_pointing, binning.stokes_weights
        tmpl = [t for t in tm.templates if t.enabled][0]
        amps_in = data[tm.amplitudes][tmpl.name]
        amps_out = data[self.out][tmpl.name]
        # full_pointing: cached pointing, resident on the device (computed once, reused by every
        # iteration); otherwise the kernels evaluate the pointing on the fly from the boresight
        pixels_op.detector_pointing.det_mask = binning.det_mask
        on_the_fly = not binning.full_pointing
        if not on_the_fly:
            cached = True
            for ob in data.obs:
                dets = ob.select_local_detectors(detectors, flagmask=binning.det_mask)
                for key in (pixels_op.pixels, weights_op.weights):
                    if key not in ob.detdata or not set(dets) <= set(ob.detdata[key].detectors):
                        cached = False
            if cached:
                # do not drag the detector quaternions (4x the pixel volume) to the device again
                for ob in data.obs:
                    self._resident(ob.detdata[pixels_op.pixels], pixels_op.pixels)
                    self._resident(ob.detdata[weights_op.weights], weights_op.weights)
            else:
                pixels_op.apply(data, detectors=detectors, use_accel=True)
                weights_op.apply(data, detectors=detectors, use_accel=True)
        dist = data[binning.pixel_dist]
        nnz = len(weights_op.mode)
        if binning.binned not in data:
            data[binning.binned] = PixelData(dist, np.float64, n_value=nnz)
        zmap = data[binning.binned]
        if not zmap.accel_exists():
            zmap.accel_create(binning.binned, zero_out=True)
        cov = self._resident(data[binning.covariance], binning.covariance)
        if "_g2l_" + binning.pixel_dist not in data:
            from ..data import SharedData

            data["_g2l_" + binning.pixel_dist] = SharedData(dist.global_submap_to_local, "g2l")
        g2l = self._resident(data["_g2l_" + binning.pixel_dist], "g2l")
        self._resident(amps_in, f"{tmpl.name}_in")
        if not amps_out.accel_exists():
            amps_out.accel_create(f"{tmpl.name}_out", zero_out=True)
        ctx = dict(on_the_fly=on_the_fly, nnz=nnz, nps=dist.n_pix_submap, n_local=dist.n_local_submap,
                   zmap=zmap, amps_in=amps_in, amps_out=amps_out, zmap_ptr=accel_device_ptr(zmap.buffer),
                   zmap_bytes=zmap.buffer.nbytes, cov_ptr=accel_device_ptr(cov.buffer), g2l_ptr=accel_device_ptr(g2l.data),
                   in_ptr=accel_device_ptr(amps_in.buffer), in_flags_ptr=accel_device_ptr(amps_in.local_flags),
                   out_ptr=accel_device_ptr(amps_out.buffer), out_bytes=amps_out.buffer.nbytes,
                   det_flag_mask=binning.det_flag_mask, shared_flag_mask=binning.shared_flag_mask,
                   tmpl_flag_mask=tmpl.det_flag_mask, passes=[], prior=None)
        world = dist._world()
        # several processes, one GPU each: sum over processes and covariance in one owner-computes pass on the stream
        ctx["owner_computes"] = bool(world is not None and binning.sync_type == "alltoallv" and world.device_comm()
                                     and dist.replicated)
        if tmpl.use_noise_prior and tmpl._prior is not None:
            tmpl._prior.to_device()
            ctx["prior"] = tmpl._prior
        for iob, ob in enumerate(data.obs):
            dets = [d for d in ob.select_local_detectors(detectors, flagmask=binning.det_mask)
                    if d in tmpl._obs_dets[iob]]
            if len(dets) == 0:
                continue
            noise = ob[binning.noise_model]
            n_samp = ob.n_local_samples
            ps = dict(iob=iob, dets=dets, step=tmpl._step_length(tmpl.step_time, tmpl._obs_rate[iob]),
                      ao=tmpl.det_amp_offsets(iob, dets),
                      nav=tmpl._obs_views[iob], n_samp=n_samp, ivl=ob.intervals[pixels_op.view].data,
                      detw=np.array([noise.detector_weight(d) for d in dets], dtype=np.float64))
            if binning.det_flags is not None:
                fd = self._resident(ob.detdata[binning.det_flags], binning.det_flags)
                ps.update(f_idx=fd.indices(dets), f_ptr=accel_device_ptr(fd.buffer), f_ns=n_samp)
            else:
                ps.update(f_idx=np.zeros(len(dets), np.int32), f_ptr=0, f_ns=0)
            if binning.shared_flags is not None:
                sf = self._resident(ob.shared[binning.shared_flags], binning.shared_flags)
                ps.update(s_ptr=accel_device_ptr(sf.data), s_n=n_samp)
            else:
                ps.update(s_ptr=0, s_n=0)
            if tmpl.det_flags is not None:
                pflags = tmpl._solver_flags(iob, ob, True)
                ps.update(pf_idx=ob.detdata[tmpl.det_flags].indices(dets), pf_ptr=accel_device_ptr(pflags), pf_n=n_samp)
            else:
                ps.update(pf_idx=None, pf_ptr=0, pf_n=0)
            if on_the_fly:
                compact = None
                if getattr(binning, "compact_cache", False):
                    compact = compact_pixel_cache(ob, dets, pixels_op, weights_op, dist, ctx["g2l_ptr"])
                ps["pt"] = otf_descriptor(ob, dets, pixels_op, weights_op, compact=compact)
            else:
                pd, wd = ob.detdata[pixels_op.pixels], ob.detdata[weights_op.weights]
                ps.update(pi=pd.indices(dets), pp=accel_device_ptr(pd.buffer), wi=wd.indices(dets),
                          wp=accel_device_ptr(wd.buffer))
            ctx["passes"].append(ps)
        # (last: the packed copies are read from the arrays whose pointers were just collected)
        if not on_the_fly:
            for ps in ctx["passes"]:
                ps["pk"] = self._pack_pass(ctx, ps)
        elif (getattr(binning, "packed_cache", False) and self.packed_pointing
              and not getattr(binning, "compact_cache", False)):
            for ps in ctx["passes"]:
                ps["pk"] = self._pack_pass_otf(ctx, ps, data.obs[ps["iob"]], pixels_op, weights_op, dist.n_submap)
        return ctx

    @staticmethod
    def _fused_first_half(c):
        """zmap = 0;  a_out = 0;  zmap += A^T N^-1 M a   (launches only, ``capi.dev``)"""
        from .. import capi

        D = capi.dev
        D.memset(c["zmap_ptr"], 0, c["zmap_bytes"])
        D.memset(c["out_ptr"], 0, c["out_bytes"])
        if c["prior"] is not None:
            # a_out = C_a^-1 a: the noise prior term of the left-hand side (mapmaker_solve.py:409-411)
            c["prior"].add_prior(c["amps_in"], c["amps_out"])
        for ps in c["passes"]:
            if ps.get("pk") is not None:
                pk = ps["pk"]
                D.offset_accumulate_packed(ps["step"], ps["ao"], ps["nav"], c["in_ptr"], c["in_flags_ptr"], c["zmap_ptr"],
                                           pk["key"], pk["qu"], pk["cal"], ps["detw"], ps["n_samp"], ps["ivl"],
                                           pair_words=pk["pair"], pair_corr=pk.get("corr", 0))
            elif c["on_the_fly"]:
                D.otf_offset_accumulate(ps["pt"], ps["step"], ps["ao"], ps["nav"], c["in_ptr"], c["in_flags_ptr"],
                                        c["g2l_ptr"], c["zmap_ptr"], c["nps"], ps["f_idx"], ps["f_ptr"], ps["f_ns"],
                                        ps["detw"], c["det_flag_mask"], ps["n_samp"], ps["ivl"], ps["s_ptr"],
                                        ps["s_n"], c["shared_flag_mask"])
            else:
                D.offset_accumulate(ps["step"], ps["ao"], ps["nav"], c["in_ptr"], c["in_flags_ptr"], c["g2l_ptr"],
                                    c["zmap_ptr"], c["nps"], c["nnz"], ps["pi"], ps["pp"], ps["wi"], ps["wp"],
                                    ps["f_idx"], ps["f_ptr"], ps["f_ns"], ps["detw"], c["det_flag_mask"],
                                    ps["n_samp"], ps["ivl"], ps["s_ptr"], ps["s_n"], c["shared_flag_mask"])

    @staticmethod
    def _fused_second_half(c):
        """zmap = C zmap;  a_out += M^T N^-1 (M a - A zmap)"""
        from .. import capi

        D = capi.dev
        if c.get("owner_computes"):
            # zmap = C . (sum over processes of zmap) in one owner-computes pass on this stream: RCCL reduce-scatter,
            # covariance on the owned pixel shard, all-gather (toast_hip_comm_map_reduce_apply_dev)
            D.comm_map_reduce_apply(c["n_local"] * c["nps"], c["nnz"], c["cov_ptr"], c["zmap_ptr"], reduce=True)
        else:
            D.cov_apply_diag(c["n_local"], c["nps"], c["nnz"], c["cov_ptr"], c["zmap_ptr"])
        for ps in c["passes"]:
            if ps.get("pk") is not None:
                pk = ps["pk"]
                D.offset_scan_project_packed(ps["step"], ps["ao"], ps["nav"], c["in_ptr"], c["out_ptr"], c["in_flags_ptr"],
                                             c["zmap_ptr"], pk["key"], pk["qu"], pk["cal"], ps["detw"], ps["n_samp"],
                                             ps["ivl"], pair_words=pk["pair"], pair_corr=pk.get("corr", 0))
            elif c["on_the_fly"]:
                D.otf_offset_scan_project(ps["pt"], ps["step"], ps["ao"], ps["nav"], c["in_ptr"], c["out_ptr"],
                                          c["in_flags_ptr"], c["g2l_ptr"], c["zmap_ptr"], c["nps"], ps["pf_idx"],
                                          ps["pf_ptr"], ps["pf_n"], c["tmpl_flag_mask"], ps["detw"], ps["n_samp"],
                                          ps["ivl"])
            else:
                D.offset_scan_project(ps["step"], ps["ao"], ps["nav"], c["in_ptr"], c["out_ptr"], c["in_flags_ptr"],
                                      c["g2l_ptr"], c["zmap_ptr"], c["nps"], c["nnz"], ps["pi"], ps["pp"], ps["wi"],
                                      ps["wp"], ps["pf_idx"], ps["pf_ptr"], c["tmpl_flag_mask"], ps["detw"],
                                      ps["n_samp"], ps["ivl"])

    def _exec_fused(self, data, detectors):
        """a' = M^T N^-1 (M a - A C A^T N^-1 M a) with two passes over the pointing and no
        timestream buffer: offset_accumulate -> [all-reduce] -> cov_apply_diag ->
        offset_scan_project (toast_amd/csrc/kernels.hip, otf_kernels.hip).

        The launch sequence of an iteration is recorded once (``capi.capture``: the C-ABI calls
        with their converted arguments) and replayed while the operands and the memory manager's
        generation are unchanged: the host then spends ~0.1 ms per iteration instead of
        re-deriving ~100 arguments, and the GPU does not idle between iterations."""
        from .. import capi
        from ..accel import native

        binning, tm = self.binning, self.template_matrix
        tmpl = [t for t in tm.templates if t.enabled][0]
        tm.det_data = self.det_temp
        for t in tm.templates:
            t.det_data = None  # no timestream needed
        tm.initialize(data)
        amps_in = data[tm.amplitudes][tmpl.name]
        amps_out = data[self.out][tmpl.name]
        key = (id(data), None if detectors is None else tuple(detectors), id(amps_in), id(amps_out),
               id(data.get(binning.binned)), id(data.get(binning.covariance)), id(data.get(binning.pixel_dist)),
               binning.full_pointing, bool(getattr(binning, "compact_cache", False)),
               bool(getattr(binning, "packed_cache", False)), binning.det_flags,
               binning.det_flag_mask, binning.shared_flags, binning.shared_flag_mask, binning.det_mask,
               tmpl.det_flags, tmpl.det_flag_mask, tmpl.step_time, tmpl.use_noise_prior, tmpl.precond_width,
               amps_in.accel_in_use())
        plan = self.__dict__.get("_fused_plan")
        if plan is None or plan["key"] != key or plan["generation"] != capi.accel_generation():
            # The context holds raw device pointers.  An allocation made while they are being collected
            # can fail and trigger the eviction handler (Data.accel_evict), which may free a buffer whose
            # pointer is already in the context.  Every create / delete bumps the manager's generation, so:
            # collect again until one whole pass ran without any (the second pass normally finds
            # everything resident and allocates nothing).
            for _attempt in range(4):
                gen0 = capi.accel_generation()
                ctx = self._fused_prepare(data, detectors)
                if capi.accel_generation() == gen0:
                    break
            else:
                raise RuntimeError("SolverLHS: device buffers keep being evicted while the fused left-hand "
                                   "side is prepared (device memory too small for its working set)")
            with capi.capture() as first:
                self._fused_first_half(ctx)
            with capi.capture() as second:
                self._fused_second_half(ctx)
            key = key[:4] + (id(data.get(binning.binned)),) + key[5:-1] + (True,)
            plan = dict(key=key, generation=capi.accel_generation(), ctx=ctx, first=first, second=second)
            self._fused_plan = plan
            # which sweeps the plan holds, one entry per observation: "packed" (the solver's packed pointing cache),
            # "fused" (the cached pixels / weights as they are), "fused-otf" (pointing evaluated in the kernels)
            self.last_route = tuple("packed" if ps.get("pk") is not None else ("fused-otf" if ctx["on_the_fly"] else "fused")
                                    for ps in ctx["passes"])
            # bytes per detector-sample and sweep of each packed observation: 20, 18 (pair words) or 14 (+ pair weights)
            self.last_pack_bytes = tuple((14 if ps["pk"].get("corr") else (18 if ps["pk"]["pair"] else 20))
                                         if ps.get("pk") is not None else None for ps in ctx["passes"])
        ctx = plan["ctx"]
        zmap = ctx["zmap"]
        zmap.accel_used(True)
        amps_out.accel_used(True)
        plan["first"].replay()
        if not ctx.get("owner_computes"):
            zmap.sync_allreduce()      # (device-resident: an RCCL all-reduce on the same stream; no-op for one process)
        plan["second"].replay()
        if not getattr(self, "keep_on_device", False):
            native().accel_synchronize()
            amps_out.accel_update_host()
            amps_in.accel_used(False)  # host copy is current again (never modified on the device)
        for t in tm.templates:
            t.det_data = self.det_temp

    def _zero_temp(self, data):
        for ob in data.obs:
            if self.det_temp in ob.detdata:
                ob.detdata[self.det_temp].reset()

    def _exec(self, data, detectors=None, **kwargs):
        for trait in ("binning", "template_matrix", "out"):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        if self._can_fuse(data):
            if self.out in data and not data[self.out].accel_in_use():
                data[self.out].reset()      # (device-current output: the first fused pass zeroes it itself)
            self._exec_fused(data, detectors)
            return
        self.last_route = ("sequence",)
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
            proj_pipe = Pipeline(detector_sets=uncached_detector_sets(), operators=ops)
        proj_pipe.apply(data, detectors=detectors)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.binning.requires()
        req["global"].append(self.template_matrix.amplitudes)
        return req

    def _provides(self):
        return {"global": [self.out]}


def _device_scalars(rhs, vectors):
    """The PCG can keep its scalars on the device when every amplitude vector is in use there and the dot products can
    be summed over the processes on the stream (one process, or the library's own communicator)."""
    import os

    if os.environ.get("TOAST_HIP_PCG_SCALARS", "device") == "host":
        return False
    mine = all(v.n_local > 0 and v.accel_in_use() for amps in (rhs,) + tuple(vectors) for v in amps.values())
    comm = next(iter(rhs.values()))._comm
    if comm is not None and comm.comm_world is not None:
        # a collective decision (the communicator is created collectively, and every rank must take the same loop)
        return bool(comm.device_comm()) and bool(int(comm.allreduce_scalar(1 if mine else 0, op="min")))
    return mine


def _pcg_device_scalars(data, detectors, lhs_op, result, residual, precond, proposal, lhs_out, sqsum_init, delta,
                        convergence, n_iter_min, n_iter_max, log, iteration_seconds):
    """The PCG loop of ``solve`` with alpha, beta, delta and the residual norms on the device (toast_hip_pcg_*,
    csrc/pcg.hip): same recurrence, same convergence / stall / iteration-limit tests as the reference
    (mapmaker_solve.py:660-755), evaluated by one-thread kernels between the vector kernels.  The host enqueues
    iteration k + 1 while the device works on iteration k and reads the status one iteration late; an iteration
    enqueued after the end is a no-op for the solution (alpha = 0)."""
    import time as _time

    from .. import capi
    from ..accel import accel_data_create, accel_data_delete, accel_device_ptr

    D = capi.dev
    state_host = np.zeros(D.pcg_state_bytes(n_iter_max), dtype=np.uint8)   # host key of the device-side state block
    accel_data_create(state_host, "pcg_state")
    try:
        d_state = accel_device_ptr(state_host)
        D.pcg_init(d_state, sqsum_init, delta, convergence, n_iter_min, n_iter_max)
        names = list(result.keys())
        first = result[names[0]]
        comm = first._comm
        reduce_dots = bool(comm is not None and comm.comm_world is not None and not first._full)

        def dot(a, b, stage):
            for i, k in enumerate(names):
                x, y = a[k], b[k]
                # one process: the stage runs in the same launch, right behind the last template's reduction
                inline = stage if (i == len(names) - 1 and not reduce_dots) else 0
                D.pcg_dot(d_state, x.n_local, x._dptr(), y._dptr(), accel_device_ptr(x.local_flags),
                          accel_device_ptr(y.local_flags), accumulate=(i > 0), stage=inline)
            if reduce_dots:
                D.pcg_stage(d_state, stage, allreduce=True)     # sum over the ranks on the stream, then the stage

        def axpby(y, a_sel, x, b_sel):
            for k in names:
                D.pcg_axpby(d_state, y[k].n_local, a_sel, x[k]._dptr(), b_sel, y[k]._dptr())

        # The two updates whose output a dot product reads run inside that dot product's launch (same bits as the
        # separate launches; TOAST_HIP_PCG_FUSE=0 keeps them apart): result / residual update + r . r, and -- for
        # templates whose preconditioner is a diagonal -- z = M^-1 r + z . r.
        import os

        fuse = os.environ.get("TOAST_HIP_PCG_FUSE", "1") != "0"
        templates = {t.name: t for t in lhs_op.template_matrix.templates if t.enabled}

        def step_and_norm():
            if not fuse:
                for k in names:       # result += alpha * proposal;  residual -= alpha * lhs_out
                    D.pcg_step(d_state, result[k].n_local, proposal[k]._dptr(), result[k]._dptr(),
                               lhs_out[k]._dptr(), residual[k]._dptr())
                dot(residual, residual, 2)
                return
            for i, k in enumerate(names):
                inline = 2 if (i == len(names) - 1 and not reduce_dots) else 0
                D.pcg_step_dot(d_state, result[k].n_local, proposal[k]._dptr(), result[k]._dptr(), lhs_out[k]._dptr(),
                               residual[k]._dptr(), accel_device_ptr(residual[k].local_flags), accumulate=(i > 0),
                               stage=inline)
            if reduce_dots:
                D.pcg_stage(d_state, 2, allreduce=True)

        def precondition_and_project():
            diag = {k: (templates[k].precond_diag_device() if (fuse and k in templates and
                                                                hasattr(templates[k], "precond_diag_device")) else None)
                    for k in names}
            if not any(v is not None for v in diag.values()):
                lhs_op.template_matrix.apply_precond(residual, precond)
                dot(precond, residual, 3)
                return
            for i, k in enumerate(names):
                inline = 3 if (i == len(names) - 1 and not reduce_dots) else 0
                r, z = residual[k], precond[k]
                if diag[k] is not None:
                    D.pcg_precond_diag_dot(d_state, r.n_local, diag[k], r._dptr(), accel_device_ptr(r.local_flags),
                                           z._dptr(), accel_device_ptr(z.local_flags), accumulate=(i > 0), stage=inline)
                else:
                    if k in templates:
                        templates[k].apply_precond(r, z)
                    D.pcg_dot(d_state, z.n_local, z._dptr(), r._dptr(), accel_device_ptr(z.local_flags),
                              accel_device_ptr(r.local_flags), accumulate=(i > 0), stage=inline)
            if reduce_dots:
                D.pcg_stage(d_state, 3, allreduce=True)

        seen = 0       # relative residuals reported so far (the log lags one iteration behind the device)
        for it in range(n_iter_max + 1):
            if iteration_seconds is not None:
                iteration_seconds.append(_time.perf_counter())
            lhs_op.apply(data, detectors=detectors)
            dot(proposal, lhs_out, 1)
            step_and_norm()
            precondition_and_project()
            axpby(proposal, D.PCG_LIVE, precond, D.PCG_BETA)         # proposal = precond + beta * proposal
            st = D.pcg_status(d_state, lag=1)                        # the status after the PREVIOUS iteration
            if log is not None and st.n_history > seen:
                log(f"MapMaker iteration {st.n_history - 1:4d}, relative residual = {st.relative:0.6e}")
            seen = int(st.n_history)
            if st.done != 0:
                break
        history, final = D.pcg_history(d_state, n_iter_max)
        if final.done == 3:
            raise RuntimeError("Residual is not finite")
        if log is not None:
            log(f"MapMaker PCG finished after {len(history)} iterations ({D.PCG_DONE.get(int(final.done), '?')}), "
                f"relative residual = {final.relative:0.6e}")
        if iteration_seconds is not None and len(iteration_seconds) > len(history):
            del iteration_seconds[len(history):]     # the speculative iteration after the end
        return [float(x) for x in history]
    finally:
        accel_data_delete(state_host, "pcg_state")


def solve(data, detectors, lhs_op, rhs_key, result_key, convergence=1.0e-12, n_iter_min=3, n_iter_max=100,
          log=None, iteration_seconds=None):
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

    # With the fused device path every PCG vector stays resident on the GPU: the per-iteration
    # algebra (3 dots, 3 axpby, the preconditioner) runs in device kernels and only the three
    # scalars cross PCIe.  Otherwise the vectors live on the host as in the reference.
    on_device = bool(getattr(lhs_op, "_can_fuse", lambda d: False)(data))

    def place(amps, name):
        if on_device:
            amps.accel_resident(name)
        return amps

    place(rhs, rhs_key)
    place(result, result_key)
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

    try:
        # residual of the starting guess
        lhs_op.template_matrix.amplitudes = result_key
        lhs_op.out = lhs_out_key
        lhs_op.keep_on_device = on_device
        lhs_op.apply(data, detectors=detectors)
        residual = rhs.duplicate()
        residual -= lhs_out
        precond = rhs.duplicate()
        precond.reset()
        lhs_op.template_matrix.apply_precond(residual, precond)
        proposal.copy_from(precond)
        lhs_op.template_matrix.amplitudes = proposal_key

        sqsum = rhs.dot(rhs)
        sqsum_init = sqsum
        sqsum_best = sqsum
        last_best = sqsum
        delta = proposal.dot(residual)
        history = []
        import time as _time

        if on_device and _device_scalars(rhs, (result, residual, precond, proposal, lhs_out)):
            if not np.isfinite(sqsum):
                raise RuntimeError("Residual is not finite")
            history = _pcg_device_scalars(data, detectors, lhs_op, result, residual, precond, proposal, lhs_out, sqsum_init,
                                          delta, convergence, n_iter_min, n_iter_max, log, iteration_seconds)
            n_iter_max = 0       # the host-scalar loop below does not run

        for it in range(n_iter_max):
            if iteration_seconds is not None:
                # every iteration ends in a dot product (a device synchronisation): wall time per
                # iteration is the time between those points
                iteration_seconds.append(_time.perf_counter())
            if not np.isfinite(sqsum):
                raise RuntimeError("Residual is not finite")
            lhs_op.apply(data, detectors=detectors)
            alpha = delta / proposal.dot(lhs_out)
            # result += alpha * proposal ; residual -= alpha * lhs_out  (mapmaker_solve.py:683-701,
            # same roundings as the reference's scaled temporary)
            result.axpby(alpha, proposal)
            residual.axpby(-alpha, lhs_out)
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
            # proposal = beta * proposal + precond
            proposal.axpby(1.0, precond, beta)
        if iteration_seconds is not None and len(iteration_seconds) > 0:
            iteration_seconds.append(_time.perf_counter())
            stamps = list(iteration_seconds)
            iteration_seconds[:] = [b - a for a, b in zip(stamps[:-1], stamps[1:])]
    finally:
        # The packed pointing cache is a snapshot of pixels, weights and flags that belongs to THIS solve: whatever ends
        # it -- convergence, the iteration limit, a non-finite residual, an error in a kernel -- gives it back, so that a
        # later use of the operator packs the arrays as they are then.
        lhs_op.keep_on_device = False
        if hasattr(lhs_op, "release_packed"):
            lhs_op.release_packed()
    # hand the solution back on the host (AmplitudesMap.accel_update_host skips host-current ones); with lazy host
    # coherence the vectors stay device-current -- ApplyAmplitudes reads them there -- and ``Amplitudes.local`` copies back
    # when the host asks
    if not getattr(data, "lazy_host", False):
        result.accel_update_host()
        rhs.accel_update_host()
    for tmp in (residual, precond):
        tmp.clear()
    proposal.clear()
    del data[proposal_key]
    lhs_out.clear()
    del data[lhs_out_key]
    return history
