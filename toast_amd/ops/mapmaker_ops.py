"""Map-domain operators of the hot path: ScanMap, NoiseWeight, BuildNoiseWeighted,
BuildHitMap, BuildInverseCovariance, CovarianceAndHits, BinMap (+ Copy / Delete / Reset helpers).

Reference: src/toast/ops/scan_map/scan_map.py:22-215, src/toast/ops/noise_weight/noise_weight.py:20-160,
src/toast/ops/mapmaker_utils/mapmaker_utils.py:34-1271, src/toast/ops/mapmaker_binning.py:27-317,
src/toast/ops/copy.py, delete.py, reset.py.
"""

import numpy as np

from ..accel import native
from ..data import defaults
from ..pixels import PixelData, covariance_apply, covariance_invert, map_reduce_apply
from ..traits import Bool, Float, ImplementationType, Instance, Int, List, Unicode
from .operator import Operator
from .pipeline import Pipeline, uncached_detector_sets

_IMPLS = [ImplementationType.DEFAULT, ImplementationType.COMPILED]


def _global_to(obj, name, use_accel, zero_new=False):
    """Move a global PixelData where the kernel will look for it (mapmaker_utils.py:745-773)."""
    if use_accel:
        if not obj.accel_exists():
            obj.accel_create(name, zero_out=zero_new)
            if zero_new:
                obj.accel_used(True)
            else:
                obj.accel_update_device()
        elif not obj.accel_in_use():
            obj.accel_update_device()
    elif obj.accel_in_use():
        obj.accel_update_host()


class ScanMap(Operator):
    """Scan a map into detector timestreams: ``tod (+|-)= sum_k w_k map[pix, k]``."""

    API = Int(0, help="Internal interface version for this operator")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for accumulating output")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Output units if creating detector data")
    det_mask = Int(defaults.det_mask_invalid, help="Bit mask value for per-detector flagging")
    det_flag_mask = Int(defaults.det_mask_invalid, help="Bit mask value for detector sample flagging")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    pixels = Unicode(defaults.pixels, help="Observation detdata key for pixel indices")
    weights = Unicode(defaults.weights, allow_none=True, help="Observation detdata key for Stokes weights")
    map_key = Unicode(None, allow_none=True, help="The Data key where the map is located")
    subtract = Bool(False, help="If True, subtract the map timestream instead of accumulating")
    zero = Bool(False, help="If True, zero the data before accumulating / subtracting")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        if self.det_data is None:
            raise RuntimeError("You must set the det_data trait before calling exec()")
        if self.map_key is None:
            raise RuntimeError("You must set the map_key trait before calling exec()")
        if self.map_key not in data:
            raise RuntimeError("The map_key '{}' does not exist in the data".format(self.map_key))
        map_data = data[self.map_key]
        if not isinstance(map_data, PixelData):
            raise RuntimeError("The map to scan must be a PixelData instance")
        if self.weights is None:
            raise NotImplementedError("ScanMap without Stokes weights is not on the compiled path")
        map_dist = map_data.distribution
        _global_to(map_data, self.map_key, use_accel)
        name = {np.dtype(np.float64): "ops_scan_map_float64", np.dtype(np.float32): "ops_scan_map_float32",
                np.dtype(np.int64): "ops_scan_map_int64", np.dtype(np.int32): "ops_scan_map_int32"}[map_data.dtype]
        kernel = getattr(native(), name)
        for ob in data.obs:
            dets = ob.select_local_detectors(detectors, flagmask=self.det_mask)
            check_nnz = 1
            if len(ob.detdata[self.weights].detector_shape) > 1:
                check_nnz = ob.detdata[self.weights].detector_shape[-1]
            if map_data.n_value != check_nnz:
                raise RuntimeError(f"Detector data '{self.weights}' in observation '{ob.name}' has {check_nnz} nnz "
                                   f"instead of {map_data.n_value} in the map")
            ob.detdata.ensure(self.det_data, detectors=dets, create_units=self.det_data_units, accel=use_accel)
            if len(dets) == 0:
                continue
            kernel(map_dist.global_submap_to_local, map_dist.n_pix_submap, map_data.arg(use_accel),
                   ob.detdata[self.det_data].arg(use_accel), ob.detdata[self.det_data].indices(dets),
                   ob.detdata[self.pixels].arg(use_accel), ob.detdata[self.pixels].indices(dets),
                   ob.detdata[self.weights].arg(use_accel), ob.detdata[self.weights].indices(dets),
                   ob.intervals[self.view].data, 1.0, bool(self.zero), bool(self.subtract), False, use_accel)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = {"global": [self.map_key], "meta": [], "shared": [], "detdata": [self.pixels, self.det_data],
               "intervals": []}
        if self.weights is not None:
            req["detdata"].append(self.weights)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        return {"detdata": [self.det_data]}

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return True


class ScanMask(Operator):
    """Flag detector samples that fall in masked pixels: ``det_flags |= det_flags_value`` where
    ``mask[pix] & mask_bits`` is set (reference: src/toast/ops/scan_map/scan_map.py:218-345,
    host NumPy only there).  With ``use_accel`` the same pass runs on the device copies
    (``toast_hip_scan_mask_dev``); the host branch mirrors the reference's bookkeeping."""

    API = Int(0, help="Internal interface version for this operator")
    det_mask = Int(defaults.det_mask_invalid, help="Bit mask value for per-detector flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for flags to use")
    det_flags_value = Int(defaults.det_mask_processing, help="The detector flag value to set where the mask result is non-zero")
    det_flag_mask = Int(defaults.det_mask_invalid, help="Bit mask value for detector sample flagging")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    pixels = Unicode(defaults.pixels, help="Observation detdata key for pixel indices")
    mask_key = Unicode(None, allow_none=True, help="The Data key where the mask is located")
    mask_bits = Int(255, help="The number to bitwise-and with each mask value to form the result")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        if self.det_flags is None:
            raise RuntimeError("You must set the det_flags trait before calling exec()")
        if self.mask_key is None:
            raise RuntimeError("You must set the mask_key trait before calling exec()")
        if self.mask_key not in data:
            raise RuntimeError("The mask_key '{}' does not exist in the data".format(self.mask_key))
        mask_data = data[self.mask_key]
        if not isinstance(mask_data, PixelData):
            raise RuntimeError("The mask to scan must be a PixelData instance")
        mask_dist = mask_data.distribution
        for ob in data.obs:
            dets = ob.select_local_detectors(detectors, flagmask=self.det_mask)
            if len(dets) == 0:
                continue
            if self.det_flags not in ob.detdata:
                ob.detdata.create(self.det_flags, dtype=np.uint8, detectors=ob.local_detectors)
            pd, fd = ob.detdata[self.pixels], ob.detdata[self.det_flags]
            if use_accel:
                # device pass (toast_hip_scan_mask_dev): pixels / flags / mask stay resident
                from .. import capi
                from ..accel import accel_device_ptr
                from ..data import SharedData

                for obj, nm in ((pd, self.pixels), (fd, self.det_flags), (mask_data, self.mask_key)):
                    if not obj.accel_exists():
                        obj.accel_create(nm)
                    if not obj.accel_in_use():
                        obj.accel_update_device()
                gkey = "_g2l_" + str(id(mask_dist))
                if gkey not in data:
                    data[gkey] = SharedData(mask_dist.global_submap_to_local, "g2l")
                g2l = data[gkey]
                if not g2l.accel_exists():
                    g2l.accel_create("g2l")
                if not g2l.accel_in_use():
                    g2l.accel_update_device()
                capi.dev.scan_mask(accel_device_ptr(g2l.data), accel_device_ptr(mask_data.buffer),
                                   mask_dist.n_pix_submap, self.mask_bits, self.det_flags_value, pd.indices(dets),
                                   accel_device_ptr(pd.buffer), fd.indices(dets), accel_device_ptr(fd.buffer),
                                   ob.n_local_samples, ob.intervals[self.view].data)
                continue
            for iv in ob.intervals[self.view]:
                for det in dets:
                    pix = pd[det, iv.first:iv.last]
                    local_sm, local_pix = mask_dist.global_pixel_to_submap(pix)
                    ok = local_sm >= 0
                    masked = np.zeros(pix.shape, dtype=bool)
                    masked[ok] = (mask_data.data[local_sm[ok], local_pix[ok], 0] & self.mask_bits) != 0
                    fd[det, iv.first:iv.last][masked] |= self.det_flags_value

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = {"meta": [], "global": [self.mask_key], "shared": [], "detdata": [self.pixels, self.det_flags],
               "intervals": []}
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        return {"detdata": [self.det_flags]}

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return True


class NoiseWeight(Operator):
    """Apply diagonal noise weighting ``tod *= detector_weight`` (N^-1 of the PCG)."""

    API = Int(0, help="Internal interface version for this operator")
    noise_model = Unicode(defaults.noise_model, help="The observation key containing the noise model")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    det_data = Unicode(None, allow_none=True, help="Observation detdata key for the timestream data")
    det_mask = Int(defaults.det_mask_invalid, help="Bit mask value for per-detector flagging")
    det_flag_mask = Int(defaults.det_mask_invalid, help="Bit mask value for detector sample flagging")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired timestream units")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        for ob in data.obs:
            if self.det_data not in ob.detdata:
                continue
            dets = ob.select_local_detectors(detectors, flagmask=self.det_mask)
            if len(dets) == 0:
                continue
            if self.noise_model not in ob:
                raise RuntimeError("Noise model {} does not exist in observation {}".format(self.noise_model, ob.name))
            noise = ob[self.noise_model]
            detector_weights = np.array([noise.detector_weight(d) for d in dets], dtype=np.float64)
            dd = ob.detdata[self.det_data]
            if use_accel and not dd.accel_in_use():
                if not dd.accel_exists():
                    dd.accel_create(self.det_data)
                dd.accel_update_device()
            native().noise_weight(dd.arg(use_accel), dd.indices(dets), ob.intervals[self.view].data, detector_weights,
                                  use_accel)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = {"meta": [self.noise_model], "detdata": [self.det_data], "intervals": []}
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        return {"detdata": [self.det_data]}

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return True


class _MapBuilder(Operator):
    """Shared traits / flag plumbing of the accumulate operators."""

    API = Int(0, help="Internal interface version for this operator")
    pixel_dist = Unicode(None, allow_none=True, help="The Data key containing the submap distribution")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    pixels = Unicode(defaults.pixels, help="Observation detdata key for pixel indices")
    det_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for per-detector flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for flags to use")
    det_flag_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for detector sample flagging")
    shared_flags = Unicode(defaults.shared_flags, allow_none=True, help="Observation shared key for telescope flags")
    shared_flag_mask = Int(defaults.shared_mask_nonscience, help="Bit mask value for optional telescope flagging")
    sync_type = Unicode("alltoallv", help="Communication algorithm: 'allreduce' or 'alltoallv'")

    def _validate_sync_type(self, check):
        if check not in ("allreduce", "alltoallv"):
            raise RuntimeError("Invalid communication algorithm")
        return check

    def _dist(self, data):
        if self.pixel_dist is None:
            raise RuntimeError("You must set the 'pixel_dist' trait before calling exec()")
        if self.pixel_dist not in data:
            raise RuntimeError("Data does not contain submap distribution '{}'".format(self.pixel_dist))
        return data[self.pixel_dist]

    def _flag_args(self, ob, dets, use_accel):
        if self.det_flags is not None:
            fd = ob.detdata[self.det_flags]
            if use_accel and not fd.accel_in_use():
                if not fd.accel_exists():
                    fd.accel_create(self.det_flags)
                fd.accel_update_device()
            flag_indx, flag_data = fd.indices(dets), fd.arg(use_accel)
        else:
            # reference quirk (mapmaker_utils.py:836-838): [-1]; our binding wants n_det entries
            flag_indx, flag_data = np.zeros(len(dets), dtype=np.int32), np.zeros((1, 1), dtype=np.uint8)
        if self.shared_flags is not None:
            sf = ob.shared[self.shared_flags]
            if use_accel and not sf.accel_in_use():
                if not sf.accel_exists():
                    sf.accel_create(self.shared_flags)
                sf.accel_update_device()
            shared = sf.data
        else:
            shared = np.zeros(1, dtype=np.uint8)
        return flag_indx, flag_data, shared

    def _weight_nnz(self, data, detectors):
        nnz = 0
        for ob in data.obs:
            if self.weights not in ob.detdata:
                raise RuntimeError(f"Stokes weights '{self.weights}' not in obs {ob.name}")
            shp = ob.detdata[self.weights].detector_shape
            nnz = 1 if len(shp) == 1 else shp[1]
        if data.comm.comm_world is not None:
            nnz = int(data.comm.allreduce_scalar(nnz, op="max"))
        return nnz

    def _sync(self, pd):
        # sync_alltoallv (every submap summed by its owner) and sync_allreduce give the same sums (reference test
        # src/toast/tests/ops_mapmaker_utils.py:211-397); device-resident maps: RCCL reduce-scatter + all-gather, or
        # one all-reduce, on the kernels' stream (pixels.py)
        if getattr(self, "_defer_sync", False):
            return          # the caller reduces and applies the covariance in one owner-computes pass
        if self.sync_type == "alltoallv":
            pd.sync_alltoallv()
        else:
            pd.sync_allreduce()

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return True


class BuildNoiseWeighted(_MapBuilder):
    """Accumulate the noise weighted map ``zmap += A^T N^-1 d``; finalize sums over processes."""

    zmap = Unicode("zmap", help="The Data key for the output noise weighted map")
    weights = Unicode(defaults.weights, help="Observation detdata key for Stokes weights")
    noise_model = Unicode(defaults.noise_model, help="Observation key containing the noise model")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired timestream units")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        dist = self._dist(data)
        if self.det_data is None:
            raise RuntimeError("You must set the det_data trait before calling exec()")
        if self.zmap in data:
            if data[self.zmap].distribution != dist:
                raise RuntimeError("Existing zmap '{}' has different data distribution".format(self.zmap))
            zmap = data[self.zmap]
        else:
            data[self.zmap] = PixelData(dist, np.float64, n_value=self._weight_nnz(data, detectors))
            zmap = data[self.zmap]
        _global_to(zmap, self.zmap, use_accel, zero_new=not zmap.accel_exists() and zmap.host_is_zero())
        for ob in data.obs:
            dets = ob.select_local_detectors(selection=detectors, flagmask=self.det_mask)
            if self.noise_model not in ob:
                raise RuntimeError("Noise model {} does not exist in observation {}".format(self.noise_model, ob.name))
            if len(dets) == 0:
                continue
            noise = ob[self.noise_model]
            detweights = np.array([noise.detector_weight(x) for x in dets], dtype=np.float64)
            flag_indx, flag_data, shared = self._flag_args(ob, dets, use_accel)
            clean = getattr(self, "_clean", None)
            if clean is not None:
                # zmap += A^T N^-1 (d - M a) in one pass: the cleaned timestream is formed in registers
                # (MapMaker._fused_final_binning decided that every observation qualifies)
                self._clean_accumulate(data, ob, dets, clean, dist, zmap, detweights, flag_indx, shared)
                continue
            native().build_noise_weighted(
                dist.global_submap_to_local, zmap.arg(use_accel), ob.detdata[self.pixels].indices(dets),
                ob.detdata[self.pixels].arg(use_accel), ob.detdata[self.weights].indices(dets), ob.detdata[self.weights].arg(use_accel),
                ob.detdata[self.det_data].indices(dets), ob.detdata[self.det_data].arg(use_accel), flag_indx, flag_data,
                detweights, self.det_flag_mask, ob.intervals[self.view].data, shared, self.shared_flag_mask,
                use_accel)

    def _clean_accumulate(self, data, ob, dets, clean, dist, zmap, detweights, flag_indx, shared):
        """One observation of the fused final binning (toast_hip_offset_clean_accumulate_dev): everything is on the
        device already -- the Pipeline staged this operator's inputs, the amplitudes are made resident here."""
        from .. import capi
        from ..accel import accel_device_ptr
        from ..data import SharedData
        from .mapmaker_solve import SolverLHS

        tmpl, amps = clean
        iob = data.obs.index(ob)
        SolverLHS._resident(amps, f"{tmpl.name}_amplitudes")
        if "_g2l_" + self.pixel_dist not in data:
            data["_g2l_" + self.pixel_dist] = SharedData(dist.global_submap_to_local, "g2l")
        g2l = SolverLHS._resident(data["_g2l_" + self.pixel_dist], "g2l")
        n_samp = ob.n_local_samples
        pd, wd, sd = ob.detdata[self.pixels], ob.detdata[self.weights], ob.detdata[self.det_data]
        if self.det_flags is not None:
            fd = ob.detdata[self.det_flags]
            f_ptr, f_ns = accel_device_ptr(fd.buffer), n_samp
        else:
            f_ptr, f_ns = 0, 0
        if self.shared_flags is not None:
            s_ptr, s_n = accel_device_ptr(ob.shared[self.shared_flags].data), n_samp
        else:
            s_ptr, s_n = 0, 0
        capi.dev.offset_clean_accumulate(
            tmpl._step_length(tmpl.step_time, tmpl._obs_rate[iob]), tmpl.det_amp_offsets(iob, dets), tmpl._obs_views[iob],
            accel_device_ptr(amps.buffer), accel_device_ptr(amps.local_flags), accel_device_ptr(g2l.data),
            accel_device_ptr(zmap.buffer), dist.n_pix_submap, 3, pd.indices(dets), accel_device_ptr(pd.buffer),
            wd.indices(dets), accel_device_ptr(wd.buffer), sd.indices(dets), accel_device_ptr(sd.buffer), flag_indx, f_ptr,
            f_ns, detweights, self.det_flag_mask, n_samp, ob.intervals[self.view].data, s_ptr, s_n, self.shared_flag_mask)

    def _finalize(self, data, use_accel=None, **kwargs):
        if self.zmap in data:
            self._sync(data[self.zmap])

    def _requires(self):
        req = {"global": [self.pixel_dist], "meta": [self.noise_model], "shared": [],
               "detdata": [self.pixels, self.weights, self.det_data], "intervals": []}
        if self.shared_flags is not None:
            req["shared"].append(self.shared_flags)
        if self.det_flags is not None:
            req["detdata"].append(self.det_flags)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        return {"global": [self.zmap]}


class BuildNoiseWeightedOnTheFly(BuildNoiseWeighted):
    """``zmap += A^T N^-1 d`` with the pointing evaluated inside the kernel
    (toast_hip_otf_build_noise_weighted_dev): the fused form of the reference's
    ``Pipeline(detector_sets=["SINGLE"], [pixels, weights, build_zmap])``
    (mapmaker_binning.py:265-271) -- no per-detector pointing scratch, one launch per observation."""

    pixel_pointing = Instance(klass=Operator, help="The pixel pointing operator")
    stokes_weights = Instance(klass=Operator, help="The Stokes weights operator")
    compact_cache = Bool(False, help="Keep an int32 local-pixel cache (4 B/det-sample) instead of recomputing "
                                     "the pixel in every pass")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        from .. import capi
        from ..accel import accel_device_ptr
        from ..data import SharedData
        from .pointing import compact_pixel_cache, otf_descriptor

        if not use_accel:
            raise RuntimeError("BuildNoiseWeightedOnTheFly runs on the accelerator only")
        dist = self._dist(data)
        nnz = len(self.stokes_weights.mode)
        if self.zmap in data:
            if data[self.zmap].distribution != dist:
                raise RuntimeError("Existing zmap '{}' has different data distribution".format(self.zmap))
            zmap = data[self.zmap]
        else:
            data[self.zmap] = PixelData(dist, np.float64, n_value=nnz)
            zmap = data[self.zmap]
        _global_to(zmap, self.zmap, True, zero_new=not zmap.accel_exists() and zmap.host_is_zero())
        gkey = "_g2l_" + self.pixel_dist
        if gkey not in data:
            data[gkey] = SharedData(dist.global_submap_to_local, "g2l")
        g2l = data[gkey]
        if not g2l.accel_exists():
            g2l.accel_create("g2l")
        if not g2l.accel_in_use():
            g2l.accel_update_device()
        view = self.pixel_pointing.view
        for ob in data.obs:
            dets = ob.select_local_detectors(selection=detectors, flagmask=self.det_mask)
            if self.noise_model not in ob:
                raise RuntimeError("Noise model {} does not exist in observation {}".format(self.noise_model, ob.name))
            if len(dets) == 0:
                continue
            noise = ob[self.noise_model]
            detweights = np.array([noise.detector_weight(x) for x in dets], dtype=np.float64)
            compact = None
            if self.compact_cache:
                compact = compact_pixel_cache(ob, dets, self.pixel_pointing, self.stokes_weights, dist,
                                              accel_device_ptr(g2l.data))
            pt = otf_descriptor(ob, dets, self.pixel_pointing, self.stokes_weights, compact=compact)
            n_samp = ob.n_local_samples
            dd = ob.detdata[self.det_data]
            if not dd.accel_in_use():
                if not dd.accel_exists():
                    dd.accel_create(self.det_data)
                dd.accel_update_device()
            flag_indx, flag_data, shared = self._flag_args(ob, dets, True)
            f_ptr, f_n = (accel_device_ptr(flag_data), n_samp) if self.det_flags is not None else (0, 0)
            s_ptr, s_n = (accel_device_ptr(shared), n_samp) if self.shared_flags is not None else (0, 0)
            clean = getattr(self, "_clean", None)
            if clean is not None:
                # zmap += A^T N^-1 (d - M a): the template subtraction inside the accumulate kernel (MapMaker's fused
                # final binning, here with the pointing evaluated on the fly)
                from .mapmaker_solve import SolverLHS

                tmpl, amps = clean
                iob = data.obs.index(ob)
                SolverLHS._resident(amps, f"{tmpl.name}_amplitudes")
                capi.dev.otf_offset_clean_accumulate(
                    pt, tmpl._step_length(tmpl.step_time, tmpl._obs_rate[iob]), tmpl.det_amp_offsets(iob, dets),
                    tmpl._obs_views[iob], accel_device_ptr(amps.buffer), accel_device_ptr(amps.local_flags),
                    accel_device_ptr(g2l.data), accel_device_ptr(zmap.buffer), dist.n_pix_submap, dd.indices(dets),
                    accel_device_ptr(dd.buffer), flag_indx, f_ptr, f_n, detweights, self.det_flag_mask, n_samp,
                    ob.intervals[view].data, s_ptr, s_n, self.shared_flag_mask)
                continue
            capi.dev.otf_build_noise_weighted(
                pt, accel_device_ptr(g2l.data), accel_device_ptr(zmap.buffer), dist.n_pix_submap, dd.indices(dets),
                accel_device_ptr(dd.buffer), flag_indx, f_ptr, f_n, detweights, self.det_flag_mask, n_samp,
                ob.intervals[view].data, s_ptr, s_n, self.shared_flag_mask)

    def _requires(self):
        dp = self.pixel_pointing.detector_pointing
        req = {"global": [self.pixel_dist], "meta": [self.noise_model], "shared": [dp.boresight],
               "detdata": [self.det_data], "intervals": []}
        for key in (dp.shared_flags, self.shared_flags, self.stokes_weights.hwp_angle):
            if key is not None and key not in req["shared"]:
                req["shared"].append(key)
        if self.det_flags is not None:
            req["detdata"].append(self.det_flags)
        if self.pixel_pointing.view is not None:
            req["intervals"].append(self.pixel_pointing.view)
        return req


class BuildHitMap(_MapBuilder):
    """``hits[pix] += 1`` for every unflagged sample (mapmaker_utils.py:34-260)."""

    hits = Unicode("hits", help="The Data key for the output hit map")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        dist = self._dist(data)
        if self.hits in data:
            if data[self.hits].distribution != dist:
                raise RuntimeError("Existing hits '{}' has different data distribution".format(self.hits))
            hits = data[self.hits]
        else:
            data[self.hits] = PixelData(dist, np.int64, n_value=1)
            hits = data[self.hits]
        _global_to(hits, self.hits, use_accel, zero_new=not hits.accel_exists() and hits.host_is_zero())
        for ob in data.obs:
            dets = ob.select_local_detectors(selection=detectors, flagmask=self.det_mask)
            if len(dets) == 0:
                continue
            flag_indx, flag_data, shared = self._flag_args(ob, dets, use_accel)
            native().build_hit_map(dist.global_submap_to_local, hits.arg(use_accel), ob.detdata[self.pixels].indices(dets),
                                   ob.detdata[self.pixels].arg(use_accel), flag_indx, flag_data, self.det_flag_mask,
                                   ob.intervals[self.view].data, shared, self.shared_flag_mask, use_accel)

    def _finalize(self, data, use_accel=None, **kwargs):
        if self.hits in data:
            self._sync(data[self.hits])

    def _requires(self):
        req = {"global": [self.pixel_dist], "meta": [], "shared": [], "detdata": [self.pixels], "intervals": []}
        if self.shared_flags is not None:
            req["shared"].append(self.shared_flags)
        if self.det_flags is not None:
            req["detdata"].append(self.det_flags)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        return {"global": [self.hits]}


class BuildInverseCovariance(_MapBuilder):
    """``invcov[pix, (j,k>=j)] += w_j w_k detector_weight`` (mapmaker_utils.py:262-560)."""

    inverse_covariance = Unicode("inv_covariance", help="The Data key for the output inverse covariance")
    weights = Unicode(defaults.weights, help="Observation detdata key for Stokes weights")
    noise_model = Unicode(defaults.noise_model, help="Observation key containing the noise model")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired timestream units")
    hits = Unicode(None, allow_none=True, help="Also accumulate the hit map under this Data key in the same pass over "
                   "the pointing (not a reference trait: CovarianceAndHits uses it instead of a separate BuildHitMap)")
    signal = Unicode(None, allow_none=True, help="With hits on the accelerator: also accumulate the noise-weighted map "
                     "A^T N^-1 d of this detdata key in the same pass (not a reference trait: SolveAmplitudes' covariance "
                     "pass and the right-hand side's BuildNoiseWeighted read the same pixels, weights and flags)")
    signal_map = Unicode(None, allow_none=True, help="The Data key of that noise-weighted map")

    def _signal_rides_along(self, use_accel):
        return bool(use_accel and self.hits is not None and self.signal is not None and self.signal_map is not None)

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        dist = self._dist(data)
        hits = None
        if self.hits is not None:
            if self.hits in data:
                if data[self.hits].distribution != dist:
                    raise RuntimeError("Existing hits '{}' has different data distribution".format(self.hits))
            else:
                data[self.hits] = PixelData(dist, np.int64, n_value=1)
            hits = data[self.hits]
            _global_to(hits, self.hits, use_accel, zero_new=not hits.accel_exists() and hits.host_is_zero())
        if self.inverse_covariance in data:
            if data[self.inverse_covariance].distribution != dist:
                raise RuntimeError("Existing inv cov '{}' has different data distribution".format(
                    self.inverse_covariance))
            invcov = data[self.inverse_covariance]
        else:
            nnz = self._weight_nnz(data, detectors)
            data[self.inverse_covariance] = PixelData(dist, np.float64, n_value=nnz * (nnz + 1) // 2)
            invcov = data[self.inverse_covariance]
        _global_to(invcov, self.inverse_covariance, use_accel,
                   zero_new=not invcov.accel_exists() and invcov.host_is_zero())
        zmap = None
        if self._signal_rides_along(use_accel):
            if self.signal_map not in data:
                data[self.signal_map] = PixelData(dist, np.float64, n_value=self._weight_nnz(data, detectors))
            zmap = data[self.signal_map]
            if zmap.distribution != dist:
                raise RuntimeError("Existing map '{}' has different data distribution".format(self.signal_map))
            _global_to(zmap, self.signal_map, True, zero_new=not zmap.accel_exists() and zmap.host_is_zero())
        self.signal_in_one_sweep = []
        for ob in data.obs:
            dets = ob.select_local_detectors(selection=detectors, flagmask=self.det_mask)
            if self.noise_model not in ob:
                raise RuntimeError("Noise model {} does not exist in observation {}".format(self.noise_model, ob.name))
            if len(dets) == 0:
                continue
            noise = ob[self.noise_model]
            detweights = np.array([noise.detector_weight(x) for x in dets], dtype=np.float64)
            flag_indx, flag_data, shared = self._flag_args(ob, dets, use_accel)
            if zmap is not None:
                self.signal_in_one_sweep.append(
                    self._with_signal(data, ob, dets, dist, invcov, hits, zmap, detweights, flag_indx))
                continue
            if hits is not None:
                native().build_inverse_covariance_and_hits(
                    dist.global_submap_to_local, invcov.arg(use_accel), hits.arg(use_accel),
                    ob.detdata[self.pixels].indices(dets), ob.detdata[self.pixels].arg(use_accel),
                    ob.detdata[self.weights].indices(dets), ob.detdata[self.weights].arg(use_accel), flag_indx,
                    flag_data, detweights, self.det_flag_mask, ob.intervals[self.view].data, shared,
                    self.shared_flag_mask, use_accel)
                continue
            native().build_inverse_covariance(
                dist.global_submap_to_local, invcov.arg(use_accel), ob.detdata[self.pixels].indices(dets),
                ob.detdata[self.pixels].arg(use_accel), ob.detdata[self.weights].indices(dets), ob.detdata[self.weights].arg(use_accel),
                flag_indx, flag_data, detweights, self.det_flag_mask, ob.intervals[self.view].data, shared,
                self.shared_flag_mask, use_accel)

    def _with_signal(self, data, ob, dets, dist, invcov, hits, zmap, detweights, flag_indx):
        """One observation of the covariance pass with the signal's map riding along
        (toast_hip_build_cov_hits_signal_dev); everything is on the device already (the Pipeline staged the inputs).
        Returns whether ONE kernel did all three."""
        from .. import capi
        from ..accel import accel_device_ptr
        from ..data import SharedData
        from .mapmaker_solve import SolverLHS

        if "_g2l_" + self.pixel_dist not in data:
            data["_g2l_" + self.pixel_dist] = SharedData(dist.global_submap_to_local, "g2l")
        g2l = SolverLHS._resident(data["_g2l_" + self.pixel_dist], "g2l")
        n_samp = ob.n_local_samples
        pd, wd, sd = ob.detdata[self.pixels], ob.detdata[self.weights], ob.detdata[self.signal]
        if sd.dtype != np.float64:
            raise RuntimeError(f"BuildInverseCovariance: signal '{self.signal}' must be float64")
        nnz = 1 if len(wd.detector_shape) == 1 else wd.detector_shape[1]
        if self.det_flags is not None:
            f_ptr, f_ns = accel_device_ptr(ob.detdata[self.det_flags].buffer), n_samp
        else:
            f_ptr, f_ns = 0, 0
        if self.shared_flags is not None:
            s_ptr, s_n = accel_device_ptr(ob.shared[self.shared_flags].data), n_samp
        else:
            s_ptr, s_n = 0, 0
        return capi.dev.build_cov_hits_signal(
            accel_device_ptr(g2l.data), accel_device_ptr(invcov.buffer), accel_device_ptr(hits.buffer),
            accel_device_ptr(zmap.buffer), dist.n_pix_submap, nnz, pd.indices(dets), accel_device_ptr(pd.buffer),
            wd.indices(dets), accel_device_ptr(wd.buffer), sd.indices(dets), accel_device_ptr(sd.buffer), flag_indx, f_ptr,
            f_ns, detweights, detweights, self.det_flag_mask, n_samp, ob.intervals[self.view].data, s_ptr, s_n,
            self.shared_flag_mask)

    def _finalize(self, data, use_accel=None, **kwargs):
        if self.inverse_covariance in data:
            self._sync(data[self.inverse_covariance])
        if self.hits is not None and self.hits in data:
            self._sync(data[self.hits])
        # (the signal's map is summed over the processes by whoever applies the covariance to it: BinMap)

    def _requires(self):
        req = {"global": [self.pixel_dist], "meta": [self.noise_model], "shared": [],
               "detdata": [self.pixels, self.weights], "intervals": []}
        if self.shared_flags is not None:
            req["shared"].append(self.shared_flags)
        if self.det_flags is not None:
            req["detdata"].append(self.det_flags)
        if self.signal is not None and self.signal_map is not None and self.hits is not None:
            req["detdata"].append(self.signal)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        prov = {"global": [self.inverse_covariance]}
        if self.hits is not None:
            prov["global"].append(self.hits)
            if self.signal is not None and self.signal_map is not None:
                prov["global"].append(self.signal_map)
        return prov


class CovarianceAndHits(Operator):
    """Hit map + inverse covariance + (inverted) covariance and its rcond in one pass over the
    pointing (reference: mapmaker_utils.py:927-1271)."""

    API = Int(0, help="Internal interface version for this operator")
    pixel_dist = Unicode("pixel_dist", help="The Data key where the PixelDistribution object is located")
    covariance = Unicode("covariance", help="The Data key where the covariance should be stored")
    inverse_covariance = Unicode(None, allow_none=True, help="The Data key where the inverse covariance is stored")
    hits = Unicode("hits", help="The Data key where the hits should be stored")
    rcond = Unicode("rcond", help="The Data key where the inverse condition number should be stored")
    det_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for per-detector flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for flags to use")
    det_flag_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for detector sample flagging")
    shared_flags = Unicode(defaults.shared_flags, allow_none=True, help="Observation shared key for telescope flags")
    shared_flag_mask = Int(defaults.shared_mask_nonscience, help="Bit mask value for optional telescope flagging")
    pixel_pointing = Instance(klass=Operator, help="The pixel pointing operator")
    stokes_weights = Instance(klass=Operator, help="The Stokes weights operator")
    noise_model = Unicode(defaults.noise_model, help="Observation key containing the noise model")
    rcond_threshold = Float(1.0e-8, help="Minimum value for inverse condition number cut.")
    sync_type = Unicode("alltoallv", help="Communication algorithm: 'allreduce' or 'alltoallv'")
    save_pointing = Bool(False, help="If True, do not clear detector pointing matrices")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired timestream units")
    signal = Unicode(None, allow_none=True, help="On the accelerator: also accumulate the noise-weighted map of this "
                     "detdata key in the same pass over the pointing (not a reference trait; see BuildInverseCovariance)")
    signal_map = Unicode(None, allow_none=True, help="The Data key of that noise-weighted map")

    def _exec(self, data, detectors=None, **kwargs):
        for trait in ("pixel_pointing", "stokes_weights"):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        for key in (self.hits, self.covariance, self.rcond):
            if key in data:
                del data[key]
        inv_key = self.inverse_covariance if self.inverse_covariance is not None else f"{self.name}_inv_covariance"
        if inv_key in data:
            del data[inv_key]
        if self.pixel_dist not in data:
            from .pointing import BuildPixelDistribution

            BuildPixelDistribution(pixel_dist=self.pixel_dist, pixel_pointing=self.pixel_pointing,
                                   save_pointing=self.save_pointing).apply(data, detectors=detectors)
        common = dict(pixel_dist=self.pixel_dist, view=self.pixel_pointing.view, pixels=self.pixel_pointing.pixels,
                      det_mask=self.det_mask, det_flags=self.det_flags, det_flag_mask=self.det_flag_mask,
                      shared_flags=self.shared_flags, shared_flag_mask=self.shared_flag_mask,
                      sync_type=self.sync_type)
        # hit map and inverse covariance read the same pixels and flags: one pass (the pair-merged kernel carries the
        # count along; other shapes run the two kernels behind the same call)
        build_invcov = BuildInverseCovariance(inverse_covariance=inv_key, weights=self.stokes_weights.weights,
                                              noise_model=self.noise_model, det_data_units=self.det_data_units,
                                              hits=self.hits, signal=self.signal, signal_map=self.signal_map, **common)
        accum = Pipeline(detector_sets=["ALL"] if self.save_pointing else uncached_detector_sets(),
                         operators=[self.pixel_pointing, self.stokes_weights, build_invcov])
        accum.apply(data, detectors=detectors)
        # per observation: did ONE kernel accumulate covariance, hits and the signal's map (None: the map was not asked for
        # or the pass ran on the host)
        self.signal_in_one_sweep = tuple(getattr(build_invcov, "signal_in_one_sweep", ())) \
            if (self.signal is not None and self.signal_map in data) else None
        invcov = data[inv_key]
        data[self.rcond] = PixelData(data[self.pixel_dist], np.float64, n_value=1)
        if invcov.accel_in_use():
            # accumulated on the device: copy, invert and keep it there (the binning applies it there); the host
            # sides are refreshed when somebody reads them (PixelData.data)
            cov = invcov.duplicate_on_device()
            data[self.covariance] = cov
            covariance_invert(cov, self.rcond_threshold, rcond=data[self.rcond],
                              use_alltoallv=(self.sync_type == "alltoallv"))
        else:
            cov = invcov.duplicate()
            data[self.covariance] = cov
            covariance_invert(cov, self.rcond_threshold, rcond=data[self.rcond],
                              use_alltoallv=(self.sync_type == "alltoallv"))
        if self.inverse_covariance is None:
            del data[inv_key]

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.pixel_pointing.requires()
        for k, v in self.stokes_weights.requires().items():
            req.setdefault(k, []).extend(v)
        req["meta"].append(self.noise_model)
        return req

    def _provides(self):
        prov = {"global": [self.pixel_dist, self.hits, self.covariance, self.rcond]}
        if self.inverse_covariance is not None:
            prov["global"].append(self.inverse_covariance)
        return prov


class BinMap(Operator):
    """Binned map ``C A^T N^-1 d``: Pipeline[pre_process, pixel_pointing, stokes_weights,
    BuildNoiseWeighted] followed by ``covariance_apply`` (mapmaker_binning.py:27-317)."""

    API = Int(0, help="Internal interface version for this operator")
    pixel_dist = Unicode("pixel_dist", help="The Data key where the PixelDistribution object is located")
    covariance = Unicode("covariance", help="The Data key containing the noise covariance PixelData instance")
    binned = Unicode("binned", help="The Data key where the binned map should be stored")
    noiseweighted = Unicode(None, allow_none=True, help="The Data key where the noiseweighted map should be stored")
    det_data = Unicode(defaults.det_data, help="Observation detdata key for the timestream data")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired timestream units")
    det_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for per-detector flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for flags to use")
    det_flag_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for detector sample flagging")
    shared_flags = Unicode(defaults.shared_flags, allow_none=True, help="Observation shared key for telescope flags")
    shared_flag_mask = Int(defaults.shared_mask_nonscience, help="Bit mask value for optional telescope flagging")
    pixel_pointing = Instance(klass=Operator, help="The pixel pointing operator")
    stokes_weights = Instance(klass=Operator, help="The Stokes weights operator")
    pre_process = Instance(klass=Operator, help="Optional extra operator to run prior to binning")
    noise_model = Unicode(defaults.noise_model, help="Observation key containing the noise model")
    sync_type = Unicode("alltoallv", help="Communication algorithm: 'allreduce' or 'alltoallv'")
    full_pointing = Bool(False, help="If True, expand pointing for all detectors and save")
    on_the_fly = Bool(True, help="With full_pointing=False on the accelerator, evaluate the pointing inside "
                                 "the accumulate kernel (not a reference trait; False = SINGLE pipelines)")
    compact_cache = Bool(False, help="With on_the_fly: keep an int32 local-pixel cache (4 B/det-sample) and "
                                     "evaluate only the Stokes weights on the fly (not a reference trait)")
    packed_cache = Bool(True, help="With on_the_fly (IQU): let the solver keep its packed pointing cache "
                                    "(18-20 B/det-sample: local pixel offset + flag bits, Q / U weights) for the "
                                    "duration of the solve, expanded from the boresight in batches of detectors, "
                                    "instead of evaluating the pointing in every sweep; falls back to the on-the-fly "
                                    "sweeps when the cache cannot be had (memory, nnz, odd row length); ignored with "
                                    "compact_cache (not a reference trait)")

    def _validate_sync_type(self, check):
        if check not in ("allreduce", "alltoallv"):
            raise RuntimeError("Invalid communication algorithm")
        return check

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        for trait in ("pixel_pointing", "stokes_weights", "det_data"):
            if getattr(self, trait) is None:
                raise RuntimeError(f"You must set the '{trait}' trait before calling exec()")
        if self.covariance not in data:
            raise RuntimeError(f"Data does not contain noise covariance '{self.covariance}'")
        cov = data[self.covariance]
        if self.pixel_dist not in data or cov.distribution != data[self.pixel_dist]:
            raise RuntimeError(f"Pixel distribution '{self.pixel_dist}' does not match the one used by covariance "
                               f"'{self.covariance}'")
        self.pixel_pointing.create_dist = None
        # SolveAmplitudes: the noise-weighted map of this call's timestream was accumulated by the covariance pass already
        # (same pixels, weights, flags: BuildInverseCovariance.signal) -- ONE call skips the accumulation
        ready = bool(self.__dict__.pop("_zmap_is_accumulated", False)) and self.binned in data
        if ready:
            fuse_sync = (self.noiseweighted is None and self.sync_type == "alltoallv" and data.comm.comm_world is not None)
            if fuse_sync:
                map_reduce_apply(cov, data[self.binned], sync_type=self.sync_type)
                return
            if self.sync_type == "alltoallv":
                data[self.binned].sync_alltoallv()
            else:
                data[self.binned].sync_allreduce()
            if self.noiseweighted is not None:
                data[self.noiseweighted] = data[self.binned].duplicate()
            covariance_apply(cov, data[self.binned], use_alltoallv=(self.sync_type == "alltoallv"))
            return
        if self.binned in data:
            if data[self.binned].distribution != data[self.pixel_dist]:
                raise RuntimeError(f"Pixel distribution '{self.pixel_dist}' does not match existing binned map "
                                   f"'{self.binned}'")
            data[self.binned].reset()
        self.pixel_pointing.detector_pointing.det_mask = self.det_mask
        self.pixel_pointing.detector_pointing.det_flag_mask = self.det_flag_mask
        if self.stokes_weights.has_trait("detector_pointing"):
            self.stokes_weights.detector_pointing.det_mask = self.det_mask
            self.stokes_weights.detector_pointing.det_flag_mask = self.det_flag_mask
        build_zmap = BuildNoiseWeighted(
            pixel_dist=self.pixel_dist, zmap=self.binned, view=self.pixel_pointing.view,
            pixels=self.pixel_pointing.pixels, weights=self.stokes_weights.weights, noise_model=self.noise_model,
            det_data=self.det_data, det_data_units=self.det_data_units, det_mask=self.det_mask,
            det_flags=self.det_flags, det_flag_mask=self.det_flag_mask, shared_flags=self.shared_flags,
            shared_flag_mask=self.shared_flag_mask, sync_type=self.sync_type)
        # Nobody needs the summed noise-weighted map itself: leave the sum over processes to the owner-computes pass
        # that also applies the covariance (one reduce-scatter + all-gather instead of all-reduce + full-map multiply)
        fuse_sync = (self.noiseweighted is None and self.sync_type == "alltoallv"
                     and data.comm.comm_world is not None)
        build_zmap._defer_sync = fuse_sync
        build_zmap._clean = getattr(self, "_clean", None)      # (MapMaker: bin det_data - M a without writing it)
        accum_ops = []
        if self.pre_process is not None:
            accum_ops.append(self.pre_process)
        if self._on_the_fly(data, detectors, use_accel):
            # full_pointing=False on the accelerator: the pointing is evaluated inside the
            # accumulate kernel instead of per-detector SINGLE passes through pointing scratch
            otf = BuildNoiseWeightedOnTheFly(
                pixel_dist=self.pixel_dist, zmap=self.binned, view=self.pixel_pointing.view,
                pixel_pointing=self.pixel_pointing, stokes_weights=self.stokes_weights,
                noise_model=self.noise_model, det_data=self.det_data, det_data_units=self.det_data_units,
                det_mask=self.det_mask, det_flags=self.det_flags, det_flag_mask=self.det_flag_mask,
                shared_flags=self.shared_flags, shared_flag_mask=self.shared_flag_mask, sync_type=self.sync_type,
                compact_cache=self.compact_cache)
            otf._defer_sync = fuse_sync
            otf._clean = getattr(self, "_clean", None)
            accum = Pipeline(detector_sets=["ALL"], operators=accum_ops + [otf])
            accum.apply(data, detectors=detectors, use_accel=True)
        else:
            accum = Pipeline(detector_sets=["ALL"] if self.full_pointing else uncached_detector_sets())
            accum_ops.extend([self.pixel_pointing, self.stokes_weights, build_zmap])
            accum.operators = accum_ops
            accum.apply(data, detectors=detectors, use_accel=use_accel)
        if self.noiseweighted is not None:
            data[self.noiseweighted] = data[self.binned].duplicate()
        if fuse_sync:
            # binned = C . (sum over processes of zmap): reduce-scatter, covariance on the owned pixels, all-gather
            map_reduce_apply(cov, data[self.binned], sync_type=self.sync_type)
        else:
            covariance_apply(cov, data[self.binned], use_alltoallv=(self.sync_type == "alltoallv"))

    def _on_the_fly(self, data, detectors, use_accel):
        """Pointing-on-the-fly applies when the pointing is not cached (``full_pointing=False``
        and no pixels / weights left over from an earlier full expansion), the accelerator is in
        use and the pointing operators are the standard trio."""
        from ..accel import accel_enabled
        from .pointing import _outputs_exist, otf_supported

        if self.full_pointing or not self.on_the_fly or use_accel is False or not accel_enabled():
            return False
        from .. import capi

        if capi.get_deterministic():
            return False      # the on-the-fly accumulate kernels have no order-deterministic form
        if not otf_supported(self.pixel_pointing, self.stokes_weights):
            return False
        if self.pre_process is not None and not self.pre_process.supports_accel():
            return False
        for ob in data.obs:
            if self.det_data in ob.detdata and ob.detdata[self.det_data].dtype != np.float64:
                return False
        # cached pointing of all detectors is cheaper to read than to recompute
        if (_outputs_exist(data, self.pixel_pointing.pixels, detectors, self.det_mask)
                and _outputs_exist(data, self.stokes_weights.weights, detectors, self.det_mask)):
            return False
        return True

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.pixel_pointing.requires()
        for k, v in self.stokes_weights.requires().items():
            req.setdefault(k, []).extend(v)
        req["global"].extend([self.pixel_dist, self.covariance])
        req["meta"].append(self.noise_model)
        req["detdata"].append(self.det_data)
        if self.shared_flags is not None:
            req["shared"].append(self.shared_flags)
        if self.det_flags is not None:
            req["detdata"].append(self.det_flags)
        return req

    def _provides(self):
        return {"global": [self.binned]}


class Copy(Operator):
    """Copy detdata objects ``[(src, dst), ...]`` (reference: src/toast/ops/copy.py)."""

    API = Int(0, help="Internal interface version for this operator")
    detdata = List([], help="List of tuples of Observation detdata keys to copy")
    meta = List([], help="List of tuples of Observation meta keys to copy")
    shared = List([], help="List of tuples of Observation shared keys to copy")
    intervals = List([], help="List of tuples of Observation intervals keys to copy")

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        import copy as _copy

        for ob in data.obs:
            for src, dst in self.meta:      # copy.py:83-90
                ob[dst] = _copy.deepcopy(ob[src])
            for src, dst in self.shared:    # copy.py:92-116
                obj = ob.shared[src]
                if obj.accel_in_use():
                    obj.accel_update_host()
                    obj.accel_used(True)
                if dst in ob.shared:
                    if ob.shared[dst].data.shape != obj.data.shape or ob.shared[dst].data.dtype != obj.data.dtype:
                        raise RuntimeError(f"Destination shared key {dst} exists with a different shape / dtype")
                    ob.shared[dst].data[:] = obj.data
                else:
                    ob.shared.create(dst, np.array(obj.data))
            for src, dst in self.intervals:  # copy.py:118-128
                ob.intervals[dst] = _copy.deepcopy(ob.intervals[src])
            for src, dst in self.detdata:
                s = ob.detdata[src]
                dets = ob.select_local_detectors(detectors)
                if use_accel and s.accel_in_use():
                    # device-to-device copy of the selected rows (runs of adjacent rows merged),
                    # nothing crosses PCIe
                    from .. import capi
                    from ..accel import accel_device_ptr

                    rows = sorted(int(r) for r in s.indices([x for x in dets if x in s.detectors]))
                    # (a new destination whose every row is about to be overwritten need not be cleared first)
                    ob.detdata.ensure(dst, sample_shape=s.detector_shape[1:], dtype=s.dtype, detectors=s.detectors,
                                      accel=True, zero_new=len(rows) < len(s.detectors))
                    d = ob.detdata[dst]
                    row_bytes = s.buffer[0].nbytes if len(s.detectors) else 0
                    sp, dp_ = accel_device_ptr(s.buffer), accel_device_ptr(d.buffer)
                    i = 0
                    while i < len(rows):
                        j = i
                        while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1:
                            j += 1
                        off = rows[i] * row_bytes
                        capi.dev.copy(dp_ + off, sp + off, (j - i + 1) * row_bytes)
                        i = j + 1
                    d.accel_used(True)
                    continue
                if s.accel_in_use():
                    s.accel_update_host()
                ob.detdata.ensure(dst, sample_shape=s.detector_shape[1:], dtype=s.dtype, detectors=s.detectors)
                d = ob.detdata[dst]
                if d.accel_in_use():
                    d.accel_update_host()
                for det in dets:
                    d[det] = s[det]

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        return {"detdata": [x[0] for x in self.detdata]}

    def _provides(self):
        return {"detdata": [x[1] for x in self.detdata]}

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return True


class Delete(Operator):
    """Delete observation (detdata, shared, intervals, meta) and global objects
    (reference: src/toast/ops/delete.py)."""

    API = Int(0, help="Internal interface version for this operator")
    global_meta = List([], help="List of global data dictionary keys to delete")
    meta = List([], help="List of Observation dictionary keys to delete")
    detdata = List([], help="List of Observation detdata keys to delete")
    shared = List([], help="List of Observation shared keys to delete")
    intervals = List([], help="List of tuples of Observation intervals keys to delete")

    def _exec(self, data, detectors=None, **kwargs):
        for key in self.global_meta:
            if key in data:
                del data[key]
        for ob in data.obs:
            for key in self.detdata:
                if key in ob.detdata:
                    del ob.detdata[key]
            for key in self.shared:
                if key in ob.shared:
                    obj = ob.shared[key]
                    if obj.accel_exists():
                        obj.accel_delete()
                    del ob.shared[key]
            for key in self.intervals:
                if key in ob.intervals:
                    del ob.intervals[key]
            for key in self.meta:
                if key in ob:
                    del ob[key]

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        return {"detdata": list(self.detdata), "global": list(self.global_meta), "shared": list(self.shared),
                "intervals": list(self.intervals), "meta": list(self.meta)}

    def _provides(self):
        return {}
