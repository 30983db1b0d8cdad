"""Pointing operators: PointingDetectorSimple, PixelsHealpix, StokesWeights.

Reference: src/toast/ops/pointing_detector/pointing_detector.py:22-330,
src/toast/ops/pixels_healpix/pixels_healpix.py:18-330,
src/toast/ops/stokes_weights/stokes_weights.py:20-330.  Same traits, same buffer
construction, same ``exists -> skip`` logic; the kernels are the HIP library's.
"""

import numpy as np

from ..accel import native
from ..data import defaults
from ..pixels import PixelDistribution, unify_local_submaps
from ..traits import Bool, Float, ImplementationType, Instance, Int, Unicode
from .operator import Operator

_IMPLS = [ImplementationType.DEFAULT, ImplementationType.COMPILED]


def _shared_to(ob, name, use_accel):
    """Put a shared object where the kernel will look for it (pointing_detector.py:168-177)."""
    obj = ob.shared[name]
    if use_accel:
        if not obj.accel_in_use():
            if not obj.accel_exists():
                obj.accel_create(name)
            obj.accel_update_device()
    elif obj.accel_in_use():
        obj.accel_update_host()


# J2000 coordinate transforms: the rotation matrices of src/toast/qarray.py:676-768 (equatorial ->
# galactic, equatorial -> ecliptic, ecliptic -> galactic "as in HEALPix"), converted to quaternions
# the way libtoast does (toast_math_qarray.cpp:957-991).
_COORD_MATRICES = {
    "C2G": (-0.054875539726, -0.873437108010, -0.483834985808, 0.494109453312, -0.444829589425, 0.746982251810,
            -0.867666135858, -0.198076386122, 0.455983795705),
    "C2E": (1.0, 0.0, 0.0, 0.0, 0.917482062069182, 0.397777155931914, 0.0, -0.397777155931914, 0.917482062069182),
    "E2G": (-0.054882486, -0.993821033, -0.096476249, 0.494116468, -0.110993846, 0.862281440, -0.867661702,
            -0.000346354, 0.497154957),
}


def _quat_from_rotmat(m):
    tr = m[0] + m[4] + m[8]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2.0
        q = [(m[7] - m[5]) / s, (m[2] - m[6]) / s, (m[3] - m[1]) / s, 0.25 * s]
    elif m[0] > m[4] and m[0] > m[8]:
        s = np.sqrt(1.0 + m[0] - m[4] - m[8]) * 2.0
        q = [0.25 * s, (m[1] + m[3]) / s, (m[2] + m[6]) / s, (m[7] - m[5]) / s]
    elif m[4] > m[8]:
        s = np.sqrt(1.0 + m[4] - m[0] - m[8]) * 2.0
        q = [(m[1] + m[3]) / s, 0.25 * s, (m[5] + m[7]) / s, (m[2] - m[6]) / s]
    else:
        s = np.sqrt(1.0 + m[8] - m[0] - m[4]) * 2.0
        q = [(m[2] + m[6]) / s, (m[5] + m[7]) / s, 0.25 * s, (m[3] - m[1]) / s]
    return np.array(q, dtype=np.float64)


def coordinate_rotation(coord_in, coord_out):
    """(quaternion, suffix) taking a boresight from ``coord_in`` to ``coord_out`` ('C', 'E', 'G'):
    pointing_detector.py:120-154; (None, "") when there is nothing to do."""
    if coord_in is None or coord_in == coord_out:
        return None, ""
    valid = ("C", "E", "G")
    if coord_in not in valid or coord_out not in valid:
        raise RuntimeError("coordinate systems must be one of 'C', 'E', 'G'")
    key = f"{coord_in}2{coord_out}"
    if key in _COORD_MATRICES:
        return _quat_from_rotmat(_COORD_MATRICES[key]), "_" + key
    inverse = _quat_from_rotmat(_COORD_MATRICES[f"{coord_out}2{coord_in}"])
    return inverse * np.array([-1.0, -1.0, -1.0, 1.0]), "_" + key   # qa.inv of a unit quaternion


class PointingDetectorSimple(Operator):
    """Boresight pointing x focalplane offsets -> detector quaternions."""

    API = Int(0, help="Internal interface version for this operator")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    shared_flags = Unicode(defaults.shared_flags, allow_none=True, help="Observation shared key for telescope flags")
    shared_flag_mask = Int(defaults.shared_mask_invalid, help="Bit mask value for optional flagging")
    det_mask = Int(defaults.det_mask_invalid, help="Bit mask value for per-detector flagging")
    det_flag_mask = Int(defaults.det_mask_invalid, help="Bit mask value for detector sample flagging")
    boresight = Unicode(defaults.boresight_radec, help="Observation shared key for boresight")
    hwp_angle = Unicode(defaults.hwp_angle, allow_none=True, help="Observation shared key for HWP angle")
    hwp_angle_offset = Float(0.0, help="HWP angle offset [rad] to apply when constructing deflection")
    hwp_deflection_radius = Float(None, allow_none=True,
                                  help="If non-zero, nominal detector pointing will be deflected in a circular "
                                       "pattern according to HWP phase [rad].")
    quats = Unicode(defaults.quats, allow_none=True, help="Observation detdata key for output quaternions")
    coord_in = Unicode(None, allow_none=True, help="The input boresight coordinate system ('C', 'E', 'G')")
    coord_out = Unicode(None, allow_none=True, help="The output coordinate system ('C', 'E', 'G')")

    def _validate_shared_flag_mask(self, check):
        if check < 0:
            raise RuntimeError("Flag mask should be a positive integer")
        return check

    def effective_boresight(self, ob):
        """Shared key of the boresight the kernels read: the boresight itself, or -- with
        ``hwp_deflection_radius`` -- its copy deflected by the HWP wobble.  The reference applies
        the deflection on the host only (pointing_detector.py:236-276: a rotation of
        ``hwp_deflection_radius`` about an axis 90 deg from the HWP fast axis, multiplied onto the
        boresight; ``NotImplementedError`` on accelerators); here the deflected boresight, one
        quaternion per time sample, is built once and then used like any boresight, resident on
        the device."""
        from .. import synth

        base = self.boresight
        coord_rot, suffix = coordinate_rotation(self.coord_in, self.coord_out)
        if coord_rot is not None:
            # the boresight in the output frame is computed once and kept: it is re-used by every
            # iteration of the amplitude solver (pointing_detector.py:156-175)
            base = f"{self.boresight}{suffix}"
            src = ob.shared[self.boresight].data
            if base not in ob.shared or getattr(ob.shared[base], "_coord_src", None) != id(src):
                if base in ob.shared:
                    if ob.shared[base].accel_exists():
                        ob.shared[base].accel_delete()
                    del ob.shared[base]
                ob.shared.create(base, np.ascontiguousarray(synth.quat_mult(coord_rot, src)))
                ob.shared[base]._coord_src = id(src)
        if self.hwp_deflection_radius is None or self.hwp_deflection_radius == 0:
            return base
        if self.hwp_angle is None or self.hwp_angle not in ob.shared:
            raise RuntimeError("hwp_deflection_radius needs the hwp_angle shared key")
        key = f"{base}_deflected"
        hwp = ob.shared[self.hwp_angle]
        sig = (float(self.hwp_deflection_radius), float(self.hwp_angle_offset), id(hwp.data), id(ob.shared[base].data))
        if key in ob.shared and getattr(ob.shared[key], "_deflection_sig", None) == sig:
            return key
        orientation = np.array(hwp.data, dtype=np.float64) + self.hwp_angle_offset + np.pi / 2
        half = 0.5 * float(self.hwp_deflection_radius)
        deflection = np.zeros((orientation.size, 4), dtype=np.float64)
        deflection[:, 0] = np.cos(orientation) * np.sin(half)
        deflection[:, 1] = np.sin(orientation) * np.sin(half)
        deflection[:, 3] = np.cos(half)
        if key in ob.shared:
            if ob.shared[key].accel_exists():
                ob.shared[key].accel_delete()
            del ob.shared[key]
        ob.shared.create(key, np.ascontiguousarray(synth.quat_mult(ob.shared[base].data, deflection)))
        ob.shared[key]._deflection_sig = sig
        return key

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        if (self.coord_in is None) != (self.coord_out is None):
            raise RuntimeError("Input and output coordinate systems should both be None or valid")
        for ob in data.obs:
            _shared_to(ob, self.effective_boresight(ob), use_accel)
            if self.shared_flags is not None:
                _shared_to(ob, self.shared_flags, use_accel)
        for ob in data.obs:
            dets = ob.select_local_detectors(detectors, flagmask=self.det_mask)
            exists = ob.detdata.ensure(self.quats, sample_shape=(4,), dtype=np.float64, detectors=dets,
                                       accel=use_accel)
            if len(dets) == 0 or exists:
                continue
            focalplane = ob.telescope.focalplane
            fp_quats = np.zeros((len(dets), 4), dtype=np.float64)
            for idet, d in enumerate(dets):
                fp_quats[idet, :] = focalplane[d]["quat"]
            quat_indx = ob.detdata[self.quats].indices(dets)
            flags = np.zeros(1, dtype=np.uint8) if self.shared_flags is None else ob.shared[self.shared_flags].data
            native().pointing_detector(fp_quats, ob.shared[self.effective_boresight(ob)].data, quat_indx,
                                       ob.detdata[self.quats].arg(use_accel), ob.intervals[self.view].data, flags,
                                       self.shared_flag_mask, use_accel)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = {"meta": [], "shared": [self.boresight], "detdata": [self.quats], "intervals": []}
        if self.shared_flags is not None:
            req["shared"].append(self.shared_flags)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        return {"detdata": [self.quats]}

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return True


def otf_supported(pixels_op, weights_op):
    """True when the pointing operator trio is the one the on-the-fly kernels reproduce
    (PointingDetectorSimple -> PixelsHealpix -> StokesWeights, same detector pointing)."""
    if not isinstance(pixels_op, PixelsHealpix) or not isinstance(weights_op, StokesWeights):
        return False
    dp = pixels_op.detector_pointing
    if not isinstance(dp, PointingDetectorSimple) or weights_op.detector_pointing is None:
        return False
    wp = weights_op.detector_pointing
    same = all(getattr(dp, t) == getattr(wp, t) for t in ("boresight", "shared_flags", "shared_flag_mask", "view",
                                                          "hwp_angle", "hwp_angle_offset", "hwp_deflection_radius",
                                                          "coord_in", "coord_out"))
    if not same:
        return False
    if weights_op.single_precision or weights_op.mode not in ("I", "IQU"):
        return False
    view = weights_op.view if weights_op.view is not None else wp.view
    return view == pixels_op.view


def otf_descriptor(ob, dets, pixels_op, weights_op, compact=None):
    """``toast_hip_otf_pointing`` for these detectors of one observation: the arguments the three
    pointing operators would pass to their kernels (pointing_detector.py:150-214,
    pixels_healpix.py:215-277, stokes_weights.py:200-310), with boresight / flags / HWP angle
    resident on the device.  ``compact`` = (DetectorData of int32 local pixel indices) or None."""
    from .. import capi
    from ..accel import accel_device_ptr

    dp = (pixels_op if pixels_op is not None else weights_op).detector_pointing
    bore_key = dp.effective_boresight(ob)
    _shared_to(ob, bore_key, True)
    bore = accel_device_ptr(ob.shared[bore_key].data)
    n_samp = ob.n_local_samples
    d_flags, n_flags = 0, 0
    if dp.shared_flags is not None:
        _shared_to(ob, dp.shared_flags, True)
        d_flags, n_flags = accel_device_ptr(ob.shared[dp.shared_flags].data), n_samp
    focalplane = ob.telescope.focalplane
    fp_quats = np.array([focalplane[d]["quat"] for d in dets], dtype=np.float64).reshape(len(dets), 4)
    nnz = len(weights_op.mode) if weights_op is not None else 1
    eps = np.array([focalplane[d]["pol_leakage"] for d in dets], dtype=np.float64)
    cal = (np.ones(len(dets)) if weights_op is None or weights_op.cal is None
           else np.array([ob[weights_op.cal][x] for x in dets], np.float64))
    gamma = np.zeros(len(dets), dtype=np.float64)
    d_hwp, n_hwp, extra_tab = 0, 0, 0
    if nnz == 3 and weights_op.hwp_angle is not None and weights_op.hwp_angle in ob.shared:
        _shared_to(ob, weights_op.hwp_angle, True)
        d_hwp, n_hwp = accel_device_ptr(ob.shared[weights_op.hwp_angle].data), n_samp
        gamma = np.array([focalplane[d][weights_op.fp_gamma] for d in dets], dtype=np.float64)
        # (cos 4 hwp, sin 4 hwp) per time sample, built once per observation and shared by all
        # detectors: the kernels then evaluate no transcendental per det-sample
        tkey = weights_op.hwp_angle + "_cs4"
        if tkey not in ob.shared:
            from ..data import SharedData

            ob.shared[tkey] = SharedData(np.zeros((n_samp, 2), dtype=np.float64), tkey)
            tab = ob.shared[tkey]
            tab.accel_create(tkey)
            capi.dev.hwp_table(d_hwp, n_samp, accel_device_ptr(tab.data))
            tab.accel_used(True)
        tab = ob.shared[tkey]
        if not tab.accel_exists():      # evicted: the values live on the host copy
            tab.accel_create(tkey)
            tab.accel_update_device()
        extra_tab = accel_device_ptr(tab.data)
    extra = {}
    if compact is not None:
        extra = dict(d_compact_pixels=accel_device_ptr(compact.buffer), compact_index=compact.indices(dets))
    if extra_tab:
        extra["d_hwp_table"] = extra_tab
    nside, nest = (pixels_op.nside, pixels_op.nest) if pixels_op is not None else (1, True)
    iau = bool(weights_op.IAU) if weights_op is not None else False
    return capi.otf_pointing(bore, fp_quats, nside, nest, nnz, d_shared_flags=d_flags,
                             n_shared_flags=n_flags, shared_flag_mask=dp.shared_flag_mask, d_hwp=d_hwp, n_hwp=n_hwp,
                             epsilon=eps, gamma=gamma, cal=cal, IAU=iau, **extra)


def _view_covers_all(ob, view):
    """True when the intervals of ``view`` cover every sample of the observation (then a kernel over the view writes
    every sample of its output rows)."""
    ivl = ob.intervals[view].data
    if len(ivl) == 0:
        return ob.n_local_samples == 0
    first = np.asarray(ivl["first"], dtype=np.int64)
    last = np.asarray(ivl["last"], dtype=np.int64)
    order = np.argsort(first, kind="stable")
    reach = 0
    for f, l in zip(first[order], last[order]):
        if f > reach:
            return False
        reach = max(reach, int(l))
    return reach >= ob.n_local_samples


def _skip_quaternions(op, data, detectors, use_accel):
    """True when this pointing operator can write its output straight from the boresight
    (toast_hip_otf_pixels_healpix_dev / _stokes_weights_dev): device-resident run, the plain
    detector pointing, and no detector quaternions lying around that we would be expected to use."""
    dp = op.detector_pointing
    if not (use_accel and op.skip_quaternions and isinstance(dp, PointingDetectorSimple)):
        return False
    for ob in data.obs:
        if dp.quats in ob.detdata:
            return False
    return True


def compact_pixel_cache(ob, dets, pixels_op, weights_op, dist, d_g2l):
    """Device-only int32 cache of the LOCAL map index of every det-sample
    (``toast_hip_otf_compact_pixels_dev``): 4 B/det-sample instead of the 8 + 8 nnz B of cached
    pixels + weights.  Stored as ``ob.detdata[<pixels>_compact]``; rows are filled on demand and
    reused until the pixel distribution changes."""
    from .. import capi
    from ..accel import accel_device_ptr

    name = pixels_op.pixels + "_compact"
    ob.detdata.ensure(name, dtype=np.int32, detectors=ob.local_detectors, accel=True)
    dd = ob.detdata[name]
    state = dd.__dict__.setdefault("_compact_state", {"dist": None, "filled": set()})
    sig = (id(dist), pixels_op.nside, pixels_op.nest, pixels_op.view)
    if state["dist"] != sig:
        state["dist"], state["filled"] = sig, set()
    missing = [d for d in dets if d not in state["filled"]]
    if missing:
        pt = otf_descriptor(ob, missing, pixels_op, weights_op)
        capi.dev.otf_compact_pixels(pt, d_g2l, dist.n_pix_submap, dist.n_local_submap, dd.indices(missing),
                                    accel_device_ptr(dd.buffer), ob.n_local_samples,
                                    ob.intervals[pixels_op.view].data)
        state["filled"].update(missing)
    dd.accel_used(True)
    return dd


def _outputs_exist(data, key, detectors, det_mask):
    """True when every observation already holds ``key`` for all requested detectors: the
    operator (and the detector pointing it would trigger) has nothing to do."""
    for ob in data.obs:
        dets = ob.select_local_detectors(detectors, flagmask=det_mask)
        if key not in ob.detdata or not set(dets) <= set(ob.detdata[key].detectors):
            return False
    return True


def _check_detector_pointing(op, traits):
    if op is not None:
        if not isinstance(op, Operator):
            raise RuntimeError("detector_pointing should be an Operator instance")
        for trt in traits:
            if not op.has_trait(trt):
                raise RuntimeError(f"detector_pointing operator should have a '{trt}' trait")
    return op


class PixelsHealpix(Operator):
    """Detector quaternions -> HEALPix pixel numbers (flagged samples -> -1), optionally
    collecting the hit submaps into a ``PixelDistribution``."""

    API = Int(0, help="Internal interface version for this operator")
    detector_pointing = Instance(klass=Operator, help="Operator that translates boresight pointing into detector frame")
    nside = Int(64, help="The NSIDE resolution")
    nside_submap = Int(16, help="The NSIDE of the submap resolution")
    nest = Bool(True, help="If True, use NESTED ordering instead of RING")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    pixels = Unicode(defaults.pixels, help="Observation detdata key for output pixel indices")
    create_dist = Unicode(None, allow_none=True,
                          help="Create the submap distribution for all detectors and store in the Data key specified")
    single_precision = Bool(False, help="If True, use 32bit int in output")
    skip_quaternions = Bool(True, help="On the accelerator, compute the pixels straight from the boresight "
                                       "without materialising detector quaternions (not a reference trait)")

    def _validate_detector_pointing(self, op):
        return _check_detector_pointing(op, ["view", "boresight", "shared_flags", "shared_flag_mask", "det_mask",
                                             "quats", "coord_in", "coord_out"])

    def _validate_nside(self, check):
        if check <= 0 or (check & (check - 1)) != 0:
            raise RuntimeError("Invalid NSIDE value")
        if check < self.nside_submap:
            raise RuntimeError("NSIDE value is less than nside_submap")
        return check

    def _validate_nside_submap(self, check):
        if check <= 0 or (check & (check - 1)) != 0:
            raise RuntimeError("Invalid NSIDE submap value")
        if check > self.nside:
            check = 16 if self.nside >= 16 else 1
        return check

    def _observe_nside(self, change):
        self._set_hpix(change["new"], self.nside_submap)

    def _observe_nside_submap(self, change):
        self._set_hpix(self.nside, change["new"])

    def __init__(self, **kwargs):
        if "nside" in kwargs and "nside_submap" not in kwargs and kwargs["nside"] < 16:
            kwargs["nside_submap"] = 1
        # nside_submap must be applied before nside is validated against it
        sub = kwargs.pop("nside_submap", None)
        nside = kwargs.pop("nside", None)
        super().__init__(**kwargs)
        if nside is not None and sub is not None and sub > nside:
            sub = 16 if nside >= 16 else 1
        if sub is not None and sub <= self.nside:
            self.nside_submap = sub
        if nside is not None:
            self.nside = nside
        if sub is not None:
            self.nside_submap = sub
        self._set_hpix(self.nside, self.nside_submap)

    def _set_hpix(self, nside, nside_submap):
        self._n_pix = 12 * nside**2
        self._n_pix_submap = 12 * nside_submap**2
        self._n_submap = (nside // nside_submap) ** 2
        self._local_submaps = None

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        if self.detector_pointing is None:
            raise RuntimeError("The detector_pointing trait must be set")
        if self.single_precision:
            raise NotImplementedError("the compiled kernel writes int64 pixels (as in the reference, "
                                      "ops_pixels_healpix.cpp:1184-1186)")
        if self._local_submaps is None and self.create_dist is not None:
            self._local_submaps = np.zeros(self._n_submap, dtype=np.uint8)
        quats_name = self.detector_pointing.quats
        view = self.view if self.view is not None else self.detector_pointing.view
        # (the reference expands detector pointing unconditionally, pixels_healpix.py:162; when
        # the pixels are cached there is nothing to feed and the quaternions may be gone)
        no_quats = _skip_quaternions(self, data, detectors, use_accel)
        if not no_quats and not _outputs_exist(data, self.pixels, detectors, self.detector_pointing.det_mask):
            self.detector_pointing.apply(data, detectors=detectors, use_accel=use_accel)
        for ob in data.obs:
            dets = ob.select_local_detectors(detectors, flagmask=self.detector_pointing.det_mask)
            exists = ob.detdata.ensure(self.pixels, sample_shape=(), dtype=np.int64, detectors=dets, accel=use_accel,
                                       zero_new=not _view_covers_all(ob, view))
            hit_submaps = self._local_submaps
            if hit_submaps is None:
                hit_submaps = np.zeros(self._n_submap, dtype=np.uint8)
            if exists:
                if self.create_dist is not None:
                    # pixels already computed but the caller wants the distribution
                    # (pixels_healpix.py:215-243): recover it from the host copy
                    pd = ob.detdata[self.pixels]
                    restore = False
                    if pd.accel_in_use():
                        pd.accel_update_host()
                        restore = True
                    for det in dets:
                        for iv in ob.intervals[view]:
                            p = pd[det, iv.first:iv.last]
                            good = p >= 0
                            self._local_submaps[p[good] // self._n_pix_submap] = 1
                    if restore:
                        pd.accel_update_device()
                continue
            if len(dets) == 0:
                continue
            if no_quats:
                self._exec_from_boresight(ob, dets, view, hit_submaps)
                if self._local_submaps is not None:
                    self._local_submaps[:] |= hit_submaps
                continue
            quat_indx = ob.detdata[quats_name].indices(dets)
            pix_indx = ob.detdata[self.pixels].indices(dets)
            if self.detector_pointing.shared_flags is None:
                flags = np.zeros(1, dtype=np.uint8)
            else:
                flags = ob.shared[self.detector_pointing.shared_flags].data
            native().pixels_healpix(quat_indx, ob.detdata[quats_name].arg(use_accel), flags,
                                    self.detector_pointing.shared_flag_mask, pix_indx, ob.detdata[self.pixels].arg(use_accel),
                                    ob.intervals[view].data, hit_submaps, self._n_pix_submap, self.nside,
                                    bool(self.nest), use_accel)
            if self._local_submaps is not None:
                self._local_submaps[:] |= hit_submaps

    def _exec_from_boresight(self, ob, dets, view, hit_submaps):
        """pointing_detector + pixels_healpix in one kernel, no quaternion buffer.  ``hit_submaps``
        is the host in/out array of the reference interface: staged through the memory manager."""
        from .. import capi
        from ..accel import (accel_data_create, accel_data_delete, accel_data_update_device, accel_data_update_host,
                             accel_device_ptr)

        pt = otf_descriptor(ob, dets, self, None)
        pd = ob.detdata[self.pixels]
        hs = np.ascontiguousarray(hit_submaps)
        accel_data_create(hs, "hit_submaps")
        try:
            accel_data_update_device(hs, "hit_submaps")
            capi.dev.otf_pixels_healpix(pt, pd.indices(dets), accel_device_ptr(pd.buffer), ob.n_local_samples,
                                        ob.intervals[view].data, accel_device_ptr(hs), self._n_submap,
                                        self._n_pix_submap)
            accel_data_update_host(hs, "hit_submaps")
        finally:
            accel_data_delete(hs, "hit_submaps")
        if hs is not hit_submaps:
            hit_submaps[:] = hs
        pd.accel_used(True)

    def _finalize(self, data, use_accel=None, **kwargs):
        if self.create_dist is not None:
            # Every process keeps the union of hit submaps so that map reductions are plain
            # all-reduces of one contiguous buffer; single process: unchanged.
            hits = unify_local_submaps(self._local_submaps, data.comm)
            submaps = np.arange(self._n_submap, dtype=np.int64)[hits == 1]
            data[self.create_dist] = PixelDistribution(n_pix=self._n_pix, n_submap=self._n_submap,
                                                       local_submaps=submaps, comm=data.comm)
            data[self.create_dist].nest = bool(self.nest)

    def _requires(self):
        req = self.detector_pointing.requires()
        req.setdefault("detdata", []).append(self.pixels)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        prov = self.detector_pointing.provides()
        prov["detdata"].append(self.pixels)
        if self.create_dist is not None:
            prov.setdefault("global", []).append(self.create_dist)
        return prov

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return self.detector_pointing is not None and self.detector_pointing.supports_accel()


class StokesWeights(Operator):
    """Detector quaternions (+ HWP angle) -> Stokes I/Q/U pointing weights."""

    API = Int(0, help="Internal interface version for this operator")
    detector_pointing = Instance(klass=Operator, help="Operator that translates boresight pointing into detector frame")
    mode = Unicode("I", help="The Stokes weights to generate (I, QU or IQU)")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    hwp_angle = Unicode(None, allow_none=True, help="Observation shared key for HWP angle")
    fp_gamma = Unicode("gamma", allow_none=True, help="Focalplane key for detector gamma offset angle")
    weights = Unicode(defaults.weights, help="Observation detdata key for output weights")
    cal = Unicode(None, allow_none=True, help="The observation key with a dictionary of pointing weight calibration")
    single_precision = Bool(False, help="If True, use 32bit float in output")
    IAU = Bool(False, help="If True, use the IAU convention rather than COSMO")
    skip_quaternions = Bool(True, help="On the accelerator, compute the weights straight from the boresight "
                                       "without materialising detector quaternions (not a reference trait)")

    def _validate_detector_pointing(self, op):
        return _check_detector_pointing(op, ["view", "boresight", "shared_flags", "shared_flag_mask", "det_mask",
                                             "quats", "coord_in", "coord_out"])

    def _validate_mode(self, check):
        if check not in ("I", "QU", "IQU"):
            raise RuntimeError("Invalid mode (must be 'I', 'QU' or 'IQU')")
        return check

    def _exec(self, data, detectors=None, use_accel=None, **kwargs):
        nnz = len(self.mode)
        implementation, use_accel = self.select_kernels(use_accel=use_accel)
        if self.detector_pointing is None:
            raise RuntimeError("The detector_pointing trait must be set")
        if self.single_precision:
            raise NotImplementedError("the compiled kernel writes float64 weights")
        if ("QU" in self.mode) and self.hwp_angle is not None and self.fp_gamma is None:
            raise RuntimeError("If using HWP, you must specify the fp_gamma key")
        quats_name = self.detector_pointing.quats
        view = self.view if self.view is not None else self.detector_pointing.view
        no_quats = self.mode != "QU" and _skip_quaternions(self, data, detectors, use_accel)
        if not no_quats and not _outputs_exist(data, self.weights, detectors, self.detector_pointing.det_mask):
            self.detector_pointing.apply(data, detectors=detectors, use_accel=use_accel)
        for ob in data.obs:
            dets = ob.select_local_detectors(detectors, flagmask=self.detector_pointing.det_mask)
            exists = ob.detdata.ensure(self.weights, sample_shape=(nnz,), dtype=np.float64, detectors=dets,
                                       accel=use_accel, zero_new=not _view_covers_all(ob, view))
            if exists or len(dets) == 0:
                continue
            if no_quats:
                from .. import capi
                from ..accel import accel_device_ptr

                pt = otf_descriptor(ob, dets, None, self)
                wd = ob.detdata[self.weights]
                capi.dev.otf_stokes_weights(pt, wd.indices(dets), accel_device_ptr(wd.buffer), ob.n_local_samples,
                                            ob.intervals[view].data)
                wd.accel_used(True)
                continue
            quat_indx = ob.detdata[quats_name].indices(dets)
            weight_indx = ob.detdata[self.weights].indices(dets)
            focalplane = ob.telescope.focalplane
            det_epsilon = np.array([focalplane[d]["pol_leakage"] for d in dets], dtype=np.float64)
            if self.cal is None:
                cal = np.ones(len(dets), dtype=np.float64)
            else:
                cal = np.array([ob[self.cal][x] for x in dets], np.float64)
            if "QU" in self.mode:
                det_gamma = np.zeros(len(dets), dtype=np.float64)
                if self.hwp_angle is None or self.hwp_angle not in ob.shared:
                    hwp_data = np.zeros(1, dtype=np.float64)
                else:
                    _shared_to(ob, self.hwp_angle, use_accel)
                    hwp_data = ob.shared[self.hwp_angle].data
                    for idet, d in enumerate(dets):
                        det_gamma[idet] = focalplane[d][self.fp_gamma]
                if self.mode == "QU":
                    # (the reference computes IQU into a temporary and copies Q, U out,
                    # stokes_weights.py:251-279; here the kernel writes the two columns directly)
                    from .. import capi

                    capi.stokes_weights_QU(quat_indx, ob.detdata[quats_name].arg(use_accel), weight_indx,
                                           ob.detdata[self.weights].arg(use_accel), hwp_data, ob.intervals[view].data,
                                           det_epsilon, det_gamma, cal, bool(self.IAU), use_accel)
                else:
                    native().stokes_weights_IQU(quat_indx, ob.detdata[quats_name].arg(use_accel), weight_indx,
                                                ob.detdata[self.weights].arg(use_accel), hwp_data,
                                                ob.intervals[view].data, det_epsilon, det_gamma, cal, bool(self.IAU),
                                                use_accel)
            else:
                # the compiled kernel takes a 2-D [n_det, n_samp] buffer (ops_stokes_weights.cpp:417-420)
                wd = ob.detdata[self.weights].arg(use_accel)
                native().stokes_weights_I(weight_indx, wd.reshape(wd.shape[0], wd.shape[1]),
                                          ob.intervals[view].data, cal, use_accel)

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        req = self.detector_pointing.requires()
        req.setdefault("detdata", []).append(self.weights)
        if self.cal is not None:
            req["meta"].append(self.cal)
        if self.hwp_angle is not None:
            req["shared"].append(self.hwp_angle)
        if self.view is not None:
            req["intervals"].append(self.view)
        return req

    def _provides(self):
        prov = self.detector_pointing.provides()
        prov["detdata"].append(self.weights)
        return prov

    def _implementations(self):
        return _IMPLS

    def _supports_accel(self):
        return self.detector_pointing is not None and self.detector_pointing.supports_accel()


class BuildPixelDistribution(Operator):
    """Runs the pixel pointing once to build the ``PixelDistribution`` (which submaps are hit
    locally / by whom) and stores it in ``data[pixel_dist]`` (reference:
    src/toast/ops/pointing.py:18-130).  One pass over all detectors when ``save_pointing`` is
    set, otherwise scratch passes (the reference forces this pass to the host, ``use_accel=False``,
    "a small amount of calculation for a huge data volume"; on the MI355X the pixel kernel runs at
    >100 G samples/s, so it stays on the device when one is in use)."""

    API = Int(0, help="Internal interface version for this operator")
    pixel_dist = Unicode("pixel_dist", help="The Data key where the PixelDist object should be stored")
    pixel_pointing = Instance(klass=Operator, allow_none=True, help="This must be an instance of a pointing operator")
    save_pointing = Bool(False, help="If True, do not clear detector pointing matrices after use")

    def _validate_pixel_pointing(self, pntg):
        if pntg is not None:
            if not isinstance(pntg, Operator):
                raise RuntimeError("pixel_pointing should be an Operator instance")
            for trt in ("pixels", "create_dist", "view"):
                if not pntg.has_trait(trt):
                    raise RuntimeError(f"pixel_pointing operator should have a '{trt}' trait")
        return pntg

    def _exec(self, data, detectors=None, **kwargs):
        from .pipeline import Pipeline, uncached_detector_sets

        if self.pixel_pointing is None:
            raise RuntimeError("You must set the 'pixel_pointing' trait before calling exec()")
        if self.pixel_dist in data:
            raise RuntimeError(f"pixel distribution `{self.pixel_dist}` already exists")
        self.pixel_pointing.create_dist = self.pixel_dist
        pipe = Pipeline(detector_sets=["ALL"] if self.save_pointing else uncached_detector_sets(),
                        operators=[self.pixel_pointing])
        pipe.apply(data, detectors=detectors)
        self.pixel_pointing.create_dist = None

    def _finalize(self, data, **kwargs):
        return

    def _requires(self):
        return self.pixel_pointing.requires()

    def _provides(self):
        prov = {"global": [self.pixel_dist]}
        if self.save_pointing:
            prov["detdata"] = [self.pixel_pointing.pixels]
        return prov
