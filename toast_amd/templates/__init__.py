"""Template amplitudes and the Offset (destriping baseline) template.

Reference: src/toast/templates/amplitudes.py (Amplitudes, AmplitudesMap),
src/toast/templates/template.py (Template), src/toast/templates/offset/offset.py (Offset,
kernels src/toast/_libtoast/template_offset.cpp:16-408).  The noise prior of the Offset
template (offset.py:455-476, 884-1005) is host-only scipy code in the reference
(``NotImplementedError`` on accelerators); here ``use_noise_prior=True`` runs on the device as
well (templates/offset_prior.py, csrc/offset_prior.hip).
"""

import re

import numpy as np

from ..accel import (
    AcceleratorObject,
    accel_data_create,
    accel_data_delete,
    accel_data_present,
    accel_data_reset,
    accel_data_update_device,
    accel_data_update_host,
    accel_device_ptr,
    native,
)
from ..data import defaults
from ..traits import Any, Bool, Float, ImplementationType, Int, TraitConfig, Unicode


class Amplitudes(AcceleratorObject):
    """Local piece of a distributed amplitude vector with flags.  Offset amplitudes are unique
    per process (each detector lives on one process), so ``dot`` = local dot + scalar
    all-reduce (amplitudes.py:523-565)."""

    def __init__(self, comm, n_global, n_local, local_indices=None, local_ranges=None, dtype=np.float64,
                 use_group=False, _full=None):
        """Constructor of the reference (src/toast/templates/amplitudes.py:77-140).  Supported distributions: every
        process holds a full copy (n_local == n_global everywhere: dot products and sync as in the reference) or
        disjoint pieces (the Offset template's case: the sum of n_local equals n_global).  Amplitudes shared between
        some processes through ``local_ranges`` / ``local_indices`` are not part of the offset-template path and
        raise."""
        super().__init__("Amplitudes")
        self._comm = comm
        self._n_global = int(n_global)
        self._n_local = int(n_local)
        if local_indices is not None or local_ranges is not None:
            raise NotImplementedError("Amplitudes with local_ranges / local_indices (values shared between some "
                                      "processes) are not supported: only full copies and disjoint pieces")
        self._local_indices = None
        self._local_ranges = None
        self._use_group = bool(use_group)
        world = None if comm is None else (comm.comm_group if use_group else comm.comm_world)
        self._full = False
        if _full is not None:
            self._full = bool(_full)      # copy of an existing distribution: no collective
        elif world is not None:
            total = comm.allreduce_scalar(self._n_local, op="sum")
            size = comm.group_size if use_group else comm.world_size
            if total == size * self._n_global:
                self._full = True
            elif total != self._n_global:
                raise RuntimeError("Total amplitudes on all processes does not equal n_global")
        elif self._n_local != self._n_global:
            raise RuntimeError("Total amplitudes on all processes does not equal n_global")
        self._local = np.zeros(self._n_local, dtype=dtype)
        self.local_flags = np.zeros(self._n_local, dtype=np.uint8)
        # nobody has seen the host values yet: they are zeros, and the first "upload" is a fill on the device (the
        # right-hand side's 3.7 M fresh amplitudes at cfg-3: 2.2 ms of copy from pageable memory with the device idle)
        self._pristine = True

    # Lazy host coherence (like PixelData.data): the solver leaves its vectors resident and device-current; the first HOST
    # access of ``local`` copies the values back and makes the host the current side again.
    @property
    def local(self):
        """The local amplitudes on the host (amplitudes.py:291-300), brought up to date first."""
        if self._accel_used:
            self.accel_update_host()
        self._pristine = False        # (the caller may write through the array it gets)
        return self._local

    @local.setter
    def local(self, value):
        self._pristine = False
        self._local = value

    @property
    def buffer(self):
        """The host array WITHOUT synchronisation: its address is the accelerator key and its contents may be stale.
        For device-side code paths only."""
        return self._local

    def arg(self, use_accel):
        """Array to hand to a kernel call: the key array when the kernel runs on the registered device memory, the
        synchronised host contents when the call is host-staged."""
        return self._local if use_accel else self.local

    n_global = property(lambda self: self._n_global)
    n_local = property(lambda self: self._n_local)
    comm = property(lambda self: self._comm)
    local_indices = property(lambda self: self._local_indices)
    local_ranges = property(lambda self: self._local_ranges)
    use_group = property(lambda self: self._use_group)

    @property
    def n_local_flagged(self):
        """Number of flagged local amplitudes (amplitudes.py:335-341)."""
        if self._n_local == 0:
            return 0
        if self.accel_in_use():
            accel_data_update_host(self.local_flags, self._accel_name + "_flags")
        return int(np.count_nonzero(self.local_flags))

    def reset_flags(self):
        """Clear all flags, on the host and on the device copy (amplitudes.py:283-290)."""
        if self._n_local == 0:
            return
        self.local_flags[:] = 0
        if self.accel_exists():
            accel_data_update_device(self.local_flags, self._accel_name + "_flags")

    def _host(self):
        if self.accel_in_use():
            self.accel_update_host()

    def accel_resident(self, name=None):
        """Make the device copy current and keep working there: later arithmetic (``+=``,
        ``axpby``, ``dot`` ...) runs in device kernels (toast_hip_vec_*_dev) until
        ``accel_update_host``."""
        if self._n_local == 0:
            return self
        if not self.accel_exists():
            self.accel_create(name)
        if not self.accel_in_use():
            self.accel_update_device()
        return self

    def _device_pair(self, other):
        """True when the operation should run on the device: either operand's device copy is
        the current one (the other is then made resident as well)."""
        if not isinstance(other, Amplitudes):
            return self.accel_in_use()
        if not (self.accel_in_use() or other.accel_in_use()):
            return False
        self.accel_resident()
        other.accel_resident()
        return True

    def _dptr(self):
        return accel_device_ptr(self._local)

    def duplicate(self):
        ret = Amplitudes(self._comm, self._n_global, self._n_local, dtype=self._local.dtype, use_group=self._use_group,
                         _full=self._full)
        if self.accel_in_use():
            ret.local_flags[:] = self.local_flags
            ret.accel_create(self._accel_name + "_dup")
            ret.accel_used(True)
            ret.copy_from(self)
        else:
            ret.local[:] = self.local
            ret.local_flags[:] = self.local_flags
        return ret

    def reset(self):
        if self.accel_in_use():
            self.accel_reset()
            return
        self._local[:] = 0
        if self.accel_exists():
            self.accel_reset()

    def clear(self):
        if self.accel_exists():
            self.accel_delete()

    def axpby(self, a, other, b=1.0):
        """self = a * other + b * self (the PCG updates without a temporary)."""
        if self._n_local == 0:
            return self
        if self._device_pair(other):
            from .. import capi

            capi.dev.vec_axpby(self._n_local, float(a), other._dptr(), float(b), self._dptr())
        elif b == 1.0:
            self.local += a * other.local
        else:
            self.local *= b
            self.local += a * other.local
        return self

    def copy_from(self, other):
        """self.local[:] = other.local on whichever side is current."""
        if self._n_local == 0:
            return self
        if self._device_pair(other):
            from .. import capi

            capi.dev.vec_axpby(self._n_local, 1.0, other._dptr(), 0.0, self._dptr())
        else:
            self.local[:] = other.local
        return self

    def __iadd__(self, other):
        if isinstance(other, Amplitudes):
            return self.axpby(1.0, other)
        self._host()
        self.local += other
        return self

    def __isub__(self, other):
        if isinstance(other, Amplitudes):
            return self.axpby(-1.0, other)
        self._host()
        self.local -= other
        return self

    def __imul__(self, other):
        if isinstance(other, Amplitudes):
            self._host()
            other._host()
            self.local *= other.local
        elif self.accel_in_use():
            from .. import capi

            capi.dev.vec_axpby(self._n_local, 0.0, self._dptr(), float(other), self._dptr())
        else:
            self.local *= other
        return self

    def dot(self, other):
        if self._n_local > 0 and self._device_pair(other):
            from .. import capi

            val = capi.dev.vec_dot(self._n_local, self._dptr(), other._dptr(), accel_device_ptr(self.local_flags),
                                   accel_device_ptr(other.local_flags))
        else:
            good = np.logical_and(self.local_flags == 0, other.local_flags == 0)
            val = float(np.dot(np.where(good, self.local, 0.0), other.local))
        # every process holds the full set: no reduction (amplitudes.py:545-554); disjoint pieces: sum of the local dots
        if self._comm is not None and self._comm.comm_world is not None and not self._full:
            val = self._comm.allreduce_scalar(val, op="sum")
        return val

    def sync(self, comm_bytes=10000000):
        """Sum over processes where amplitudes are replicated (amplitudes.py:357-470): full copies are all-reduced
        with flagged values contributing zero; disjoint pieces (the Offset template) need no communication."""
        if self._comm is None or self._comm.comm_world is None or self._n_global == 0 or not self._full:
            return
        self._host()
        send = np.where(self.local_flags != 0, 0, self.local)
        self._comm.allreduce_array_(send)
        self.local[:] = send
        if self.accel_exists():
            self.accel_update_device()

    # accelerator protocol: values and flags are two registered buffers
    def _accel_exists(self):
        return self._n_local > 0 and accel_data_present(self._local, self._accel_name)

    def _accel_create(self, zero_out=False):
        # (amplitudes are what the M^T kernels scatter into: KIND_SCATTER)
        accel_data_create(self._local, self._accel_name, zero_out=zero_out, owner=self, kind=2)
        accel_data_create(self.local_flags, self._accel_name + "_flags", owner=self)
        accel_data_update_device(self.local_flags, self._accel_name + "_flags")

    def _accel_update_device(self):
        if getattr(self, "_pristine", False):
            accel_data_reset(self._local, self._accel_name)          # zeros that never left the constructor
        else:
            accel_data_update_device(self._local, self._accel_name)
        accel_data_update_device(self.local_flags, self._accel_name + "_flags")

    def _accel_update_host(self):
        self._pristine = False
        accel_data_update_host(self._local, self._accel_name)

    def _accel_delete(self):
        accel_data_delete(self._local, self._accel_name)
        accel_data_delete(self.local_flags, self._accel_name + "_flags")

    def _accel_reset(self):
        accel_data_reset(self._local, self._accel_name)


class AmplitudesMap(dict):
    """name -> Amplitudes with vector arithmetic (amplitudes.py AmplitudesMap)."""

    def duplicate(self):
        ret = AmplitudesMap()
        for k, v in self.items():
            ret[k] = v.duplicate()
        return ret

    def reset(self):
        for v in self.values():
            v.reset()

    def reset_flags(self):
        for v in self.values():
            v.reset_flags()

    def clear(self):
        for v in self.values():
            v.clear()
        super().clear()

    def dot(self, other):
        return sum(v.dot(other[k]) for k, v in self.items())

    def _binary(self, other, fn):
        for k, v in self.items():
            fn(v, other[k] if isinstance(other, AmplitudesMap) else other)
        return self

    def __iadd__(self, other):
        return self._binary(other, lambda a, b: a.__iadd__(b))

    def __isub__(self, other):
        return self._binary(other, lambda a, b: a.__isub__(b))

    def __imul__(self, other):
        return self._binary(other, lambda a, b: a.__imul__(b))

    def axpby(self, a, other, b=1.0):
        for k, v in self.items():
            v.axpby(a, other[k], b)
        return self

    def copy_from(self, other):
        for k, v in self.items():
            v.copy_from(other[k])
        return self

    def accel_resident(self, name):
        for k, v in self.items():
            v.accel_resident(f"{name}_{k}")
        return self

    def accel_exists(self):
        return all(v.accel_exists() for v in self.values()) and len(self) > 0

    def accel_in_use(self):
        return any(v.accel_in_use() for v in self.values())

    def accel_create(self, name, zero_out=False):
        for k, v in self.items():
            if not v.accel_exists():
                v.accel_create(f"{name}_{k}", zero_out=zero_out)

    def accel_used(self, state):
        for v in self.values():
            v.accel_used(state)

    def accel_update_device(self):
        for v in self.values():
            if v.accel_exists() and not v.accel_in_use():
                v.accel_update_device()

    def accel_update_host(self):
        for v in self.values():
            if v.accel_in_use():
                v.accel_update_host()

    def accel_delete(self):
        for v in self.values():
            if v.accel_exists():
                v.accel_delete()


class Template(TraitConfig):
    """Base class of timestream templates (templates/template.py)."""

    data = Any(None, help="This must be an instance of a Data class (or None)")
    view = Unicode(None, allow_none=True, help="Use this view of the data in all observations")
    det_data = Unicode(defaults.det_data, allow_none=True, help="Observation detdata key for the timestream data")
    det_data_units = Unicode(defaults.det_data_units, allow_none=True, help="Desired units of detector data")
    det_mask = Int(defaults.det_mask_invalid, help="Bit mask value for per-detector flagging")
    det_flags = Unicode(defaults.det_flags, allow_none=True, help="Observation detdata key for solver flags to use")
    det_flag_mask = Int(defaults.det_mask_nonscience, help="Bit mask value for solver flags")
    pattern = Unicode(None, allow_none=True, help="Regex pattern to match against detector names. "
                                                  "Only these are projected.")

    def _observe_data(self, change):
        if change["new"] is not None:
            self._initialize(change["new"])

    def initialize(self, new_data):
        self.data = new_data

    def _check_enabled(self):
        if self.data is None:
            raise RuntimeError("You must set the data trait before calling template methods")
        return self.enabled

    def detectors(self):
        return self._detectors() if self._check_enabled() else []

    def zeros(self):
        return self._zeros() if self._check_enabled() else None

    def add_to_signal(self, detector, amplitudes, **kwargs):
        if self._check_enabled():
            self._add_to_signal(detector, amplitudes, **kwargs)

    def project_signal(self, detector, amplitudes, **kwargs):
        if self._check_enabled():
            self._project_signal(detector, amplitudes, **kwargs)

    def add_prior(self, amplitudes_in, amplitudes_out, **kwargs):
        if self._check_enabled():
            self._add_prior(amplitudes_in, amplitudes_out, **kwargs)

    def apply_precond(self, amplitudes_in, amplitudes_out, **kwargs):
        if self._check_enabled():
            self._apply_precond(amplitudes_in, amplitudes_out, **kwargs)


class Offset(Template):
    """One amplitude per detector per ``step_time`` of every view (baseline offsets)."""

    step_time = Float(10000.0, help="Time per baseline step [s]")
    times = Unicode(defaults.times, help="Observation shared key for timestamps")
    noise_model = Unicode(None, allow_none=True, help="Observation key containing the optional noise model")
    good_fraction = Float(0.5, help="Fraction of unflagged samples needed to keep a given offset amplitude")
    use_noise_prior = Bool(False, help="Use detector PSDs to build the noise prior and preconditioner")
    precond_width = Int(20, help="Preconditioner width in terms of offsets / baselines")

    def __init__(self, **kwargs):
        self._flag_cache = {}
        super().__init__(**kwargs)

    # The offset variances are computed on the device (``_init_variances_device``) and read there by the preconditioner:
    # the host copy is fetched when somebody asks for it.
    @property
    def _offsetvar(self):
        if getattr(self, "_offsetvar_stale", False):
            self._offsetvar_stale = False
            accel_data_update_host(self._offsetvar_buf, f"{self.name}_offsetvar")
        return self._offsetvar_buf

    @_offsetvar.setter
    def _offsetvar(self, value):
        self._drop_offsetvar_device()
        self._offsetvar_buf = value
        self._offsetvar_stale = False

    def _drop_offsetvar_device(self):
        if getattr(self, "_offsetvar_on_dev", False):
            accel_data_delete(self._offsetvar_buf, f"{self.name}_offsetvar")
        self._offsetvar_on_dev = False
        self._offsetvar_stale = False

    def _step_length(self, stime, rate):
        return int(np.rint(stime * rate))   # offset.py:723-724 (round half to even)

    def _initialize(self, new_data):
        if self.use_noise_prior and self.noise_model is None:
            raise RuntimeError("cannot use noise prior without specifying noise_model")
        # with a noise prior the baselines run through the whole observation and the view only
        # flags samples (offset.py:135-140)
        self._bounds_view = None if self.use_noise_prior else self.view
        if getattr(self, "_prior", None) is not None:
            self._prior.clear()
        self._prior = None
        self._obs_views, self._obs_view_flags, self._obs_rate, self._obs_dets = {}, {}, {}, {}
        all_dets = {}
        for iob, ob in enumerate(new_data.obs):
            rate = ob.telescope.focalplane.sample_rate
            if self.times in ob.shared and ob.shared[self.times].data.size > 1:
                t = ob.shared[self.times].data
                # rate_from_times; a median over all time stamps (8 ms per 720 000), remembered per time-stamp buffer
                memo = ob.__dict__.setdefault("_rate_from_times", {})
                key = (t.ctypes.data, t.size, float(t[0]), float(t[-1]))
                if key not in memo:
                    memo.clear()
                    memo[key] = 1.0 / np.median(np.diff(t))
                rate = memo[key]
            self._obs_rate[iob] = rate
            step_length = self._step_length(self.step_time, rate)
            views = []
            for vw in ob.intervals[self._bounds_view]:
                view_len = vw.last - vw.first
                n = view_len // step_length
                if n * step_length < view_len:
                    n += 1
                views.append(n)
            self._obs_views[iob] = np.array(views, dtype=np.int64)
            vf = np.ones(ob.n_local_samples, dtype=np.uint8)
            for vw in ob.intervals[self.view]:
                vf[vw.first:vw.last] = 0
            self._obs_view_flags[iob] = vf
            self._obs_dets[iob] = set()
            det_pat = re.compile(self.pattern) if self.pattern is not None else None
            # (sets: `d in list` over 1024 detectors twice per detector was 5 ms of this function)
            have_data = set(ob.detdata[self.det_data].detectors) if self.det_data in ob.detdata else None
            have_flags = set(ob.detdata[self.det_flags].detectors) \
                if (self.det_flags is not None and self.det_flags in ob.detdata) else None
            for d in ob.select_local_detectors(flagmask=self.det_mask):
                if have_data is not None and d not in have_data:
                    continue
                if det_pat is not None and det_pat.match(d) is None:
                    continue  # offset.py:226-236
                if have_flags is not None and d not in have_flags:
                    continue  # no solver flags for it: not part of this (split) run
                self._obs_dets[iob].add(d)
                all_dets.setdefault(d, None)
        self._all_dets = list(all_dets.keys())
        self._det_start = {}
        view_totals = {iob: int(np.sum(v)) for iob, v in self._obs_views.items()}
        offset = 0
        for det in self._all_dets:
            self._det_start[det] = offset
            for iob, ob in enumerate(new_data.obs):
                if det in self._obs_dets[iob]:
                    offset += view_totals[iob]
        self._n_local = offset
        comm = new_data.comm
        self._n_global = self._n_local
        if comm.comm_world is not None:
            self._n_global = int(comm.allreduce_scalar(self._n_local, op="sum"))
        self._amp_flags = np.zeros(self._n_local, dtype=np.uint8)
        self._offsetvar = np.zeros(self._n_local, dtype=np.float64)
        self._amp_offset_cache = {}
        # offset variance / flags: offset.py:262-343
        from ..accel import accel_enabled

        if accel_enabled() and self._n_local > 0 and self._bounds_view == self.view:
            self._init_variances_device(new_data)
        else:
            # (also when the baselines do not follow the view: the samples outside the view count
            # as flagged, which the host form takes from the view flags)
            self._init_variances_host(new_data)
        self._flag_cache = {}
        if self.use_noise_prior and self._n_local > 0:
            # the prior builds its filters from the variances on the host and registers the array under its own name
            _ = self._offsetvar
            self._drop_offsetvar_device()
            self._init_noise_prior(new_data)

    def _init_noise_prior(self, new_data):
        """Filters and preconditioners of every (detector, observation, view) segment
        (offset.py:200-223, 345-566), built once on the host; see templates/offset_prior.py."""
        from .offset_prior import OffsetPrior, prior_frequencies

        freq = {}
        for iob, ob in enumerate(new_data.obs):
            t = ob.shared[self.times].data
            obstime = float(t[-1] - t[0])
            if obstime / self.step_time < 1.0:
                # the reference disables the prior for such an observation and then fails on the
                # missing frequency grid (offset.py:212-218, 365-371)
                raise RuntimeError(f"obs {ob.name} has only one offset amplitude: cannot use the noise prior")
            freq[iob] = prior_frequencies(obstime, self.step_time, self._obs_rate[iob])
        segments = []
        offset = 0
        for det in self._all_dets:
            for iob, ob in enumerate(new_data.obs):
                if det not in self._obs_dets[iob]:
                    continue
                noise = ob[self.noise_model]
                psdfreq = np.ascontiguousarray(noise.freq(det), dtype=np.float64)
                psd = np.ascontiguousarray(noise.psd(det), dtype=np.float64)
                detnoise = float(noise.detector_weight(det))
                for n_amp_view in self._obs_views[iob]:
                    segments.append(dict(first=offset, n_amp=int(n_amp_view), freq=freq[iob], psdfreq=psdfreq,
                                         psd=psd, detnoise=detnoise))
                    offset += int(n_amp_view)
        self._prior = OffsetPrior(self.name, self.precond_width).build(segments, self._offsetvar, self.step_time)

    def _init_variances_device(self, new_data):
        """Per-amplitude counts of flagged samples from one kernel over the resident flags
        (toast_hip_offset_count_flagged_dev), flags and variances of all amplitudes from a second one
        (toast_hip_offset_variance_dev); only the two finished vectors come back to the host."""
        from .. import capi
        from ..accel import accel_data_create, accel_data_delete, accel_data_update_device, accel_data_update_host

        # per-amplitude counts of flagged samples: device scratch, never on the host
        bad_bytes = 8 * self._n_local
        bad_ptr = capi.device_malloc(bad_bytes)
        capi.dev.memset(bad_ptr, 0, bad_bytes)
        # an amplitude that no (detector, observation) block claims stays cut, like a baseline without samples
        self._amp_flags[:] = 1
        flag_name, var_name = f"{self.name}_init_flags", f"{self.name}_offsetvar"
        accel_data_create(self._amp_flags, flag_name)
        accel_data_update_device(self._amp_flags, flag_name)
        # (the variances stay on the device under the name the preconditioner looks for; host copy on demand)
        accel_data_create(self._offsetvar_buf, var_name, zero_out=True, owner=self)
        self._offsetvar_on_dev = True
        try:
            for iob, ob in enumerate(new_data.obs):
                dets = [d for d in self._all_dets if d in self._obs_dets[iob]]
                if len(dets) == 0:
                    continue
                step_length = self._step_length(self.step_time, self._obs_rate[iob])
                # lengths of the baselines of one detector of this observation, view after view
                lens = []
                for ivw, vw in enumerate(ob.intervals[self._bounds_view]):
                    n_amp_view = int(self._obs_views[iob][ivw])
                    if n_amp_view == 0:
                        continue
                    a = np.full(n_amp_view, step_length, dtype=np.int64)
                    a[-1] = (vw.last - vw.first) - (n_amp_view - 1) * step_length
                    lens.append(a)
                lens = np.concatenate(lens) if lens else np.zeros(0, dtype=np.int64)
                offs = self.det_amp_offsets(iob, dets)
                w = np.ones(len(dets), dtype=np.float64)
                if self.noise_model is not None:
                    w = np.array([ob[self.noise_model].detector_weight(d) for d in dets], dtype=np.float64)
                if self.det_flags is not None:
                    fd = ob.detdata[self.det_flags]
                    if not fd.accel_in_use():
                        if not fd.accel_exists():
                            fd.accel_create(self.det_flags)
                        fd.accel_update_device()
                    capi.dev.offset_count_flagged(step_length, offs, self._obs_views[iob], bad_ptr,
                                                  fd.indices(dets), accel_device_ptr(fd.buffer), self.det_flag_mask,
                                                  ob.n_local_samples, ob.intervals[self._bounds_view].data)
                capi.dev.offset_variance(offs, w, lens, bad_ptr, self.good_fraction,
                                         accel_device_ptr(self._amp_flags), accel_device_ptr(self._offsetvar_buf))
            accel_data_update_host(self._amp_flags, flag_name)
            self._offsetvar_stale = True
        except BaseException:
            self._drop_offsetvar_device()
            raise
        finally:
            accel_data_delete(self._amp_flags, flag_name)
            capi.device_release(bad_ptr, bad_bytes)

    def _init_variances_host(self, new_data):
        offset = 0
        for det in self._all_dets:
            for iob, ob in enumerate(new_data.obs):
                if det not in self._obs_dets[iob]:
                    continue
                detnoise = 1.0
                if self.noise_model is not None:
                    detnoise = ob[self.noise_model].detector_weight(det)
                step_length = self._step_length(self.step_time, self._obs_rate[iob])
                for ivw, vw in enumerate(ob.intervals[self._bounds_view]):
                    n_amp_view = int(self._obs_views[iob][ivw])
                    view_samples = vw.last - vw.first
                    if detnoise <= 0:
                        self._amp_flags[offset:offset + n_amp_view] = 1
                    else:
                        flags = np.array(self._obs_view_flags[iob][vw.first:vw.last], dtype=np.uint8)
                        if self.det_flags is not None:
                            flags |= ob.detdata[self.det_flags][det, vw.first:vw.last] & self.det_flag_mask
                        # per-baseline count of good samples, vectorised (offset.py:318-343)
                        starts = np.arange(n_amp_view, dtype=np.int64) * step_length
                        bad = np.add.reduceat((flags != 0).astype(np.int64), starts)
                        amplen = np.full(n_amp_view, step_length, dtype=np.int64)
                        amplen[-1] = view_samples - starts[-1]
                        n_good = amplen - bad
                        cut = (n_good / amplen) <= self.good_fraction
                        sl = slice(offset, offset + n_amp_view)
                        self._amp_flags[sl][cut] = 1
                        with np.errstate(divide="ignore"):
                            var = 1.0 / (detnoise * n_good)
                        self._offsetvar[sl] = np.where(cut, 0.0, var)
                    offset += n_amp_view
    def _detectors(self):
        return self._all_dets

    def _zeros(self):
        z = Amplitudes(self.data.comm, self._n_global, self._n_local)
        z.local_flags[:] = self._amp_flags
        return z

    def _supports_accel(self):
        return True

    def supports_accel(self):
        return self._supports_accel()

    def _implementations(self):
        return [ImplementationType.DEFAULT, ImplementationType.COMPILED]

    def _amps_to(self, amplitudes, use_accel):
        if use_accel:
            if not amplitudes.accel_exists():
                amplitudes.accel_create(f"{self.name}_amps")
                amplitudes.accel_update_device()
            elif not amplitudes.accel_in_use():
                amplitudes.accel_update_device()
        elif amplitudes.accel_in_use():
            amplitudes.accel_update_host()

    def det_amp_offsets(self, iob, dets):
        """First amplitude of each detector's block for observation ``iob`` (the running
        ``amp_offset`` of offset.py:727-760), cached."""
        key = (iob, tuple(dets))
        cache = self.__dict__.setdefault("_amp_offset_cache", {})
        if key not in cache:
            out = []
            for d in dets:
                off = self._det_start[d]
                for job in range(iob):
                    if d in self._obs_dets[job]:
                        off += int(np.sum(self._obs_views[job]))
                out.append(off)
            cache[key] = np.array(out, dtype=np.int64)
        return cache[key]

    def add_to_signal_multi(self, detectors, amplitudes, **kwargs):
        """All detectors in one launch per observation (device-resident buffers only)."""
        from .. import capi
        from ..accel import accel_device_ptr

        if not self._check_enabled():
            return
        self._amps_to(amplitudes, True)
        for iob, ob in enumerate(self.data.obs):
            dets = [d for d in detectors if d in self._obs_dets[iob]]
            if len(dets) == 0:
                continue
            dd = ob.detdata[self.det_data]
            capi.dev.offset_add_to_signal_multi(
                self._step_length(self.step_time, self._obs_rate[iob]), self.det_amp_offsets(iob, dets),
                self._obs_views[iob], accel_device_ptr(amplitudes.buffer), accel_device_ptr(amplitudes.local_flags),
                dd.indices(dets), accel_device_ptr(dd.buffer), ob.n_local_samples, ob.intervals[self._bounds_view].data)

    def project_signal_multi(self, detectors, amplitudes, **kwargs):
        from .. import capi
        from ..accel import accel_device_ptr

        if not self._check_enabled():
            return
        self._amps_to(amplitudes, True)
        for iob, ob in enumerate(self.data.obs):
            dets = [d for d in detectors if d in self._obs_dets[iob]]
            if len(dets) == 0:
                continue
            dd = ob.detdata[self.det_data]
            if self.det_flags is not None:
                f_idx = ob.detdata[self.det_flags].indices(dets)
                f_ptr = accel_device_ptr(self._solver_flags(iob, ob, True))
            else:
                f_idx, f_ptr = None, 0
            capi.dev.offset_project_signal_multi(
                dd.indices(dets), accel_device_ptr(dd.buffer), f_idx, f_ptr, self.det_flag_mask,
                self._step_length(self.step_time, self._obs_rate[iob]), self.det_amp_offsets(iob, dets),
                self._obs_views[iob], accel_device_ptr(amplitudes.buffer), accel_device_ptr(amplitudes.local_flags),
                ob.n_local_samples, ob.intervals[self._bounds_view].data)

    def _add_to_signal(self, detector, amplitudes, use_accel=None, **kwargs):
        if detector not in self._all_dets:
            return
        use_accel = bool(use_accel)
        self._amps_to(amplitudes, use_accel)
        amp_offset = self._det_start[detector]
        for iob, ob in enumerate(self.data.obs):
            if detector not in self._obs_dets[iob]:
                continue
            det_indx = ob.detdata[self.det_data].indices([detector])
            step_length = self._step_length(self.step_time, self._obs_rate[iob])
            n_amp_views = self._obs_views[iob]
            native().template_offset_add_to_signal(step_length, amp_offset, n_amp_views, amplitudes.arg(use_accel),
                                                   amplitudes.local_flags, int(det_indx[0]),
                                                   ob.detdata[self.det_data].arg(use_accel), ob.intervals[self._bounds_view].data,
                                                   use_accel)
            amp_offset += int(np.sum(n_amp_views))

    def _solver_flags(self, iob, ob, use_accel):
        """det_flags | view flags, cached per observation (offset.py:834-843 builds this copy on
        every call); registered on the device when needed."""
        key = (iob, use_accel)
        fd = ob.detdata[self.det_flags]
        if use_accel and not self._obs_view_flags[iob].any():
            # no sample lies outside the view: the solver flags ARE the detector flags, and they
            # are used where they live (no 1 B/det-sample round trip through the host)
            if not fd.accel_in_use():
                if not fd.accel_exists():
                    fd.accel_create(self.det_flags)
                fd.accel_update_device()
            return fd.buffer
        if key not in self._flag_cache and use_accel and (self.det_flag_mask & 1) and fd.accel_in_use():
            # device-current flags and a view: (flags & mask) != 0 inside the view, 1 outside, built
            # on the device (toast_hip_combine_flags_dev); equivalent under `& det_flag_mask` to
            # the reference's det_flags | mask * view_flags as long as bit 0 is in the mask
            from .. import capi

            buf = fd.buffer
            rows = np.arange(buf.shape[0], dtype=np.int32)
            flag_data = np.empty_like(buf)   # host key only: the contents live on the device
            accel_data_create(flag_data, f"{self.name}_solver_flags", owner=self)
            capi.dev.combine_flags(accel_device_ptr(flag_data), rows, accel_device_ptr(buf), ob.n_local_samples, rows,
                                   self.det_flag_mask, 0, 0, 0, ob.n_local_samples, ob.intervals[self.view].data,
                                   n_out_rows=buf.shape[0], outside_value=1)
            self._flag_cache[key] = flag_data
        if key not in self._flag_cache:
            if fd.accel_in_use():
                fd.accel_update_host()
                fd.accel_used(True)
            flag_data = np.copy(fd.data)
            flag_data |= (self.det_flag_mask * self._obs_view_flags[iob]).astype(np.uint8)
            if use_accel:
                accel_data_create(flag_data, f"{self.name}_solver_flags", owner=self)
                accel_data_update_device(flag_data, f"{self.name}_solver_flags")
            self._flag_cache[key] = flag_data
        return self._flag_cache[key]

    def _project_signal(self, detector, amplitudes, use_accel=None, **kwargs):
        if detector not in self._all_dets:
            return
        use_accel = bool(use_accel)
        self._amps_to(amplitudes, use_accel)
        amp_offset = self._det_start[detector]
        for iob, ob in enumerate(self.data.obs):
            if detector not in self._obs_dets[iob]:
                continue
            det_indx = ob.detdata[self.det_data].indices([detector])
            if self.det_flags is not None:
                flag_indx = int(ob.detdata[self.det_flags].indices([detector])[0])
                flag_data = self._solver_flags(iob, ob, use_accel)
            else:
                flag_indx = -1
                flag_data = np.zeros((1, 1), dtype=np.uint8)
            step_length = self._step_length(self.step_time, self._obs_rate[iob])
            n_amp_views = self._obs_views[iob]
            native().template_offset_project_signal(int(det_indx[0]), ob.detdata[self.det_data].arg(use_accel), flag_indx,
                                                    flag_data, self.det_flag_mask, step_length, amp_offset,
                                                    n_amp_views, amplitudes.arg(use_accel), amplitudes.local_flags,
                                                    ob.intervals[self._bounds_view].data, use_accel)
            amp_offset += int(np.sum(n_amp_views))

    def _on_device(self, amplitudes_in, amplitudes_out, fn):
        """Run ``fn`` with both vectors resident; hand the result back to the host when the
        caller's vectors were host-current (the PCG keeps them resident and skips this)."""
        stay = amplitudes_in.accel_in_use() or amplitudes_out.accel_in_use()
        in_was = amplitudes_in.accel_in_use()
        amplitudes_in.accel_resident(f"{self.name}_amps_in")
        amplitudes_out.accel_resident(f"{self.name}_amps_out")
        fn()
        if not stay:
            native().accel_synchronize()
            amplitudes_out.accel_update_host()
        if not in_was:
            amplitudes_in.accel_used(False)  # never modified on the device

    def _add_prior(self, amplitudes_in, amplitudes_out, use_accel=None, **kwargs):
        if not self.use_noise_prior or self._n_local == 0:
            # no noise prior: nothing to accumulate (offset.py:884-893)
            return
        self._on_device(amplitudes_in, amplitudes_out, lambda: self._prior.add_prior(amplitudes_in, amplitudes_out))

    def _apply_precond(self, amplitudes_in, amplitudes_out, use_accel=None, **kwargs):
        if self._n_local == 0:
            return
        if self.use_noise_prior:
            # the left-hand side holds the inverse baseline covariance, so does the preconditioner
            # (offset.py:963-1005)
            self._on_device(amplitudes_in, amplitudes_out,
                            lambda: self._prior.apply_precond(amplitudes_in, amplitudes_out))
            return
        # diagonal preconditioner (offset.py:1007-1028 -> template_offset_apply_diag_precond)
        if amplitudes_in.accel_in_use() or amplitudes_out.accel_in_use():
            # device-resident PCG vectors: the variances are uploaded once per template
            amplitudes_in.accel_resident()
            amplitudes_out.accel_resident()
            self._offsetvar_to_device()
            native().template_offset_apply_diag_precond(self._offsetvar_buf, amplitudes_in.buffer,
                                                        amplitudes_in.local_flags, amplitudes_out.buffer, True)
            return
        native().template_offset_apply_diag_precond(self._offsetvar, amplitudes_in.local, amplitudes_in.local_flags,
                                                    amplitudes_out.local, False)

    def precond_diag_device(self):
        """Device pointer of the diagonal preconditioner (the offset variances), or None when this template's
        preconditioner is not a diagonal (noise prior: banded).  Lets the solver fuse ``apply_precond`` with the dot
        product that reads its output (toast_hip_pcg_precond_diag_dot_dev)."""
        if self.use_noise_prior or not self._check_enabled() or self._n_local == 0:
            return None
        self._offsetvar_to_device()
        from ..accel import accel_device_ptr

        return accel_device_ptr(self._offsetvar_buf)

    def _offsetvar_to_device(self):
        if not getattr(self, "_offsetvar_on_dev", False):
            accel_data_create(self._offsetvar_buf, f"{self.name}_offsetvar", owner=self)
            accel_data_update_device(self._offsetvar_buf, f"{self.name}_offsetvar")
            self._offsetvar_on_dev = True

    def clear(self):
        if getattr(self, "_prior", None) is not None:
            self._prior.clear()
        if getattr(self, "_offsetvar_buf", None) is not None:
            _ = self._offsetvar          # (somebody may still read the variances on the host)
            self._drop_offsetvar_device()
        for (iob, on_dev), buf in self._flag_cache.items():
            if on_dev:
                accel_data_delete(buf, f"{self.name}_solver_flags")
        self._flag_cache = {}
