"""ctypes view of the C ABI in ``include/toast_hip.h`` (``toast_amd/libtoast_hip.so``).

Two groups of callables:

* ``pixels_healpix(...)``, ``ops_scan_map_float64(...)``, ``build_noise_weighted(...)`` ... take
  NumPy arrays in exactly the argument order of the reference's ``toast._libtoast`` bindings
  (including the trailing ``use_accel``) and call the host-pointer level
  entry points.  Buffer validation mirrors ``extract_buffer``
  (reference: src/toast/_libtoast/common.hpp:32-124).
* ``dev.<kernel>(...)`` take raw device pointers (``int``; e.g. ``tensor.data_ptr()``) for the
  large arrays and NumPy arrays for the small per-call ones.

There is no CPU fallback: if the library is missing or no gfx950 device is usable, calls raise.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TOAST_HIP_LIBRARY: load another build of the same library (kernel experiments, profiles/)
LIB_PATH = os.environ.get("TOAST_HIP_LIBRARY") or os.path.join(_HERE, "libtoast_hip.so")

interval_dtype = np.dtype(
    {
        "names": ["start", "stop", "first", "last"],
        "formats": ["d", "d", "q", "q"],
        "offsets": [0, 8, 16, 24],
    }
)

MAP_F64, MAP_F32, MAP_I64, MAP_I32 = 0, 1, 2, 3
_MAP_CODES = {
    np.dtype(np.float64): MAP_F64,
    np.dtype(np.float32): MAP_F32,
    np.dtype(np.int64): MAP_I64,
    np.dtype(np.int32): MAP_I32,
}

_lib = None


class CallPlan:
    """A recorded sequence of C-ABI calls with their fully converted arguments: the host-side
    analogue of a captured graph.  ``capture()`` records instead of executing; ``replay()``
    issues the calls.  The small host arrays the calls point at are kept alive with the plan
    (the library copies them into its cached parameter blocks at every call)."""

    def __init__(self):
        self.calls = []
        self.keep = []

    def replay(self):
        for fn, args in self.calls:
            if fn(*args) != 0:
                raise RuntimeError(_lib.toast_hip_last_error().decode())


class _Recorder:
    def __init__(self, plan):
        self._plan = plan

    def __getattr__(self, name):
        real = getattr(_lib, name)
        plan = self._plan

        def record(*args):
            plan.calls.append((real, args))
            return 0

        return record


_capture = None


class capture:
    """``with capi.capture() as plan:`` -- every ``capi.dev`` / C-ABI call made inside is
    recorded into ``plan`` and NOT executed."""

    def __enter__(self):
        global _capture
        lib()
        if _capture is not None:
            raise RuntimeError("capi.capture() does not nest")
        self.plan = CallPlan()
        _capture = _Recorder(self.plan)
        return self.plan

    def __exit__(self, *exc):
        global _capture
        _capture = None
        return False


def lib():
    """Load libtoast_hip.so (raises if it has not been built: no fallback)."""
    global _lib
    if _capture is not None and _lib is not None:
        return _capture
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m toast_amd.build` "
                "(the HIP library is the only implementation of this path)"
            )
        # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64.so (SONAME
        # libamdhip64.so.7) and look it up by file name, so if /opt/rocm's copy were loaded first
        # torch would load a second runtime and lose the GPU.  Importing torch first makes the
        # loader resolve our NEEDED libamdhip64.so.7 / librocfft.so.0 to the already loaded ones.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.toast_hip_last_error.restype = C.c_char_p
        _lib.toast_hip_version.restype = C.c_char_p
    return _lib


def real_lib():
    """The loaded library itself, never the recorder of an active ``capture()`` block (finalizers
    and other calls that must EXECUTE now, not be recorded into a replayed plan)."""
    if _lib is None:
        saved = _capture
        try:
            globals()["_capture"] = None
            lib()
        finally:
            globals()["_capture"] = saved
    return _lib


def set_deterministic(on):
    """Order-deterministic A^T scatter (toast_hip_set_deterministic; also TOAST_HIP_DETERMINISTIC=1)."""
    real_lib().toast_hip_set_deterministic(C.c_int(1 if on else 0))


def set_tuning(key, value):
    """Run-time switches of the library (toast_hip_set_tuning): "pair" = 0 / 1, "det_major" = 0 / 1, "vec2" = 0 / 1."""
    _check(real_lib().toast_hip_set_tuning(key.encode(), C.c_int(int(value))))


def get_deterministic():
    return bool(real_lib().toast_hip_get_deterministic())


def set_stokes_reference_nan(on):
    """NaN Q / U weights where the reference's formulation produces them (poles within rounding)."""
    real_lib().toast_hip_set_stokes_reference_nan(C.c_int(1 if on else 0))


def accel_generation():
    """Memory-manager generation counter (toast_hip_accel_generation)."""
    out = C.c_uint64(0)
    lib()
    if _lib.toast_hip_accel_generation(C.byref(out)) != 0:
        raise RuntimeError(_lib.toast_hip_last_error().decode())
    return int(out.value)


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().toast_hip_last_error().decode())


# --------------------------------------------------------------------------- argument helpers
def _buf(a, name, dtype, ndim, shape=None):
    """extract_buffer-style validation; returns the array (common.hpp:50-121)."""
    if not isinstance(a, np.ndarray):
        raise RuntimeError(f"Object {name} is not a NumPy array")
    if a.dtype != dtype:
        raise RuntimeError(f"Object {name} has dtype {a.dtype}, expected {np.dtype(dtype)}")
    if a.ndim != ndim:
        raise RuntimeError(f"Object {name} has {a.ndim} dimensions instead of {ndim}")
    if not a.flags["C_CONTIGUOUS"]:
        raise RuntimeError(f"Object {name} is not contiguous.")
    if shape is not None:
        for i, (got, want) in enumerate(zip(a.shape, shape)):
            if want >= 0 and got != want:
                raise RuntimeError(f"Object {name} dimension {i} has length {got} instead of {want}")
    return a


def _p(a):
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        if _capture is not None:
            _capture._plan.keep.append(a)
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(int(a))


def _i64(x):
    return C.c_int64(int(x))


def _u8(x):
    return C.c_uint8(int(x))


def _int(x):
    return C.c_int(int(bool(x)))


# --------------------------------------------------------------------------- memory manager
def accel_enabled():
    return bool(lib().toast_hip_accel_enabled())


def accel_assign_device(node_procs, node_rank, mem_gb, disabled):
    _check(lib().toast_hip_accel_assign_device(int(node_procs), int(node_rank), C.c_double(mem_gb), _int(disabled)))


def accel_get_device():
    d = C.c_int(0)
    _check(lib().toast_hip_accel_get_device(C.byref(d)))
    return d.value


def _raw(buf):
    a = np.asarray(buf)
    if not a.flags["C_CONTIGUOUS"]:
        raise RuntimeError("accel_* buffers must be contiguous")
    return a


def accel_present(buf, name="NA"):
    a = _raw(buf)
    r = C.c_int(0)
    _check(lib().toast_hip_accel_present(_p(a), C.c_size_t(a.nbytes), C.byref(r)))
    return bool(r.value)


def accel_create(buf, name="NA"):
    a = _raw(buf)
    _check(lib().toast_hip_accel_create(_p(a), C.c_size_t(a.nbytes), name.encode()))


def accel_reset(buf, name="NA"):
    a = _raw(buf)
    _check(lib().toast_hip_accel_reset(_p(a), C.c_size_t(a.nbytes), name.encode()))


def accel_update_device(buf, name="NA"):
    a = _raw(buf)
    _check(lib().toast_hip_accel_update_device(_p(a), C.c_size_t(a.nbytes), name.encode()))


def accel_update_host(buf, name="NA"):
    a = _raw(buf)
    _check(lib().toast_hip_accel_update_host(_p(a), C.c_size_t(a.nbytes), name.encode()))


def device_malloc(nbytes, flags=-1):
    """Raw device memory: flags = -1 with the memory manager's allocation / placement policy (what operators get),
    0 a plain hipMalloc, otherwise hipExtMallocWithFlags flags.  Returns the device address."""
    out = C.c_void_p(0)
    _check(real_lib().toast_hip_device_malloc(C.c_size_t(int(nbytes)), C.c_int(int(flags)), C.byref(out)))
    return int(out.value or 0)


def device_free(ptr):
    _check(real_lib().toast_hip_device_free(C.c_void_p(int(ptr))))


def device_release(ptr, nbytes):
    """Give a block from ``device_malloc`` back to the arena (same as ``device_free``)."""
    _check(real_lib().toast_hip_device_release(C.c_void_p(int(ptr)), C.c_size_t(int(nbytes))))


def device_malloc_vmm(nbytes, chunk_mb, shuffled=False):
    """EXPERIMENT: a virtual range backed by separately created physical chunks (toast_hip_device_malloc_vmm)."""
    out = C.c_void_p(0)
    _check(real_lib().toast_hip_device_malloc_vmm(C.c_size_t(int(nbytes)), C.c_int(int(chunk_mb)), _int(shuffled),
                                                  C.byref(out)))
    return int(out.value or 0)


def probe_stream_split(ptrs, nbytes_each):
    """ms of the probe pass with its rows dealt round-robin to the ranges ``ptrs`` (toast_hip_probe_stream_split)."""
    ms = C.c_double(0.0)
    arr = (C.c_void_p * len(ptrs))(*[int(p) for p in ptrs])
    _check(real_lib().toast_hip_probe_stream_split(arr, C.c_int(len(ptrs)), C.c_size_t(int(nbytes_each)), C.byref(ms)))
    return float(ms.value)


def probe_byte_mix(d_pixels, d_weights, d_tod, d_out, n_det, n_samp, stream=0):
    """The byte mix of scan_map (d_out given) / build_noise_weighted (d_out = 0) as plain streams
    (toast_hip_probe_byte_mix_dev); asynchronous on ``stream``."""
    _check(real_lib().toast_hip_probe_byte_mix_dev(C.c_void_p(int(d_pixels)), C.c_void_p(int(d_weights)), C.c_void_p(int(d_tod)),
                                                   C.c_void_p(int(d_out)) if d_out else None, C.c_int64(int(n_det)),
                                                   C.c_int64(int(n_samp)), C.c_void_p(int(stream)) if stream else None))


def accel_mem_info():
    """(free, total) bytes of device memory (toast_hip_accel_mem_info)."""
    f, t = C.c_size_t(0), C.c_size_t(0)
    _check(real_lib().toast_hip_accel_mem_info(C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def accel_release_cached():
    """Released device blocks kept for reuse, and slow candidates held by the placement policy, go back to the driver."""
    _check(real_lib().toast_hip_accel_release_cached())


class _ArenaStats(C.Structure):
    _fields_ = [("slabs", C.c_int64), ("slab_bytes", C.c_uint64), ("used_bytes", C.c_uint64),
                ("peak_used_bytes", C.c_uint64), ("slab_mallocs", C.c_int64), ("slab_frees", C.c_int64),
                ("malloc_ms", C.c_double), ("max_malloc_ms", C.c_double), ("touch_ms", C.c_double),
                ("allocs", C.c_int64), ("releases", C.c_int64), ("direct_mallocs", C.c_int64), ("failed", C.c_int64),
                ("interleaved_slabs", C.c_int64), ("chunks", C.c_int64), ("chunks_other_zone", C.c_int64),
                ("chunks_created", C.c_int64), ("interleave_ms", C.c_double), ("same_zone_TBs", C.c_double),
                ("chunks_other_wanted", C.c_int64), ("searches", C.c_int64), ("searches_exhausted", C.c_int64),
                ("searches_capped_ms", C.c_int64), ("probes", C.c_int64), ("probes_by_clock", C.c_int64),
                ("create_ms_per_chunk", C.c_double), ("search_ms", C.c_double), ("slabs_third_zone", C.c_int64), ("read_mostly_zones", C.c_int64)]


def alloc_stats():
    """Counters of the device memory arena of this process (toast_hip_arena_stats): slabs held, bytes in live blocks,
    hipMalloc calls and the wall time inside them."""
    st = _ArenaStats()
    _check(real_lib().toast_hip_arena_stats(C.byref(st)))
    out = {name: getattr(st, name) for name, _ in _ArenaStats._fields_}
    out["slab_GB"] = out.pop("slab_bytes") / 2.0 ** 30
    out["used_GB"] = out.pop("used_bytes") / 2.0 ** 30
    out["peak_used_GB"] = out.pop("peak_used_bytes") / 2.0 ** 30
    # the verdict of toast_hip_arena_placement_status, from the same counters
    out["placement_ok"] = bool(out["interleaved_slabs"] > 0 and out["chunks_other_zone"] >= out["chunks_other_wanted"])
    out["search_exhausted"] = bool(out["searches_exhausted"] > 0)
    return out


def arena_placement_status():
    """(placement_ok, search_exhausted, chunks_other_zone, chunks_other_wanted) of the zone-interleaved slabs
    (toast_hip_arena_placement_status): ok = every slot meant for the other HBM zone holds a chunk that measured clear of
    the read-mostly slabs."""
    ok, ex = C.c_int(0), C.c_int(0)
    a, b = C.c_int64(0), C.c_int64(0)
    _check(real_lib().toast_hip_arena_placement_status(C.byref(ok), C.byref(ex), C.byref(a), C.byref(b)))
    return bool(ok.value), bool(ex.value), int(a.value), int(b.value)


def arena_reserve(nbytes, streamed=False):
    """Make the arena (or its part for streamed blocks) hold at least ``nbytes`` (toast_hip_arena_reserve[_streamed])."""
    fn = real_lib().toast_hip_arena_reserve_streamed if streamed else real_lib().toast_hip_arena_reserve
    _check(fn(C.c_size_t(int(nbytes))))


def arena_block_zone(device_ptr, nbytes):
    """(inside a zone-interleaved slab, chunks of the read-mostly zone it touches, chunks of the other zone)
    (toast_hip_arena_block_zone)."""
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    _check(real_lib().toast_hip_arena_block_zone(C.c_void_p(int(device_ptr)), C.c_size_t(int(nbytes)), C.byref(a), C.byref(b),
                                                 C.byref(c)))
    return bool(a.value), int(b.value), int(c.value)


def arena_zone_threshold(rates, level):
    """The zone search's class threshold on given rates (toast_hip_arena_zone_threshold; no device needed)."""
    r = np.ascontiguousarray(rates, dtype=np.float64)
    thr = C.c_double(0.0)
    _check(real_lib().toast_hip_arena_zone_threshold(r.ctypes.data_as(C.c_void_p), C.c_int(r.size), C.c_double(float(level)),
                                                     C.byref(thr)))
    return float(thr.value)


def arena_selftest(seed, n_ops, granule, slab_bytes, max_block):
    """The arena's sub-allocation logic on host memory (toast_hip_arena_selftest); raises on an inconsistency."""
    _check(real_lib().toast_hip_arena_selftest(C.c_uint64(int(seed)), C.c_int(int(n_ops)), C.c_size_t(int(granule)),
                                               C.c_size_t(int(slab_bytes)), C.c_size_t(int(max_block))))


def probe_stream(ptr, nbytes):
    """ms of one read + write pass over a device range with 1024 rows in flight (toast_hip_probe_stream)."""
    ms = C.c_double(0.0)
    _check(real_lib().toast_hip_probe_stream(C.c_void_p(int(ptr)), C.c_size_t(int(nbytes)), C.byref(ms)))
    return float(ms.value)


def accel_update_device_parts(buf, part_end, name="NA"):
    """Start the upload of a registered buffer in parts (byte offsets ``part_end``, the last one = its size) on the
    library's upload stream; returns once the copies are enqueued."""
    a = _raw(buf)
    pe = np.ascontiguousarray(part_end, dtype=np.uint64)
    _check(real_lib().toast_hip_accel_update_device_parts(_p(a), C.c_size_t(a.nbytes), name.encode(), _p(pe),
                                                          C.c_int(pe.size)))


def accel_update_device_wait(buf, part, stream=0):
    """Later work on ``stream`` waits for part ``part`` of the upload (the host does not)."""
    _check(real_lib().toast_hip_accel_update_device_wait(_p(_raw(buf)), C.c_int(int(part)), _p(stream)))


def accel_update_device_arrived(buf, part):
    """True once part ``part`` of the upload is on the device (does not wait)."""
    out = C.c_int(0)
    _check(real_lib().toast_hip_accel_update_device_arrived(_p(_raw(buf)), C.c_int(int(part)), C.byref(out)))
    return bool(out.value)


def accel_update_device_finish(buf):
    """Block until every part of the upload has arrived."""
    _check(real_lib().toast_hip_accel_update_device_finish(_p(_raw(buf))))


def accel_delete(buf, name="NA"):
    a = _raw(buf)
    _check(lib().toast_hip_accel_delete(_p(a), C.c_size_t(a.nbytes), name.encode()))


def accel_device_ptr(buf):
    a = _raw(buf)
    out = C.c_void_p(0)
    _check(lib().toast_hip_accel_device_ptr(_p(a), C.byref(out)))
    return out.value


def accel_dump():
    _check(lib().toast_hip_accel_dump())


def set_stream(stream):
    _check(lib().toast_hip_set_stream(C.c_void_p(int(stream))))


def synchronize():
    _check(lib().toast_hip_synchronize())


# --------------------------------------------------------------------------- host-pointer level
def pointing_detector(focalplane, boresight, quat_index, quats, intervals, shared_flags,
                      shared_flag_mask, use_accel=False):
    qi = _buf(quat_index, "quat_index", np.int32, 1)
    n_det = qi.shape[0]
    fp = _buf(focalplane, "focalplane", np.float64, 2, (n_det, 4))
    bore = _buf(boresight, "boresight", np.float64, 2, (-1, 4))
    n_samp = bore.shape[0]
    q = _buf(quats, "quats", np.float64, 3, (-1, n_samp, 4))
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    fl = _buf(shared_flags, "flags", np.uint8, 1)
    _check(lib().toast_hip_pointing_detector(
        _p(fp), _p(bore), _p(qi), _i64(n_det), _p(q), _i64(q.shape[0]), _i64(n_samp), _p(iv),
        _i64(iv.shape[0]), _p(fl), _i64(fl.shape[0]), _u8(shared_flag_mask), _int(use_accel)))


def pixels_healpix(quat_index, quats, shared_flags, shared_flag_mask, pixel_index, pixels, intervals,
                   hit_submaps, n_pix_submap, nside, nest, use_accel=False):
    qi = _buf(quat_index, "quat_index", np.int32, 1)
    n_det = qi.shape[0]
    pi = _buf(pixel_index, "pixel_index", np.int32, 1, (n_det,))
    px = _buf(pixels, "pixels", np.int64, 2)
    n_samp = px.shape[1]
    q = _buf(quats, "quats", np.float64, 3, (-1, n_samp, 4))
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    hs = _buf(hit_submaps, "hit_submaps", np.uint8, 1)
    fl = _buf(shared_flags, "flags", np.uint8, 1)
    _check(lib().toast_hip_pixels_healpix(
        _p(qi), _i64(n_det), _p(q), _i64(q.shape[0]), _p(fl), _i64(fl.shape[0]), _u8(shared_flag_mask),
        _p(pi), _p(px), _i64(px.shape[0]), _i64(n_samp), _p(iv), _i64(iv.shape[0]), _p(hs),
        _i64(hs.shape[0]), _i64(n_pix_submap), _i64(nside), _int(nest), _int(use_accel)))


def stokes_weights_IQU(quat_index, quats, weight_index, weights, hwp, intervals, epsilon, gamma, cal,
                       IAU, use_accel=False):
    qi = _buf(quat_index, "quat_index", np.int32, 1)
    n_det = qi.shape[0]
    wi = _buf(weight_index, "weight_index", np.int32, 1, (n_det,))
    w = _buf(weights, "weights", np.float64, 3, (-1, -1, 3))
    n_samp = w.shape[1]
    q = _buf(quats, "quats", np.float64, 3, (-1, n_samp, 4))
    h = _buf(hwp, "hwp", np.float64, 1)
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    e = _buf(epsilon, "epsilon", np.float64, 1, (n_det,))
    cl = _buf(cal, "cal", np.float64, 1, (n_det,))
    g = _buf(gamma, "gamma", np.float64, 1, (n_det,))
    _check(lib().toast_hip_stokes_weights_IQU(
        _p(qi), _i64(n_det), _p(q), _i64(q.shape[0]), _p(wi), _p(w), _i64(w.shape[0]), _i64(n_samp),
        _p(h), _i64(h.shape[0]), _p(iv), _i64(iv.shape[0]), _p(e), _p(g), _p(cl), _int(IAU),
        _int(use_accel)))


def stokes_weights_QU(quat_index, quats, weight_index, weights, hwp, intervals, epsilon, gamma, cal,
                       IAU, use_accel=False):
    qi = _buf(quat_index, "quat_index", np.int32, 1)
    n_det = qi.shape[0]
    wi = _buf(weight_index, "weight_index", np.int32, 1, (n_det,))
    w = _buf(weights, "weights", np.float64, 3, (-1, -1, 2))
    n_samp = w.shape[1]
    q = _buf(quats, "quats", np.float64, 3, (-1, n_samp, 4))
    h = _buf(hwp, "hwp", np.float64, 1)
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    e = _buf(epsilon, "epsilon", np.float64, 1, (n_det,))
    cl = _buf(cal, "cal", np.float64, 1, (n_det,))
    g = _buf(gamma, "gamma", np.float64, 1, (n_det,))
    _check(lib().toast_hip_stokes_weights_QU(
        _p(qi), _i64(n_det), _p(q), _i64(q.shape[0]), _p(wi), _p(w), _i64(w.shape[0]), _i64(n_samp),
        _p(h), _i64(h.shape[0]), _p(iv), _i64(iv.shape[0]), _p(e), _p(g), _p(cl), _int(IAU),
        _int(use_accel)))


def stokes_weights_I(weight_index, weights, intervals, cal, use_accel=False):
    wi = _buf(weight_index, "weight_index", np.int32, 1)
    n_det = wi.shape[0]
    w = _buf(weights, "weights", np.float64, 2, (n_det, -1))
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    cl = _buf(cal, "cal", np.float64, 1, (n_det,))
    _check(lib().toast_hip_stokes_weights_I(
        _p(wi), _i64(n_det), _p(w), _i64(w.shape[0]), _i64(w.shape[1]), _p(iv), _i64(iv.shape[0]),
        _p(cl), _int(use_accel)))


def _weights_nnz(weights, n_samp):
    if not isinstance(weights, np.ndarray):
        raise RuntimeError("Object weights is not a NumPy array")
    if weights.ndim == 2:
        return _buf(weights, "weights", np.float64, 2, (-1, n_samp)), 1
    w = _buf(weights, "weights", np.float64, 3, (-1, n_samp, -1))
    return w, w.shape[2]


def _scan_map(map_dtype, global2local, n_pix_submap, mapdata, det_data, data_index, pixels, pixel_index,
              weights, weight_index, intervals, data_scale, should_zero, should_subtract, should_scale,
              use_accel):
    pi = _buf(pixel_index, "pixel_index", np.int32, 1)
    n_det = pi.shape[0]
    px = _buf(pixels, "pixels", np.int64, 2)
    n_samp = px.shape[1]
    wi = _buf(weight_index, "weight_index", np.int32, 1, (n_det,))
    w, nnz = _weights_nnz(weights, n_samp)
    di = _buf(data_index, "data_index", np.int32, 1, (n_det,))
    dd = _buf(det_data, "det_data", np.float64, 2, (-1, n_samp))
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    g2l = _buf(global2local, "global2local", np.int64, 1)
    m = _buf(mapdata, "mapdata", map_dtype, 3, (-1, n_pix_submap, nnz))
    _check(lib().toast_hip_scan_map(
        C.c_int(_MAP_CODES[np.dtype(map_dtype)]), _p(g2l), _i64(g2l.shape[0]), _i64(n_pix_submap), _p(m),
        _i64(m.shape[0]), _i64(nnz), _p(dd), _i64(dd.shape[0]), _p(di), _p(px), _i64(px.shape[0]), _p(pi),
        _p(w), _i64(w.shape[0]), _p(wi), _i64(n_det), _i64(n_samp), _p(iv), _i64(iv.shape[0]),
        C.c_double(data_scale), _int(should_zero), _int(should_subtract), _int(should_scale),
        _int(use_accel)))


def ops_scan_map_float64(*args):
    return _scan_map(np.float64, *args)


def ops_scan_map_float32(*args):
    return _scan_map(np.float32, *args)


def ops_scan_map_int64(*args):
    return _scan_map(np.int64, *args)


def ops_scan_map_int32(*args):
    return _scan_map(np.int32, *args)


def build_noise_weighted(global2local, zmap, pixel_index, pixels, weight_index, weights, data_index,
                         det_data, flag_index, det_flags, det_scale, det_flag_mask, intervals,
                         shared_flags, shared_flag_mask, use_accel=False):
    pi = _buf(pixel_index, "pixel_index", np.int32, 1)
    n_det = pi.shape[0]
    px = _buf(pixels, "pixels", np.int64, 2)
    n_samp = px.shape[1]
    wi = _buf(weight_index, "weight_index", np.int32, 1, (n_det,))
    w, nnz = _weights_nnz(weights, n_samp)
    di = _buf(data_index, "data_index", np.int32, 1, (n_det,))
    dd = _buf(det_data, "det_data", np.float64, 2, (-1, n_samp))
    fi = _buf(flag_index, "flag_index", np.int32, 1, (n_det,))
    ds = _buf(det_scale, "det_scale", np.float64, 1, (n_det,))
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    g2l = _buf(global2local, "global2local", np.int64, 1)
    z = _buf(zmap, "zmap", np.float64, 3, (-1, -1, nnz))
    sf = _buf(shared_flags, "flags", np.uint8, 1)
    df = _buf(det_flags, "det_flags", np.uint8, 2)
    _check(lib().toast_hip_build_noise_weighted(
        _p(g2l), _i64(g2l.shape[0]), _p(z), _i64(z.shape[0]), _i64(z.shape[1]), _i64(nnz), _p(pi), _p(px),
        _i64(px.shape[0]), _p(wi), _p(w), _i64(w.shape[0]), _p(di), _p(dd), _i64(dd.shape[0]), _p(fi),
        _p(df), _i64(df.shape[0]), _i64(df.shape[1]), _p(ds), _u8(det_flag_mask), _i64(n_det),
        _i64(n_samp), _p(iv), _i64(iv.shape[0]), _p(sf), _i64(sf.shape[0]), _u8(shared_flag_mask),
        _int(use_accel)))


def noise_weight(det_data, data_index, intervals, detector_weights, use_accel=False):
    di = _buf(data_index, "data_index", np.int32, 1)
    n_det = di.shape[0]
    dd = _buf(det_data, "det_data", np.float64, 2)
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    dw = _buf(detector_weights, "detector_weights", np.float64, 1, (n_det,))
    _check(lib().toast_hip_noise_weight(
        _p(dd), _i64(dd.shape[0]), _i64(dd.shape[1]), _p(di), _i64(n_det), _p(iv), _i64(iv.shape[0]),
        _p(dw), _int(use_accel)))


def cov_apply_diag(nsub, subsize, nnz, mat, vec, use_accel=False):
    m = np.asarray(mat)
    v = np.asarray(vec)
    if m.dtype != np.float64 or v.dtype != np.float64:
        raise RuntimeError("cov_apply_diag needs float64 buffers")
    _check(lib().toast_hip_cov_apply_diag(_i64(nsub), _i64(subsize), _i64(nnz), _p(m), _p(v), _int(use_accel)))


def cov_mult_diag(nsub, subsize, nnz, data1, data2, use_accel=False):
    a = np.asarray(data1)
    b = np.asarray(data2)
    if a.dtype != np.float64 or b.dtype != np.float64:
        raise RuntimeError("cov_mult_diag needs float64 buffers")
    if a.size != b.size:
        raise RuntimeError("Buffer sizes are not consistent.")
    _check(lib().toast_hip_cov_mult_diag(_i64(nsub), _i64(subsize), _i64(nnz), _p(a), _p(b), _int(use_accel)))


def template_offset_add_to_signal(step_length, amp_offset, n_amp_views, amplitudes, amplitude_flags,
                                  data_index, det_data, intervals, use_accel=False):
    a = _buf(amplitudes, "amplitudes", np.float64, 1)
    af = _buf(amplitude_flags, "amplitude_flags", np.uint8, 1, (a.shape[0],))
    dd = _buf(det_data, "det_data", np.float64, 2)
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    nv = _buf(n_amp_views, "n_amp_views", np.int64, 1, (iv.shape[0],))
    _check(lib().toast_hip_template_offset_add_to_signal(
        _i64(step_length), _i64(amp_offset), _p(nv), _p(a), _p(af), _i64(a.shape[0]),
        C.c_int32(int(data_index)), _p(dd), _i64(dd.shape[0]), _i64(dd.shape[1]), _p(iv),
        _i64(iv.shape[0]), _int(use_accel)))


def template_offset_project_signal(data_index, det_data, flag_index, flag_data, flag_mask, step_length,
                                   amp_offset, n_amp_views, amplitudes, amplitude_flags, intervals,
                                   use_accel=False):
    a = _buf(amplitudes, "amplitudes", np.float64, 1)
    af = _buf(amplitude_flags, "amplitude_flags", np.uint8, 1, (a.shape[0],))
    dd = _buf(det_data, "det_data", np.float64, 2)
    n_samp = dd.shape[1]
    iv = _buf(intervals, "intervals", interval_dtype, 1)
    nv = _buf(n_amp_views, "n_amp_views", np.int64, 1, (iv.shape[0],))
    n_flag_rows = 0
    fd = None
    if int(flag_index) >= 0:
        fd = _buf(flag_data, "flag_data", np.uint8, 2, (-1, n_samp))
        n_flag_rows = fd.shape[0]
    _check(lib().toast_hip_template_offset_project_signal(
        C.c_int32(int(data_index)), _p(dd), _i64(dd.shape[0]), C.c_int32(int(flag_index)), _p(fd),
        _i64(n_flag_rows), _u8(flag_mask), _i64(step_length), _i64(amp_offset), _p(nv), _p(a), _p(af),
        _i64(a.shape[0]), _i64(n_samp), _p(iv), _i64(iv.shape[0]), _int(use_accel)))


def template_offset_apply_diag_precond(offset_var, amplitudes_in, amplitude_flags, amplitudes_out,
                                       use_accel=False):
    ai = _buf(amplitudes_in, "amplitudes_in", np.float64, 1)
    n = ai.shape[0]
    ao = _buf(amplitudes_out, "amplitudes_out", np.float64, 1, (n,))
    ov = _buf(offset_var, "offset_var", np.float64, 1, (n,))
    af = _buf(amplitude_flags, "amplitude_flags", np.uint8, 1, (n,))
    _check(lib().toast_hip_template_offset_apply_diag_precond(_p(ov), _p(ai), _p(af), _p(ao), _i64(n),
                                                              _int(use_accel)))


# --------------------------------------------------------------------------- device-pointer level

class OtfPointing(C.Structure):
    """``toast_hip_otf_pointing`` (include/toast_hip.h): arguments of the three pointing operators
    for the on-the-fly kernels.  Build with :func:`otf_pointing`, which keeps the host arrays alive."""

    _fields_ = [
        ("d_boresight", C.c_void_p),
        ("d_shared_flags", C.c_void_p),
        ("n_shared_flags", C.c_int64),
        ("shared_flag_mask", C.c_uint8),
        ("focalplane", C.c_void_p),
        ("d_hwp", C.c_void_p),
        ("n_hwp", C.c_int64),
        ("epsilon", C.c_void_p),
        ("gamma", C.c_void_p),
        ("cal", C.c_void_p),
        ("IAU", C.c_int),
        ("nside", C.c_int64),
        ("nest", C.c_int),
        ("nnz", C.c_int),
        ("d_compact_pixels", C.c_void_p),
        ("compact_index", C.c_void_p),
        ("d_hwp_table", C.c_void_p),
    ]


def otf_pointing(d_boresight, focalplane, nside, nest, nnz, d_shared_flags=0, n_shared_flags=0, shared_flag_mask=0,
                 d_hwp=0, n_hwp=0, epsilon=None, gamma=None, cal=None, IAU=False, d_compact_pixels=0,
                 compact_index=None, d_hwp_table=0):
    fp = np.ascontiguousarray(focalplane, dtype=np.float64)
    if fp.ndim != 2 or fp.shape[1] != 4:
        raise RuntimeError("focalplane should be a [n_det, 4] float64 array")
    keep = [fp]

    def small(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != (fp.shape[0],):
            raise RuntimeError("epsilon / gamma / cal should have one entry per detector")
        keep.append(a)
        return a.ctypes.data

    pt = OtfPointing(int(d_boresight), int(d_shared_flags) or None, int(n_shared_flags), int(shared_flag_mask),
                     fp.ctypes.data, int(d_hwp) or None, int(n_hwp), small(epsilon), small(gamma), small(cal),
                     int(bool(IAU)), int(nside), int(bool(nest)), int(nnz), None, None, int(d_hwp_table) or None)
    if d_compact_pixels:
        ci = np.ascontiguousarray(compact_index, dtype=np.int32)
        if ci.shape != (fp.shape[0],):
            raise RuntimeError("compact_index should have one entry per detector")
        keep.append(ci)
        pt.d_compact_pixels = int(d_compact_pixels)
        pt.compact_index = ci.ctypes.data
    pt._keep = keep
    pt.n_det = fp.shape[0]
    return pt


class _Dev:
    """``toast_hip_*_dev``: large arrays are device pointers (ints), small ones NumPy arrays."""

    @staticmethod
    def _small(a, dtype):
        return np.ascontiguousarray(a, dtype=dtype)

    def pointing_detector(self, focalplane, d_boresight, quat_index, d_quats, n_samp, intervals,
                          d_shared_flags=0, n_flags=0, mask=0, stream=0):
        fp = self._small(focalplane, np.float64)
        qi = self._small(quat_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_pointing_detector_dev(
            _p(fp), _p(d_boresight), _p(qi), _i64(qi.size), _p(d_quats), _i64(n_samp), _p(iv),
            _i64(iv.size), _p(d_shared_flags), _i64(n_flags), _u8(mask), _p(stream)))

    def pixels_healpix(self, quat_index, d_quats, d_shared_flags, n_flags, mask, pixel_index, d_pixels,
                       n_samp, intervals, d_hit_submaps, n_submap, n_pix_submap, nside, nest, stream=0):
        qi = self._small(quat_index, np.int32)
        pi = self._small(pixel_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_pixels_healpix_dev(
            _p(qi), _i64(qi.size), _p(d_quats), _p(d_shared_flags), _i64(n_flags), _u8(mask), _p(pi),
            _p(d_pixels), _i64(n_samp), _p(iv), _i64(iv.size), _p(d_hit_submaps), _i64(n_submap),
            _i64(n_pix_submap), _i64(nside), _int(nest), _p(stream)))

    def stokes_weights_IQU(self, quat_index, d_quats, weight_index, d_weights, n_samp, d_hwp, n_hwp,
                           intervals, epsilon, gamma, cal, iau, stream=0):
        qi = self._small(quat_index, np.int32)
        wi = self._small(weight_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        e = self._small(epsilon, np.float64)
        g = self._small(gamma, np.float64)
        cl = self._small(cal, np.float64)
        _check(lib().toast_hip_stokes_weights_IQU_dev(
            _p(qi), _i64(qi.size), _p(d_quats), _p(wi), _p(d_weights), _i64(n_samp), _p(d_hwp),
            _i64(n_hwp), _p(iv), _i64(iv.size), _p(e), _p(g), _p(cl), _int(iau), _p(stream)))

    def stokes_weights_QU(self, quat_index, d_quats, weight_index, d_weights, n_samp, d_hwp, n_hwp,
                           intervals, epsilon, gamma, cal, iau, stream=0):
        qi = self._small(quat_index, np.int32)
        wi = self._small(weight_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        e = self._small(epsilon, np.float64)
        g = self._small(gamma, np.float64)
        cl = self._small(cal, np.float64)
        _check(lib().toast_hip_stokes_weights_QU_dev(
            _p(qi), _i64(qi.size), _p(d_quats), _p(wi), _p(d_weights), _i64(n_samp), _p(d_hwp),
            _i64(n_hwp), _p(iv), _i64(iv.size), _p(e), _p(g), _p(cl), _int(iau), _p(stream)))

    def stokes_weights_I(self, weight_index, d_weights, n_samp, intervals, cal, stream=0):
        wi = self._small(weight_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        cl = self._small(cal, np.float64)
        _check(lib().toast_hip_stokes_weights_I_dev(_p(wi), _i64(wi.size), _p(d_weights), _i64(n_samp),
                                                    _p(iv), _i64(iv.size), _p(cl), _p(stream)))

    def scan_map(self, map_dtype, d_g2l, n_pix_submap, d_mapdata, nnz, d_det_data, data_index, d_pixels,
                 pixel_index, d_weights, weight_index, n_samp, intervals, data_scale=1.0, should_zero=False,
                 should_subtract=False, should_scale=False, det_weights=None, stream=0):
        di = self._small(data_index, np.int32)
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        dw = None if det_weights is None else self._small(det_weights, np.float64)
        _check(lib().toast_hip_scan_map_dev(
            C.c_int(_MAP_CODES[np.dtype(map_dtype)]), _p(d_g2l), _i64(n_pix_submap), _p(d_mapdata), _i64(nnz),
            _p(d_det_data), _p(di), _p(d_pixels), _p(pi), _p(d_weights), _p(wi), _i64(di.size),
            _i64(n_samp), _p(iv), _i64(iv.size), C.c_double(data_scale), _int(should_zero),
            _int(should_subtract), _int(should_scale), _p(dw), _p(stream)))

    def build_noise_weighted(self, d_g2l, d_zmap, n_pix_submap, nnz, pixel_index, d_pixels, weight_index,
                             d_weights, data_index, d_det_data, flag_index, d_det_flags, n_flag_samp,
                             det_scale, det_flag_mask, n_samp, intervals, d_shared_flags, n_shared_flags,
                             shared_flag_mask, stream=0):
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        di = self._small(data_index, np.int32)
        fi = self._small(flag_index, np.int32)
        ds = self._small(det_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_build_noise_weighted_dev(
            _p(d_g2l), _p(d_zmap), _i64(n_pix_submap), _i64(nnz), _p(pi), _p(d_pixels), _p(wi),
            _p(d_weights), _p(di), _p(d_det_data), _p(fi), _p(d_det_flags), _i64(n_flag_samp), _p(ds),
            _u8(det_flag_mask), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(d_shared_flags),
            _i64(n_shared_flags), _u8(shared_flag_mask), _p(stream)))

    def build_cov_hits_signal(self, d_g2l, d_invcov, d_hits, d_zmap, n_pix_submap, nnz, pixel_index, d_pixels,
                              weight_index, d_weights, data_index, d_det_data, flag_index, d_det_flags, n_flag_samp,
                              det_scale, data_scale, det_flag_mask, n_samp, intervals, d_shared_flags, n_shared_flags,
                              shared_flag_mask, stream=0):
        """Inverse covariance, hits and zmap += A^T N^-1 d in one sweep (toast_hip_build_cov_hits_signal_dev).  Returns True
        when one kernel did all three, False when the library ran the separate sweeps (same results)."""
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        di = self._small(data_index, np.int32)
        fi = self._small(flag_index, np.int32)
        ds = self._small(det_scale, np.float64)
        ss = self._small(data_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        fused = C.c_int(0)
        _check(lib().toast_hip_build_cov_hits_signal_dev(
            _p(d_g2l), _p(d_invcov), _p(d_hits), _p(d_zmap), _i64(n_pix_submap), _i64(nnz), _p(pi), _p(d_pixels), _p(wi),
            _p(d_weights), _p(di), _p(d_det_data), _p(fi), _p(d_det_flags), _i64(n_flag_samp), _p(ds), _p(ss),
            _u8(det_flag_mask), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(d_shared_flags),
            _i64(n_shared_flags), _u8(shared_flag_mask), C.byref(fused), _p(stream)))
        return bool(fused.value)

    def noise_weight(self, d_det_data, n_samp, data_index, intervals, detector_weights, stream=0):
        di = self._small(data_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        dw = self._small(detector_weights, np.float64)
        _check(lib().toast_hip_noise_weight_dev(_p(d_det_data), _i64(n_samp), _p(di), _i64(di.size), _p(iv),
                                                _i64(iv.size), _p(dw), _p(stream)))

    def cov_apply_diag(self, n_sub, subsize, nnz, d_mat, d_vec, stream=0):
        _check(lib().toast_hip_cov_apply_diag_dev(_i64(n_sub), _i64(subsize), _i64(nnz), _p(d_mat), _p(d_vec),
                                                  _p(stream)))

    def template_offset_add_to_signal(self, step_length, amp_offset, n_amp_views, d_amplitudes,
                                      d_amplitude_flags, data_index, d_det_data, n_samp, intervals, stream=0):
        iv = self._small(intervals, interval_dtype)
        nv = self._small(n_amp_views, np.int64)
        _check(lib().toast_hip_template_offset_add_to_signal_dev(
            _i64(step_length), _i64(amp_offset), _p(nv), _p(d_amplitudes), _p(d_amplitude_flags),
            C.c_int32(int(data_index)), _p(d_det_data), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def template_offset_project_signal(self, data_index, d_det_data, flag_index, d_flag_data, flag_mask,
                                       step_length, amp_offset, n_amp_views, d_amplitudes,
                                       d_amplitude_flags, n_samp, intervals, stream=0):
        iv = self._small(intervals, interval_dtype)
        nv = self._small(n_amp_views, np.int64)
        _check(lib().toast_hip_template_offset_project_signal_dev(
            C.c_int32(int(data_index)), _p(d_det_data), C.c_int32(int(flag_index)), _p(d_flag_data),
            _u8(flag_mask), _i64(step_length), _i64(amp_offset), _p(nv), _p(d_amplitudes),
            _p(d_amplitude_flags), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def template_offset_apply_diag_precond(self, d_offset_var, d_amp_in, d_amp_flags, d_amp_out, n_amp,
                                           stream=0):
        _check(lib().toast_hip_template_offset_apply_diag_precond_dev(
            _p(d_offset_var), _p(d_amp_in), _p(d_amp_flags), _p(d_amp_out), _i64(n_amp), _p(stream)))

    def offset_add_to_signal_multi(self, step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags,
                                   data_index, d_det_data, n_samp, intervals, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        di = self._small(data_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_template_offset_add_to_signal_multi_dev(
            _i64(step_length), _p(ao), _p(nv), _p(d_amplitudes), _p(d_amplitude_flags), _p(di), _i64(di.size),
            _p(d_det_data), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def offset_project_signal_multi(self, data_index, d_det_data, flag_index, d_flag_data, flag_mask, step_length,
                                    amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags, n_samp, intervals,
                                    stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        di = self._small(data_index, np.int32)
        fi = None if flag_index is None else self._small(flag_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_template_offset_project_signal_multi_dev(
            _p(di), _p(d_det_data), _p(fi), _p(d_flag_data), _u8(flag_mask), _i64(step_length), _p(ao), _p(nv),
            _p(d_amplitudes), _p(d_amplitude_flags), _i64(di.size), _i64(n_samp), _p(iv), _i64(iv.size),
            _p(stream)))

    def offset_accumulate(self, step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags, d_g2l,
                          d_zmap, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights, flag_index,
                          d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_samp, intervals, d_shared_flags,
                          n_shared_flags, shared_flag_mask, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        fi = self._small(flag_index, np.int32)
        ds = self._small(det_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_offset_accumulate_dev(
            _i64(step_length), _p(ao), _p(nv), _p(d_amplitudes), _p(d_amplitude_flags), _p(d_g2l), _p(d_zmap),
            _i64(n_pix_submap), _i64(nnz), _p(pi), _p(d_pixels), _p(wi), _p(d_weights), _p(fi), _p(d_det_flags),
            _i64(n_flag_samp), _p(ds), _u8(det_flag_mask), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size),
            _p(d_shared_flags), _i64(n_shared_flags), _u8(shared_flag_mask), _p(stream)))

    def offset_clean_accumulate(self, step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags, d_g2l,
                                d_zmap, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights, data_index,
                                d_signal, flag_index, d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_samp,
                                intervals, d_shared_flags, n_shared_flags, shared_flag_mask, stream=0):
        """zmap += A^T N^-1 (d - M a): the cleaned signal binned in one pass (toast_hip_offset_clean_accumulate_dev)."""
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        di = self._small(data_index, np.int32)
        fi = self._small(flag_index, np.int32)
        ds = self._small(det_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_offset_clean_accumulate_dev(
            _i64(step_length), _p(ao), _p(nv), _p(d_amplitudes), _p(d_amplitude_flags), _p(d_g2l), _p(d_zmap),
            _i64(n_pix_submap), _i64(nnz), _p(pi), _p(d_pixels), _p(wi), _p(d_weights), _p(di), _p(d_signal), _p(fi),
            _p(d_det_flags), _i64(n_flag_samp), _p(ds), _u8(det_flag_mask), _i64(pi.size), _i64(n_samp), _p(iv),
            _i64(iv.size), _p(d_shared_flags), _i64(n_shared_flags), _u8(shared_flag_mask), _p(stream)))

    def offset_scan_project(self, step_length, amp_offsets, n_amp_views, d_amps_in, d_amps_out, d_amplitude_flags,
                            d_g2l, d_map, n_pix_submap, nnz, pixel_index, d_pixels, weight_index, d_weights,
                            flag_index, d_flag_data, flag_mask, det_weights, n_samp, intervals, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        fi = None if flag_index is None else self._small(flag_index, np.int32)
        dw = self._small(det_weights, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_offset_scan_project_dev(
            _i64(step_length), _p(ao), _p(nv), _p(d_amps_in), _p(d_amps_out), _p(d_amplitude_flags), _p(d_g2l),
            _p(d_map), _i64(n_pix_submap), _i64(nnz), _p(pi), _p(d_pixels), _p(wi), _p(d_weights), _p(fi),
            _p(d_flag_data), _u8(flag_mask), _p(dw), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size),
            _p(stream)))

    def offset_pack_pointing(self, d_g2l, n_pix_submap, pixel_index, d_pixels, weight_index, d_weights, acc_flag_index,
                             d_det_flags, n_flag_samp, det_flag_mask, d_shared_flags, n_shared_flags, shared_flag_mask,
                             proj_flag_index, d_proj_flags, n_proj_flag_samp, proj_flag_mask, n_samp, intervals, d_key,
                             d_qu, d_cal, pair_words=True, stream=0):
        """(packable, pair_words) of toast_hip_offset_pack_pointing_dev; waits for the stream."""
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        fa = self._small(acc_flag_index if acc_flag_index is not None else np.zeros(pi.size), np.int32)
        fp = self._small(proj_flag_index if proj_flag_index is not None else np.zeros(pi.size), np.int32)
        iv = self._small(intervals, interval_dtype)
        ok, pair = C.c_int(0), C.c_int(0)
        _check(real_lib().toast_hip_offset_pack_pointing_dev(
            _p(d_g2l), _i64(n_pix_submap), _p(pi), _p(d_pixels), _p(wi), _p(d_weights), _p(fa), _p(d_det_flags),
            _i64(n_flag_samp), _u8(det_flag_mask), _p(d_shared_flags), _i64(n_shared_flags), _u8(shared_flag_mask),
            _p(fp), _p(d_proj_flags), _i64(n_proj_flag_samp), _u8(proj_flag_mask), _i64(pi.size), _i64(n_samp), _p(iv),
            _i64(iv.size), _p(d_key), _p(d_qu), _p(d_cal), C.byref(ok), C.byref(pair) if pair_words else None,
            _p(stream)))
        return bool(ok.value), bool(pair.value)

    def offset_pack_pointing_onepass(self, d_g2l, n_pix_submap, pixel_index, d_pixels, weight_index, d_weights,
                                     acc_flag_index, d_det_flags, n_flag_samp, det_flag_mask, d_shared_flags,
                                     n_shared_flags, shared_flag_mask, proj_flag_index, d_proj_flags, n_proj_flag_samp,
                                     proj_flag_mask, n_samp, intervals, d_key, d_qu, d_cal, d_corr, stream=0):
        """(packable, pair_words, pair_weights) of toast_hip_offset_pack_pointing_onepass_dev: the pack, the pair words
        and the pair weight sums in one sweep; waits for the stream."""
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        fa = self._small(acc_flag_index if acc_flag_index is not None else np.zeros(pi.size), np.int32)
        fp = self._small(proj_flag_index if proj_flag_index is not None else np.zeros(pi.size), np.int32)
        iv = self._small(intervals, interval_dtype)
        ok, pair, pw = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(real_lib().toast_hip_offset_pack_pointing_onepass_dev(
            _p(d_g2l), _i64(n_pix_submap), _p(pi), _p(d_pixels), _p(wi), _p(d_weights), _p(fa), _p(d_det_flags),
            _i64(n_flag_samp), _u8(det_flag_mask), _p(d_shared_flags), _i64(n_shared_flags), _u8(shared_flag_mask),
            _p(fp), _p(d_proj_flags), _i64(n_proj_flag_samp), _u8(proj_flag_mask), _i64(pi.size), _i64(n_samp), _p(iv),
            _i64(iv.size), _p(d_key), _p(d_qu), _p(d_cal), _p(d_corr), C.byref(ok), C.byref(pair), C.byref(pw),
            _p(stream)))
        return bool(ok.value), bool(pair.value), bool(pw.value)

    def offset_pack_pairs(self, d_key, n_det, n_samp, intervals, stream=0):
        """True when row 2b now holds one word per pair-sample (toast_hip_offset_pack_pairs_dev); waits for the stream."""
        iv = self._small(intervals, interval_dtype)
        pair = C.c_int(0)
        _check(real_lib().toast_hip_offset_pack_pairs_dev(_p(d_key), _i64(n_det), _i64(n_samp), _p(iv), _i64(iv.size),
                                                          C.byref(pair), _p(stream)))
        return bool(pair.value)

    def offset_pack_pair_weights(self, d_qu, d_corr, n_det, n_samp, intervals, stream=0):
        """True when d_corr now holds the pair sums q_a + q_b, u_a + u_b (float2 per pair-sample) from which the sweeps
        rebuild the partner's weights exactly (toast_hip_offset_pack_pair_weights_dev); waits for the stream."""
        iv = self._small(intervals, interval_dtype)
        ok = C.c_int(0)
        _check(real_lib().toast_hip_offset_pack_pair_weights_dev(_p(d_qu), _p(d_corr), _i64(n_det), _i64(n_samp), _p(iv),
                                                                 _i64(iv.size), C.byref(ok), _p(stream)))
        return bool(ok.value)

    def offset_accumulate_packed(self, step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags, d_zmap,
                                 d_key, d_qu, d_cal, det_scale, n_samp, intervals, pair_words=False, pair_corr=0,
                                 stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        ds = self._small(det_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_offset_accumulate_packed_dev(
            _i64(step_length), _p(ao), _p(nv), _p(d_amplitudes), _p(d_amplitude_flags), _p(d_zmap), _p(d_key), _p(d_qu),
            _p(d_cal), _p(ds), C.c_int(1 if pair_words else 0), _p(pair_corr), _i64(ao.size), _i64(n_samp), _p(iv),
            _i64(iv.size), _p(stream)))

    def offset_scan_project_packed(self, step_length, amp_offsets, n_amp_views, d_amps_in, d_amps_out,
                                   d_amplitude_flags, d_map, d_key, d_qu, d_cal, det_weights, n_samp, intervals,
                                   pair_words=False, pair_corr=0, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        dw = self._small(det_weights, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_offset_scan_project_packed_dev(
            _i64(step_length), _p(ao), _p(nv), _p(d_amps_in), _p(d_amps_out), _p(d_amplitude_flags), _p(d_map), _p(d_key),
            _p(d_qu), _p(d_cal), _p(dw), C.c_int(1 if pair_words else 0), _p(pair_corr), _i64(ao.size), _i64(n_samp),
            _p(iv), _i64(iv.size), _p(stream)))

    def offset_scan_project_signal(self, step_length, amp_offsets, n_amp_views, signal_index, d_signal, d_amps_out,
                                   d_amplitude_flags, d_g2l, d_map, n_pix_submap, nnz, pixel_index, d_pixels,
                                   weight_index, d_weights, flag_index, d_flag_data, flag_mask, det_weights, n_samp,
                                   intervals, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        si = self._small(signal_index, np.int32)
        pi = self._small(pixel_index, np.int32)
        wi = self._small(weight_index, np.int32)
        fi = None if flag_index is None else self._small(flag_index, np.int32)
        dw = self._small(det_weights, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_offset_scan_project_signal_dev(
            _i64(step_length), _p(ao), _p(nv), _p(si), _p(d_signal), _p(d_amps_out), _p(d_amplitude_flags), _p(d_g2l),
            _p(d_map), _i64(n_pix_submap), _i64(nnz), _p(pi), _p(d_pixels), _p(wi), _p(d_weights), _p(fi),
            _p(d_flag_data), _u8(flag_mask), _p(dw), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size),
            _p(stream)))

    # -- pointing on the fly (otf_kernels.hip)
    def otf_build_noise_weighted(self, pt, d_g2l, d_zmap, n_pix_submap, data_index, d_det_data, flag_index,
                                 d_det_flags, n_flag_samp, det_scale, det_flag_mask, n_samp, intervals,
                                 d_shared_flags=0, n_shared_flags=0, shared_flag_mask=0, stream=0):
        di = self._small(data_index, np.int32)
        fi = self._small(flag_index, np.int32)
        ds = self._small(det_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_build_noise_weighted_dev(
            C.byref(pt), _p(d_g2l), _p(d_zmap), _i64(n_pix_submap), _p(di), _p(d_det_data), _p(fi),
            _p(d_det_flags), _i64(n_flag_samp), _p(ds), _u8(det_flag_mask), _i64(di.size), _i64(n_samp), _p(iv),
            _i64(iv.size), _p(d_shared_flags), _i64(n_shared_flags), _u8(shared_flag_mask), _p(stream)))

    def compact_pixels(self, d_g2l, n_pix_submap, n_local_submap, pixel_index, d_pixels, compact_index,
                       d_compact_pixels, n_samp, intervals, stream=0):
        pi = self._small(pixel_index, np.int32)
        ci = self._small(compact_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_compact_pixels_dev(
            _p(d_g2l), _i64(n_pix_submap), _i64(n_local_submap), _p(pi), _p(d_pixels), _p(ci),
            _p(d_compact_pixels), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def otf_pixels_healpix(self, pt, pixel_index, d_pixels, n_samp, intervals, d_hit_submaps, n_submap, n_pix_submap,
                           stream=0):
        pi = self._small(pixel_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_pixels_healpix_dev(
            C.byref(pt), _p(pi), _p(d_pixels), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(d_hit_submaps),
            _i64(n_submap), _i64(n_pix_submap), _p(stream)))

    def otf_stokes_weights(self, pt, weight_index, d_weights, n_samp, intervals, stream=0):
        wi = self._small(weight_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_stokes_weights_dev(
            C.byref(pt), _p(wi), _p(d_weights), _i64(wi.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def hwp_table(self, d_hwp, n_samp, d_table, stream=0):
        _check(lib().toast_hip_hwp_table_dev(_p(d_hwp), _i64(n_samp), _p(d_table), _p(stream)))

    def otf_compact_pixels(self, pt, d_g2l, n_pix_submap, n_local_submap, compact_index, d_compact_pixels, n_samp,
                           intervals, stream=0):
        ci = self._small(compact_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_compact_pixels_dev(
            C.byref(pt), _p(d_g2l), _i64(n_pix_submap), _i64(n_local_submap), _p(ci), _p(d_compact_pixels),
            _i64(ci.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def otf_scan_map(self, pt, d_g2l, d_map, n_pix_submap, d_det_data, data_index, n_samp, intervals,
                     data_scale=1.0, should_zero=False, should_subtract=False, det_weights=None, stream=0):
        di = self._small(data_index, np.int32)
        dw = None if det_weights is None else self._small(det_weights, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_scan_map_dev(
            C.byref(pt), _p(d_g2l), _p(d_map), _i64(n_pix_submap), _p(d_det_data), _p(di), _i64(di.size),
            _i64(n_samp), _p(iv), _i64(iv.size), C.c_double(data_scale), _int(should_zero), _int(should_subtract),
            _p(dw), _p(stream)))

    def otf_offset_accumulate(self, pt, step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags,
                              d_g2l, d_zmap, n_pix_submap, flag_index, d_det_flags, n_flag_samp, det_scale,
                              det_flag_mask, n_samp, intervals, d_shared_flags=0, n_shared_flags=0,
                              shared_flag_mask=0, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        fi = self._small(flag_index, np.int32)
        ds = self._small(det_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_offset_accumulate_dev(
            C.byref(pt), _i64(step_length), _p(ao), _p(nv), _p(d_amplitudes), _p(d_amplitude_flags), _p(d_g2l),
            _p(d_zmap), _i64(n_pix_submap), _p(fi), _p(d_det_flags), _i64(n_flag_samp), _p(ds),
            _u8(det_flag_mask), _i64(ao.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(d_shared_flags),
            _i64(n_shared_flags), _u8(shared_flag_mask), _p(stream)))

    def otf_offset_clean_accumulate(self, pt, step_length, amp_offsets, n_amp_views, d_amplitudes, d_amplitude_flags,
                                    d_g2l, d_zmap, n_pix_submap, data_index, d_signal, flag_index, d_det_flags,
                                    n_flag_samp, det_scale, det_flag_mask, n_samp, intervals, d_shared_flags=0,
                                    n_shared_flags=0, shared_flag_mask=0, stream=0):
        """zmap += P^T N^-1 (d - M a) with on-the-fly pointing (toast_hip_otf_offset_clean_accumulate_dev)."""
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        di = self._small(data_index, np.int32)
        fi = self._small(flag_index, np.int32)
        ds = self._small(det_scale, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_offset_clean_accumulate_dev(
            C.byref(pt), _i64(step_length), _p(ao), _p(nv), _p(d_amplitudes), _p(d_amplitude_flags), _p(d_g2l),
            _p(d_zmap), _i64(n_pix_submap), _p(di), _p(d_signal), _p(fi), _p(d_det_flags), _i64(n_flag_samp), _p(ds),
            _u8(det_flag_mask), _i64(ao.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(d_shared_flags),
            _i64(n_shared_flags), _u8(shared_flag_mask), _p(stream)))

    def otf_offset_scan_project(self, pt, step_length, amp_offsets, n_amp_views, d_amps_in, d_amps_out,
                                d_amplitude_flags, d_g2l, d_map, n_pix_submap, flag_index, d_flag_data,
                                n_flag_samp, flag_mask, det_weights, n_samp, intervals, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        fi = None if flag_index is None else self._small(flag_index, np.int32)
        dw = self._small(det_weights, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_offset_scan_project_dev(
            C.byref(pt), _i64(step_length), _p(ao), _p(nv), _p(d_amps_in), _p(d_amps_out), _p(d_amplitude_flags),
            _p(d_g2l), _p(d_map), _i64(n_pix_submap), _p(fi), _p(d_flag_data), _i64(n_flag_samp), _u8(flag_mask),
            _p(dw), _i64(ao.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def otf_offset_scan_project_signal(self, pt, step_length, amp_offsets, n_amp_views, signal_index, d_signal,
                                       d_amps_out, d_amplitude_flags, d_g2l, d_map, n_pix_submap, flag_index,
                                       d_flag_data, n_flag_samp, flag_mask, det_weights, n_samp, intervals, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        si = self._small(signal_index, np.int32)
        fi = None if flag_index is None else self._small(flag_index, np.int32)
        dw = self._small(det_weights, np.float64)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_otf_offset_scan_project_signal_dev(
            C.byref(pt), _i64(step_length), _p(ao), _p(nv), _p(si), _p(d_signal), _p(d_amps_out),
            _p(d_amplitude_flags), _p(d_g2l), _p(d_map), _i64(n_pix_submap), _p(fi), _p(d_flag_data),
            _i64(n_flag_samp), _u8(flag_mask), _p(dw), _i64(ao.size), _i64(n_samp), _p(iv), _i64(iv.size),
            _p(stream)))

    def scan_mask(self, d_g2l, d_mask, n_pix_submap, mask_bits, flag_value, pixel_index, d_pixels, flag_index,
                  d_det_flags, n_samp, intervals, stream=0):
        pi = self._small(pixel_index, np.int32)
        fi = self._small(flag_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_scan_mask_dev(
            _p(d_g2l), _p(d_mask), _i64(n_pix_submap), _u8(mask_bits), _u8(flag_value), _p(pi), _p(d_pixels), _p(fi),
            _p(d_det_flags), _i64(pi.size), _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def offset_count_flagged(self, step_length, amp_offsets, n_amp_views, d_counts, flag_index, d_det_flags, flag_mask,
                             n_samp, intervals, stream=0):
        ao = self._small(amp_offsets, np.int64)
        nv = self._small(n_amp_views, np.int64)
        fi = self._small(flag_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_offset_count_flagged_dev(
            _i64(step_length), _p(ao), _p(nv), _p(d_counts), _p(fi), _p(d_det_flags), _u8(flag_mask), _i64(ao.size),
            _i64(n_samp), _p(iv), _i64(iv.size), _p(stream)))

    def threshold_mask(self, n, d_value, threshold, bit, d_mask, stream=0):
        _check(lib().toast_hip_threshold_mask_dev(_i64(n), _p(d_value), C.c_double(float(threshold)), _u8(bit),
                                                  _p(d_mask), _p(stream)))

    def offset_variance(self, amp_offsets, det_weight, amp_len, d_n_bad, good_fraction, d_amp_flags, d_variance,
                        stream=0):
        ao = self._small(amp_offsets, np.int64)
        dw = self._small(det_weight, np.float64)
        al = self._small(amp_len, np.int64)
        _check(lib().toast_hip_offset_variance_dev(
            _i64(ao.size), _i64(al.size), _p(ao), _p(dw), _p(al), _p(d_n_bad), C.c_double(float(good_fraction)),
            _p(d_amp_flags), _p(d_variance), _p(stream)))

    def offset_convolve(self, n_amp, n_seg, d_seg_start, max_segment_len, d_filt_start, d_filt_len, max_filter_len,
                        d_filters, d_amp_in, d_amp_flags, d_amp_out, accumulate, stream=0):
        _check(lib().toast_hip_template_offset_convolve_dev(
            _i64(n_amp), _i64(n_seg), _p(d_seg_start), _i64(max_segment_len), _p(d_filt_start), _p(d_filt_len),
            _i64(max_filter_len), _p(d_filters), _p(d_amp_in), _p(d_amp_flags), _p(d_amp_out),
            C.c_int(1 if accumulate else 0), _p(stream)))

    def offset_banded_solve(self, n_seg, d_seg_start, d_band_width, max_band_width, d_band_start, d_forward,
                            d_backward, d_amp_in, d_amp_flags, d_amp_out, stream=0):
        _check(lib().toast_hip_template_offset_banded_solve_dev(
            _i64(n_seg), _p(d_seg_start), _p(d_band_width), C.c_int32(int(max_band_width)), _p(d_band_start),
            _p(d_forward), _p(d_backward), _p(d_amp_in), _p(d_amp_flags), _p(d_amp_out), _p(stream)))

    def offset_banded_cholesky(self, n_seg, d_seg_start, d_band_width, max_band_width, d_band_start, d_toep_start,
                               d_toep_len, d_toeplitz, d_diag_scale, d_offset_var, d_forward, d_backward, d_status,
                               stream=0):
        _check(lib().toast_hip_template_offset_banded_cholesky_dev(
            _i64(n_seg), _p(d_seg_start), _p(d_band_width), C.c_int32(int(max_band_width)), _p(d_band_start),
            _p(d_toep_start), _p(d_toep_len), _p(d_toeplitz), _p(d_diag_scale), _p(d_offset_var), _p(d_forward),
            _p(d_backward), _p(d_status), _p(stream)))

    def legendre_templates(self, d_x, n_samp, start_order, stop_order, d_templates, stream=0):
        _check(lib().toast_hip_legendre_templates_dev(_p(d_x), _i64(n_samp), _i64(start_order), _i64(stop_order),
                                                      _p(d_templates), _p(stream)))

    def template_select(self, d_src, d_key, value, keep_equal, n_samp, d_out, stream=0):
        _check(lib().toast_hip_template_select_dev(_p(d_src), _p(d_key), C.c_int32(int(value)),
                                                   C.c_int(1 if keep_equal else 0), _i64(n_samp), _p(d_out), _p(stream)))

    def template_fit(self, d_templates, n_template, n_samp, signal_index, d_signal, flag_index, d_det_flags,
                     det_flag_mask, d_shared_flags, shared_flag_mask, d_proj, d_gram_common, d_gram_flagged, d_n_flagged,
                     stream=0):
        si = self._small(signal_index, np.int32)
        fi = self._small(flag_index if flag_index is not None else np.zeros(si.size), np.int32)
        _check(lib().toast_hip_template_fit_dev(
            _p(d_templates), _i64(n_template), _i64(n_samp), _p(si), _p(d_signal), _p(fi), _p(d_det_flags),
            _u8(det_flag_mask), _p(d_shared_flags), _u8(shared_flag_mask), _i64(si.size), _p(d_proj), _p(d_gram_common),
            _p(d_gram_flagged), _p(d_n_flagged), _p(stream)))

    def template_subtract(self, d_templates, n_template, first_template, n_samp, signal_index, d_signal, d_coeff,
                          stream=0):
        si = self._small(signal_index, np.int32)
        _check(lib().toast_hip_template_subtract_dev(
            _p(d_templates), _i64(n_template), _i64(first_template), _i64(n_samp), _p(si), _p(d_signal), _p(d_coeff),
            _i64(si.size), _p(stream)))

    def combine_flags(self, d_out, out_index, d_det_flags, n_flag_samp, flag_index, det_flag_mask, d_shared_flags,
                      n_shared_flags, shared_flag_mask, n_samp, intervals, n_out_rows=0, outside_value=-1, stream=0):
        oi = self._small(out_index, np.int32)
        fi = self._small(flag_index, np.int32)
        iv = self._small(intervals, interval_dtype)
        _check(lib().toast_hip_combine_flags_dev(
            _p(d_out), _p(oi), _p(d_det_flags), _i64(n_flag_samp), _p(fi), _u8(det_flag_mask), _p(d_shared_flags),
            _i64(n_shared_flags), _u8(shared_flag_mask), _i64(oi.size), _i64(n_samp), _p(iv), _i64(iv.size),
            _i64(n_out_rows), C.c_int(int(outside_value)), _p(stream)))

    def cov_eigendecompose_diag(self, n_sub, subsize, nnz, d_data, d_cond, threshold, invert, stream=0):
        _check(lib().toast_hip_cov_eigendecompose_diag_dev(_i64(n_sub), _i64(subsize), _i64(nnz), _p(d_data), _p(d_cond),
                                                           C.c_double(float(threshold)), C.c_int(1 if invert else 0),
                                                           _p(stream)))

    def memset(self, d_dst, value, nbytes, stream=0):
        _check(lib().toast_hip_memset_dev(_p(d_dst), C.c_int(int(value)), C.c_size_t(int(nbytes)), _p(stream)))

    def block_move(self, d_dst, d_src, block_bytes, dst_index, src_index, stream=0):
        """Blocks of ``block_bytes``: block ``dst_index[i]`` of ``d_dst`` <- block ``src_index[i]`` of ``d_src``."""
        di = self._small(dst_index, np.int64)
        si = self._small(src_index, np.int64)
        if di.size != si.size:
            raise RuntimeError("block_move: one source block per destination block")
        _check(lib().toast_hip_block_move_dev(_p(d_dst), _p(d_src), _i64(di.size), _i64(block_bytes), _p(di), _p(si),
                                              _p(stream)))

    def copy(self, d_dst, d_src, nbytes, stream=0):
        _check(lib().toast_hip_copy_dev(_p(d_dst), _p(d_src), C.c_size_t(int(nbytes)), _p(stream)))

    def vec_axpby(self, n, a, d_x, b, d_y, stream=0):
        _check(lib().toast_hip_vec_axpby_dev(_i64(n), C.c_double(a), _p(d_x), C.c_double(b), _p(d_y), _p(stream)))

    def vec_dot(self, n, d_x, d_y, d_fx=0, d_fy=0, stream=0):
        out = C.c_double(0.0)
        _check(lib().toast_hip_vec_dot_dev(_i64(n), _p(d_x), _p(d_y), _p(d_fx), _p(d_fy), C.byref(out), _p(stream)))
        return out.value

    # ---- PCG with device-side scalars (toast_hip_pcg_*)
    PCG_ONE, PCG_ALPHA, PCG_NEG_ALPHA, PCG_BETA, PCG_LIVE = 0, 1, 2, 3, 4
    PCG_DONE = {0: "running", 1: "converged", 2: "stalled", 3: "not finite", 4: "iteration limit"}

    class PcgStatus(C.Structure):
        _fields_ = [("iteration", C.c_int64), ("done", C.c_int64), ("n_history", C.c_int64), ("relative", C.c_double),
                    ("sqsum", C.c_double)]

    def pcg_state_bytes(self, n_iter_max):
        out = C.c_size_t(0)
        _check(real_lib().toast_hip_pcg_state_bytes(_i64(n_iter_max), C.byref(out)))
        return int(out.value)

    def pcg_init(self, d_state, sqsum_init, delta, convergence, n_iter_min, n_iter_max, stream=0):
        _check(lib().toast_hip_pcg_init_dev(_p(d_state), C.c_double(float(sqsum_init)), C.c_double(float(delta)),
                                            C.c_double(float(convergence)), _i64(n_iter_min), _i64(n_iter_max),
                                            _p(stream)))

    def pcg_dot(self, d_state, n, d_x, d_y, d_fx=0, d_fy=0, accumulate=False, stage=0, stream=0):
        _check(lib().toast_hip_pcg_dot_dev(_p(d_state), _i64(n), _p(d_x), _p(d_y), _p(d_fx), _p(d_fy),
                                           C.c_int(1 if accumulate else 0), C.c_int(int(stage)), _p(stream)))

    def pcg_step(self, d_state, n, d_proposal, d_result, d_lhs_out, d_residual, stream=0):
        _check(lib().toast_hip_pcg_step_dev(_p(d_state), _i64(n), _p(d_proposal), _p(d_result), _p(d_lhs_out),
                                            _p(d_residual), _p(stream)))

    def pcg_step_dot(self, d_state, n, d_proposal, d_result, d_lhs_out, d_residual, d_flags=0, accumulate=False,
                     stage=0, stream=0):
        _check(lib().toast_hip_pcg_step_dot_dev(_p(d_state), _i64(n), _p(d_proposal), _p(d_result), _p(d_lhs_out),
                                                _p(d_residual), _p(d_flags), C.c_int(1 if accumulate else 0),
                                                C.c_int(int(stage)), _p(stream)))

    def pcg_precond_diag_dot(self, d_state, n, d_var, d_residual, d_flags_residual, d_out, d_flags_out=0,
                             accumulate=False, stage=0, stream=0):
        _check(lib().toast_hip_pcg_precond_diag_dot_dev(_p(d_state), _i64(n), _p(d_var), _p(d_residual),
                                                        _p(d_flags_residual), _p(d_out), _p(d_flags_out),
                                                        C.c_int(1 if accumulate else 0), C.c_int(int(stage)),
                                                        _p(stream)))

    def pcg_stage(self, d_state, stage, allreduce=False, stream=0):
        _check(lib().toast_hip_pcg_stage_dev(_p(d_state), C.c_int(int(stage)), C.c_int(1 if allreduce else 0),
                                             _p(stream)))

    def pcg_axpby(self, d_state, n, a_sel, d_x, b_sel, d_y, stream=0):
        _check(lib().toast_hip_pcg_axpby_dev(_p(d_state), _i64(n), C.c_int(int(a_sel)), _p(d_x), C.c_int(int(b_sel)),
                                             _p(d_y), _p(stream)))

    def pcg_status(self, d_state, lag=1, stream=0):
        st = self.PcgStatus()
        _check(real_lib().toast_hip_pcg_status_dev(_p(d_state), C.c_int(int(lag)), C.byref(st), _p(stream)))
        return st

    def pcg_history(self, d_state, capacity, stream=0):
        hist = np.zeros(max(int(capacity), 1), dtype=np.float64)
        st = self.PcgStatus()
        _check(real_lib().toast_hip_pcg_history_dev(_p(d_state), _p(hist), _i64(capacity), C.byref(st), _p(stream)))
        return hist[:min(int(st.n_history), int(capacity))].copy(), st

    # ---- the process' RCCL communicator (toast_hip_comm_*): collectives on the kernels' stream
    COMM_DTYPES = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.int64): 2, np.dtype(np.int32): 3,
                   np.dtype(np.uint8): 4}
    COMM_OPS = {"sum": 0, "max": 1, "min": 2}

    def comm_available(self):
        return bool(real_lib().toast_hip_comm_available())

    def comm_unique_id(self):
        buf = (C.c_ubyte * 128)()
        _check(real_lib().toast_hip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, n_ranks, rank):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id has 128 bytes")
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        _check(real_lib().toast_hip_comm_init(buf, C.c_int(int(n_ranks)), C.c_int(int(rank))))

    def comm_info(self):
        n, r, v = C.c_int(0), C.c_int(-1), C.c_int(0)
        _check(real_lib().toast_hip_comm_info(C.byref(n), C.byref(r), C.byref(v)))
        return int(n.value), int(r.value), int(v.value)

    def comm_destroy(self):
        _check(real_lib().toast_hip_comm_destroy())

    def comm_allreduce(self, d_buf, count, dtype, op="sum", stream=0):
        _check(lib().toast_hip_comm_allreduce_dev(_p(d_buf), _i64(count), C.c_int(self.COMM_DTYPES[np.dtype(dtype)]),
                                                  C.c_int(self.COMM_OPS[op]), _p(stream)))

    def comm_broadcast(self, d_buf, count, dtype, root=0, stream=0):
        _check(lib().toast_hip_comm_broadcast_dev(_p(d_buf), _i64(count), C.c_int(self.COMM_DTYPES[np.dtype(dtype)]),
                                                  C.c_int(int(root)), _p(stream)))

    def comm_reduce_scatter(self, d_send, d_recv, recv_count, dtype, op="sum", stream=0):
        _check(lib().toast_hip_comm_reduce_scatter_dev(_p(d_send), _p(d_recv), _i64(recv_count),
                                                       C.c_int(self.COMM_DTYPES[np.dtype(dtype)]),
                                                       C.c_int(self.COMM_OPS[op]), _p(stream)))

    def comm_all_gather(self, d_send, d_recv, send_count, dtype, stream=0):
        _check(lib().toast_hip_comm_all_gather_dev(_p(d_send), _p(d_recv), _i64(send_count),
                                                   C.c_int(self.COMM_DTYPES[np.dtype(dtype)]), _p(stream)))

    def comm_pixel_shard(self, n_px):
        first, count = C.c_int64(0), C.c_int64(0)
        _check(real_lib().toast_hip_comm_pixel_shard(_i64(n_px), C.byref(first), C.byref(count)))
        return int(first.value), int(count.value)

    def comm_map_reduce_apply(self, n_px, nnz, d_cov, d_map, reduce=True, stream=0):
        _check(lib().toast_hip_comm_map_reduce_apply_dev(_i64(n_px), _i64(nnz), _p(d_cov), _p(d_map),
                                                         C.c_int(1 if reduce else 0), _p(stream)))

    def comm_set_mode(self, mode):
        """'owner' | 'allreduce' | 'peer' | 'peer:flags': how comm_map_reduce_apply works (toast_hip_comm_set_mode; collective:
        every rank the same)."""
        _check(real_lib().toast_hip_comm_set_mode(str(mode).encode()))

    def comm_set_peer_width(self, nbytes):
        """8 or 16 bytes per lane and access in the 'peer' exchange kernels (toast_hip_comm_set_peer_width)."""
        _check(real_lib().toast_hip_comm_set_peer_width(C.c_int(int(nbytes))))

    def comm_check(self, stream=0):
        """Synchronise the stream and raise a pending 'peer:flags' time-out (toast_hip_comm_check)."""
        _check(real_lib().toast_hip_comm_check(_p(stream)))

    def comm_peer_stats(self):
        """(reductions, establishments, exchange_bytes) of mode 'peer' (toast_hip_comm_peer_stats)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(real_lib().toast_hip_comm_peer_stats(C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def comm_peer_mem(self):
        """'fine' / 'coarse': the memory kind of the 'peer' exchange buffer (toast_hip_comm_peer_mem)."""
        a = C.c_int(0)
        _check(real_lib().toast_hip_comm_peer_mem(C.byref(a)))
        return "fine" if a.value else "coarse"

    def comm_get_mode(self):
        buf = C.create_string_buffer(32)
        _check(real_lib().toast_hip_comm_get_mode(buf, C.c_size_t(32)))
        return buf.value.decode()

    def comm_cov_invert(self, n_px, nnz, d_cov, d_rcond, threshold, invert=True, stream=0):
        _check(lib().toast_hip_comm_cov_invert_dev(_i64(n_px), _i64(nnz), _p(d_cov), _p(d_rcond),
                                                   C.c_double(float(threshold)), C.c_int(1 if invert else 0), _p(stream)))

    def comm_cov_mult(self, n_px, nnz, d_cov1, d_cov2, stream=0):
        _check(lib().toast_hip_comm_cov_mult_dev(_i64(n_px), _i64(nnz), _p(d_cov1), _p(d_cov2), _p(stream)))

    def test_math(self, op, n, d_a, d_b, d_out, stream=0):
        _check(lib().toast_hip_test_math_dev(C.c_int(op), _i64(n), _p(d_a), _p(d_b), _p(d_out), _p(stream)))


dev = _Dev()
