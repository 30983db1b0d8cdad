"""ctypes front end of the CPU oracle (oracle/libtoast_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by ``toast_amd``.

The functions take NumPy arrays with exactly the argument order of the reference's
``toast._libtoast`` bindings, minus the trailing ``use_accel``, so a
parity test can call the reference build (``oracle/_ref``), this restatement and the HIP
library with one argument tuple.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtoast_oracle.so")

#: NumPy dtype of an interval; reference: src/toast/intervals.py:26-45,
#: src/toast/_libtoast/intervals.hpp:10-15.
interval_dtype = np.dtype(
    {
        "names": ["start", "stop", "first", "last"],
        "formats": ["d", "d", "q", "q"],
        "offsets": [0, 8, 16, 24],
    }
)


def build(force=False):
    """Compile the restatement (and, when /root/reference exists, oracle/_ref)."""
    if force or not os.path.isfile(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "toast_oracle.cpp")
    ):
        subprocess.check_call(["make", "-C", _HERE, "libtoast_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def load_ref():
    """Import the reference's own compiled bindings from oracle/_ref (or None)."""
    import importlib.util
    import glob

    hits = glob.glob(os.path.join(_HERE, "_ref", "_toast_ref*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("_toast_ref", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # OmpManager must be told it has no device before any kernel runs
    # (reference: src/toast/_libtoast/accelerator.cpp:308-317).
    mod.accel_assign_device(1, 0, 1.0, True)
    return mod


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i64(x):
    return C.c_int64(int(x))


def _chk(a, dtype, ndim=None):
    assert isinstance(a, np.ndarray), type(a)
    assert a.dtype == dtype, (a.dtype, dtype)
    assert a.flags["C_CONTIGUOUS"]
    if ndim is not None:
        assert a.ndim == ndim, (a.ndim, ndim)
    return a


# ------------------------------------------------------------------ healpix utilities
def healpix_ang2pix(nside, theta, phi, nest=True):
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    out = np.empty(theta.shape, dtype=np.int64)
    lib().oracle_healpix_ang2pix(_i64(nside), C.c_int(bool(nest)), _i64(theta.size), _p(theta), _p(phi), _p(out))
    return out


def healpix_vec2pix(nside, vec, nest=True):
    vec = np.ascontiguousarray(vec, dtype=np.float64).reshape(-1, 3)
    out = np.empty(vec.shape[0], dtype=np.int64)
    lib().oracle_healpix_vec2pix(_i64(nside), C.c_int(bool(nest)), _i64(vec.shape[0]), _p(vec), _p(out))
    return out


def healpix_ring2nest(nside, pix):
    pix = np.ascontiguousarray(pix, dtype=np.int64)
    out = np.empty_like(pix)
    lib().oracle_healpix_ring2nest(_i64(nside), _i64(pix.size), _p(pix), _p(out))
    return out


def healpix_nest2ring(nside, pix):
    pix = np.ascontiguousarray(pix, dtype=np.int64)
    out = np.empty_like(pix)
    lib().oracle_healpix_nest2ring(_i64(nside), _i64(pix.size), _p(pix), _p(out))
    return out


# ------------------------------------------------------------------ hot-path kernels
def pointing_detector(focalplane, boresight, quat_index, quats, intervals, shared_flags, shared_flag_mask):
    _chk(focalplane, np.float64, 2)
    _chk(boresight, np.float64, 2)
    _chk(quat_index, np.int32, 1)
    _chk(quats, np.float64, 3)
    _chk(intervals, interval_dtype, 1)
    _chk(shared_flags, np.uint8, 1)
    n_samp = boresight.shape[0]
    lib().oracle_pointing_detector(
        _p(focalplane), _p(boresight), _p(quat_index), _p(quats), _p(intervals),
        _i64(intervals.size), _p(shared_flags), _i64(shared_flags.size),
        C.c_uint8(shared_flag_mask), _i64(quat_index.size), _i64(n_samp),
    )


def pixels_healpix(quat_index, quats, shared_flags, shared_flag_mask, pixel_index, pixels,
                   intervals, hit_submaps, n_pix_submap, nside, nest):
    _chk(quat_index, np.int32, 1)
    _chk(quats, np.float64, 3)
    _chk(shared_flags, np.uint8, 1)
    _chk(pixel_index, np.int32, 1)
    _chk(pixels, np.int64, 2)
    _chk(intervals, interval_dtype, 1)
    _chk(hit_submaps, np.uint8, 1)
    n_samp = pixels.shape[1]
    lib().oracle_pixels_healpix(
        _p(quat_index), _p(quats), _p(shared_flags), _i64(shared_flags.size),
        C.c_uint8(shared_flag_mask), _p(pixel_index), _p(pixels), _p(intervals),
        _i64(intervals.size), _p(hit_submaps), _i64(n_pix_submap), _i64(nside),
        C.c_int(bool(nest)), _i64(quat_index.size), _i64(n_samp),
    )


def stokes_weights_IQU(quat_index, quats, weight_index, weights, hwp, intervals, epsilon, gamma, cal, IAU):
    _chk(quats, np.float64, 3)
    _chk(weights, np.float64, 3)
    _chk(hwp, np.float64, 1)
    n_samp = weights.shape[1]
    lib().oracle_stokes_weights_IQU(
        _p(_chk(quat_index, np.int32, 1)), _p(quats), _p(_chk(weight_index, np.int32, 1)),
        _p(weights), _p(hwp), _i64(hwp.size), _p(_chk(intervals, interval_dtype, 1)),
        _i64(intervals.size), _p(_chk(epsilon, np.float64, 1)), _p(_chk(gamma, np.float64, 1)),
        _p(_chk(cal, np.float64, 1)), C.c_int(bool(IAU)), _i64(quat_index.size), _i64(n_samp),
    )


def stokes_weights_I(weight_index, weights, intervals, cal):
    _chk(weights, np.float64, 2)
    lib().oracle_stokes_weights_I(
        _p(_chk(weight_index, np.int32, 1)), _p(weights), _p(_chk(intervals, interval_dtype, 1)),
        _i64(intervals.size), _p(_chk(cal, np.float64, 1)), _i64(weight_index.size),
        _i64(weights.shape[1]),
    )


_SCAN = {
    np.dtype(np.float64): "oracle_scan_map_f64",
    np.dtype(np.float32): "oracle_scan_map_f32",
    np.dtype(np.int64): "oracle_scan_map_i64",
    np.dtype(np.int32): "oracle_scan_map_i32",
}


def scan_map(global2local, n_pix_submap, mapdata, det_data, data_index, pixels, pixel_index,
             weights, weight_index, intervals, data_scale, should_zero, should_subtract, should_scale):
    """ops_scan_map_<dtype>; the map dtype picks the instantiation."""
    _chk(global2local, np.int64, 1)
    _chk(det_data, np.float64, 2)
    _chk(pixels, np.int64, 2)
    _chk(weights, np.float64)
    assert mapdata.ndim == 3 and mapdata.flags["C_CONTIGUOUS"]
    nnz = 1 if weights.ndim == 2 else weights.shape[2]
    assert mapdata.shape[1] == n_pix_submap and mapdata.shape[2] == nnz
    n_samp = pixels.shape[1]
    getattr(lib(), _SCAN[mapdata.dtype])(
        _p(global2local), _i64(n_pix_submap), _p(mapdata), _p(det_data),
        _p(_chk(data_index, np.int32, 1)), _p(pixels), _p(_chk(pixel_index, np.int32, 1)),
        _p(weights), _p(_chk(weight_index, np.int32, 1)), _i64(nnz),
        _p(_chk(intervals, interval_dtype, 1)), _i64(intervals.size), _i64(pixel_index.size),
        _i64(n_samp), C.c_double(data_scale), C.c_int(bool(should_zero)),
        C.c_int(bool(should_subtract)), C.c_int(bool(should_scale)),
    )


def build_noise_weighted(global2local, zmap, pixel_index, pixels, weight_index, weights, data_index,
                         det_data, flag_index, det_flags, det_scale, det_flag_mask, intervals,
                         shared_flags, shared_flag_mask):
    _chk(global2local, np.int64, 1)
    _chk(zmap, np.float64, 3)
    _chk(pixels, np.int64, 2)
    _chk(weights, np.float64)
    _chk(det_data, np.float64, 2)
    _chk(det_flags, np.uint8, 2)
    _chk(shared_flags, np.uint8, 1)
    nnz = 1 if weights.ndim == 2 else weights.shape[2]
    assert zmap.shape[2] == nnz
    n_samp = pixels.shape[1]
    lib().oracle_build_noise_weighted(
        _p(global2local), _p(zmap), _i64(zmap.shape[1]), _i64(nnz),
        _p(_chk(pixel_index, np.int32, 1)), _p(pixels), _p(_chk(weight_index, np.int32, 1)),
        _p(weights), _p(_chk(data_index, np.int32, 1)), _p(det_data),
        _p(_chk(flag_index, np.int32, 1)), _p(det_flags), _i64(det_flags.shape[1]),
        _p(_chk(det_scale, np.float64, 1)), C.c_uint8(det_flag_mask),
        _p(_chk(intervals, interval_dtype, 1)), _i64(intervals.size), _p(shared_flags),
        _i64(shared_flags.size), C.c_uint8(shared_flag_mask), _i64(pixel_index.size), _i64(n_samp),
    )


def noise_weight(det_data, data_index, intervals, detector_weights):
    _chk(det_data, np.float64, 2)
    lib().oracle_noise_weight(
        _p(det_data), _p(_chk(data_index, np.int32, 1)), _p(_chk(intervals, interval_dtype, 1)),
        _i64(intervals.size), _p(_chk(detector_weights, np.float64, 1)), _i64(data_index.size),
        _i64(det_data.shape[1]),
    )


def template_offset_add_to_signal(step_length, amp_offset, n_amp_views, amplitudes, amplitude_flags,
                                  data_index, det_data, intervals):
    _chk(det_data, np.float64, 2)
    lib().oracle_offset_add_to_signal(
        _i64(step_length), _i64(amp_offset), _p(_chk(n_amp_views, np.int64, 1)),
        _p(_chk(amplitudes, np.float64, 1)), _p(_chk(amplitude_flags, np.uint8, 1)),
        C.c_int32(data_index), _p(det_data), _p(_chk(intervals, interval_dtype, 1)),
        _i64(intervals.size), _i64(det_data.shape[1]),
    )


def template_offset_project_signal(data_index, det_data, flag_index, flag_data, flag_mask, step_length,
                                   amp_offset, n_amp_views, amplitudes, amplitude_flags, intervals):
    _chk(det_data, np.float64, 2)
    _chk(flag_data, np.uint8, 2)
    lib().oracle_offset_project_signal(
        C.c_int32(data_index), _p(det_data), C.c_int32(flag_index), _p(flag_data),
        C.c_uint8(flag_mask), _i64(step_length), _i64(amp_offset),
        _p(_chk(n_amp_views, np.int64, 1)), _p(_chk(amplitudes, np.float64, 1)),
        _p(_chk(amplitude_flags, np.uint8, 1)), _p(_chk(intervals, interval_dtype, 1)),
        _i64(intervals.size), _i64(det_data.shape[1]),
    )


def template_offset_apply_diag_precond(offset_var, amp_in, amplitude_flags, amp_out):
    lib().oracle_offset_apply_diag_precond(
        _p(_chk(offset_var, np.float64, 1)), _p(_chk(amp_in, np.float64, 1)),
        _p(_chk(amplitude_flags, np.uint8, 1)), _p(_chk(amp_out, np.float64, 1)), _i64(amp_in.size),
    )


def cov_apply_diag(nsub, subsize, nnz, mat, vec):
    lib().oracle_cov_apply_diag(
        _i64(nsub), _i64(subsize), _i64(nnz), _p(_chk(mat, np.float64)), _p(_chk(vec, np.float64))
    )


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    """Set the OpenMP thread count of the restatement AND of oracle/_ref (one libgomp instance
    per process serves both)."""
    lib().oracle_set_num_threads(C.c_int(int(n)))


def libm_atan2(y, x):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(y)
    lib().oracle_atan2(_i64(y.size), _p(y), _p(x), _p(out))
    return out


def libm_sqrt(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().oracle_sqrt(_i64(x.size), _p(x), _p(out))
    return out


def ieee_div(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty_like(a)
    lib().oracle_div(_i64(a.size), _p(a), _p(b), _p(out))
    return out
