"""toast_amd -- MI355X (gfx950) implementation of TOAST's map-making hot path.

Layers:

* ``libtoast_hip.so``   hand-written HIP kernels + device memory manager behind the C ABI of
                        ``include/toast_hip.h``
* ``_libtoast_hip``     pybind11 module with the names / signatures of ``toast._libtoast``
* ``capi``              ctypes view of the same C ABI (host- and device-pointer levels)
* ``ops``, ``accel``    Python mirror of the reference's Operator / accelerator interface

One HIP runtime per process: PyTorch wheels bundle their own ``libamdhip64.so`` and must be
imported before anything links ``/opt/rocm``'s copy, so torch (used for device tensors,
streams and torch.distributed/RCCL) is imported here first when it is installed.
"""

try:  # see docstring
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover
    _torch = None

__version__ = "0.1.0"


def _seed_rocfft_cache():
    """rocFFT compiles its kernels at run time (~1.9 s for the first 2^21-point plan on a fresh
    machine).  Ship the compiled-kernel cache of the BASELINE transform lengths
    (tools/make_rocfft_cache.py) and seed the user's cache file with it, unless the user already
    points ROCFFT_RTC_CACHE_PATH somewhere.  A stale or unusable cache only means rocFFT compiles."""
    import os
    import shutil

    if "ROCFFT_RTC_CACHE_PATH" in os.environ:
        return
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rocfft_rtc_cache_gfx950.db")
    if not os.path.isfile(src):
        return
    # working copy (rocFFT appends to it): next to the package when that is writable, else in
    # the user's cache directory
    here = os.path.dirname(os.path.abspath(__file__))
    user = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"),
                        "toast_amd")
    for dst_dir in (os.path.join(here, ".rocfft_cache"), user):
        try:
            os.makedirs(dst_dir, exist_ok=True)
            dst = os.path.join(dst_dir, "rocfft_rtc_cache_gfx950.db")
            if not os.path.isfile(dst):
                tmp = dst + ".%d.tmp" % os.getpid()
                shutil.copyfile(src, tmp)
                os.replace(tmp, dst)
            os.environ["ROCFFT_RTC_CACHE_PATH"] = dst
            return
        except OSError:
            continue


_seed_rocfft_cache()


def load_native():
    """Import the pybind11 module (raises if it has not been built -- there is no fallback)."""
    from . import _libtoast_hip

    return _libtoast_hip
