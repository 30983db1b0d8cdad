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


# No import-time side effects: the rocFFT kernel-cache seeding of round 1 (a 0.5 MB binary copied
# into the user's cache directory and ROCFFT_RTC_CACHE_PATH set at import) is gone.  The noise
# weighting of map-making-sized timestreams runs on the hand-written kernels of csrc/fft_fused.hip
# (nothing to compile at run time); rocFFT serves only short transforms and the FFTPlanReal1D
# plans, where its one-off run-time compilation is its own documented behaviour
# (tools/make_rocfft_cache.py builds a cache file for users who want to pre-seed it themselves).


def load_native():
    """Import the pybind11 module (raises if it has not been built -- there is no fallback)."""
    from . import _libtoast_hip

    return _libtoast_hip
