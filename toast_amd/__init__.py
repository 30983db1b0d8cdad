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


def load_native():
    """Import the pybind11 module (raises if it has not been built -- there is no fallback)."""
    from . import _libtoast_hip

    return _libtoast_hip
