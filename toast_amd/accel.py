"""Accelerator interface with the names of ``toast.accelerator.accel``
(reference: src/toast/accelerator/accel.py:20-305), routed to the HIP memory manager."""

import os

import numpy as np

from . import load_native

_native = None


def native():
    """The pybind11 module (``toast._libtoast`` counterpart); raises if not built."""
    global _native
    if _native is None:
        _native = load_native()
    return _native


_assigned = False


def accel_enabled():
    """True when a gfx950 device is usable (reference accel.py:69-77)."""
    if os.environ.get("TOAST_GPU_DISABLE", "0") not in ("0", "", "false", "False"):
        return False
    return bool(native().accel_enabled())


def accel_assign_device(node_procs, node_rank, mem_gb, disabled=False):
    """reference accel.py:94-118"""
    global _assigned
    native().accel_assign_device(int(node_procs), int(node_rank), float(mem_gb), bool(disabled))
    _assigned = True
    # one HIP runtime, one current device: torch (device tensors handed to RCCL, the collectives'
    # scratch tensors) must sit on the GPU the memory manager chose for this process
    dev = int(native().accel_get_device())
    if dev >= 0:
        import torch

        if torch.cuda.is_available() and torch.cuda.current_device() != dev:
            torch.cuda.set_device(dev)


def ensure_assigned():
    """Assign a device on first use: one process per GPU, device = LOCAL_RANK."""
    if not _assigned:
        nproc = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        rank = int(os.environ.get("LOCAL_RANK", "0"))
        # the reference's default pool size and its knob (src/toast/mpi.py:60-63)
        accel_assign_device(max(nproc, 1), rank, float(os.environ.get("TOAST_GPU_MEM_GB", "2.0")), False)


def accel_get_device():
    ensure_assigned()
    return native().accel_get_device()


def _key(data):
    return np.asarray(data)


def accel_data_present(data, name="None"):
    """reference accel.py:121-144"""
    if data is None:
        return False
    ensure_assigned()
    arr = _key(data)
    if arr.size == 0:
        return False   # empty buffers (an observation without valid detectors) are never registered
    return bool(native().accel_present(arr, name))


_finalizers = {}


def _release(ptr, nbytes, name):
    """Finalizer: drop the device copy of a host buffer whose owner was garbage collected.
    Device copies are keyed by host address (like the reference's OmpManager), so a leaked
    entry would be mistaken for the next allocation at that address."""
    import ctypes

    from . import capi

    _finalizers.pop(ptr, None)
    try:
        # the real handle: a garbage collection inside a capi.capture() block must free the device
        # copy now, not record the call into the plan that is replayed every iteration
        lib = capi.real_lib()
        present = ctypes.c_int(0)
        rc = lib.toast_hip_accel_present(ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), ctypes.byref(present))
        if rc == 0 and present.value:
            lib.toast_hip_accel_delete(ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), name.encode())
    except Exception:  # interpreter shutdown
        pass


_eviction_handlers = []


def add_eviction_handler(fn):
    """Register ``fn() -> bytes_freed``; called when a device allocation fails so that lazily
    retained buffers can be written back and released before one retry."""
    _eviction_handlers.append(fn)


#: what an array is to the kernels (toast_hip_accel_create_kind): decides where in HBM the arena puts its device copy
KIND_DEFAULT, KIND_STREAMED, KIND_SCATTER = 0, 1, 2


def accel_data_create(data, name="None", zero_out=False, owner=None, kind=KIND_DEFAULT):
    """reference accel.py:147-176.  ``owner``: object whose lifetime bounds the device copy.  ``kind``: KIND_STREAMED for
    a timestream that kernels read and write in their sweeps, KIND_SCATTER for the target of a scatter with atomics (a
    map, an amplitude vector); toast_hip_accel_create_kind."""
    import weakref

    ensure_assigned()
    arr = _key(data)
    if arr.size == 0:
        return data
    try:
        native().accel_create(arr, name, int(kind))
    except RuntimeError as err:
        if "allocation failed" not in str(err):
            raise
        freed = 0
        for fn in list(_eviction_handlers):
            freed += int(fn() or 0)
        if freed == 0:
            raise
        native().accel_create(arr, name, int(kind))
    if zero_out:
        native().accel_reset(arr, name)
    if owner is not None:
        ptr = arr.ctypes.data
        _finalizers[ptr] = weakref.finalize(owner, _release, ptr, arr.nbytes, name)
    return data


def accel_data_reset(data, name="None"):
    ensure_assigned()
    if _key(data).size > 0:
        native().accel_reset(_key(data), name)
    return data


def accel_data_update_device(data, name="None"):
    ensure_assigned()
    if _key(data).size > 0:
        native().accel_update_device(_key(data), name)
    return data


def accel_data_update_host(data, name="None"):
    ensure_assigned()
    if _key(data).size > 0:
        native().accel_update_host(_key(data), name)
    return data


def accel_data_delete(data, name="None"):
    ensure_assigned()
    arr = _key(data)
    if arr.size == 0:
        return data
    fin = _finalizers.pop(arr.ctypes.data, None)
    if fin is not None:
        fin.detach()
    native().accel_delete(arr, name)
    return data


def accel_device_ptr(data):
    ensure_assigned()
    if _key(data).size == 0:
        return 0
    return int(native().accel_device_ptr(_key(data)))


class accel_hold:
    """``with accel_hold(a, b): ...`` -- the device copies of ``a`` and ``b`` are not evicted inside the block
    (``Data.accel_evict`` skips held objects).  For code outside a Pipeline that allocates device memory while it
    holds another object's device state: a failed allocation triggers eviction, and writing back / freeing the very
    operand the code is about to read would leave it with a stale device pointer."""

    def __init__(self, *objs):
        self._objs = [o for o in objs if o is not None]

    def __enter__(self):
        for o in self._objs:
            o._accel_hold = getattr(o, "_accel_hold", 0) + 1
        return self

    def __exit__(self, *exc):
        for o in self._objs:
            o._accel_hold -= 1
        return False


class AcceleratorObject:
    """Mix-in for objects with a device copy (reference: accel.py:308-520).

    ``accel_exists``: a device buffer is allocated.  ``accel_in_use``: the device copy is the
    current one (the host copy may be stale)."""

    def __init__(self, accel_name="(blank)"):
        self._accel_used = False
        self._accel_name = accel_name

    def _accel_exists(self):
        return False

    def accel_exists(self):
        if not accel_enabled():
            return False
        return self._accel_exists()

    def accel_in_use(self):
        return self._accel_used

    def accel_used(self, state):
        if state and not self.accel_exists():
            raise RuntimeError("Data is not present on device, cannot set as 'used'")
        self._accel_used = bool(state)

    def _accel_create(self, **kwargs):
        pass

    def accel_create(self, name=None, **kwargs):
        if name is not None:
            self._accel_name = name
        if self.accel_exists():
            raise RuntimeError(f"Data already exists on device, cannot create ({self._accel_name})")
        self._accel_create(**kwargs)

    def _accel_update_device(self):
        pass

    def accel_update_device(self):
        if not self.accel_exists():
            raise RuntimeError(f"Data does not exist on device, cannot update ({self._accel_name})")
        if self.accel_in_use():
            raise RuntimeError("Active data is already on device, cannot update")
        self._accel_update_device()
        self.accel_used(True)

    def _accel_update_host(self):
        pass

    def accel_update_host(self):
        if not self.accel_exists():
            raise RuntimeError(f"Data does not exist on device, cannot update host ({self._accel_name})")
        if not self.accel_in_use():
            raise RuntimeError("Active data is already on host, cannot update")
        self._accel_update_host()
        self.accel_used(False)

    def _accel_delete(self):
        pass

    def accel_delete(self):
        if not self.accel_exists():
            raise RuntimeError(f"Data does not exist on device, cannot delete ({self._accel_name})")
        self._accel_delete()
        self._accel_used = False

    def _accel_reset(self):
        pass

    def accel_reset(self):
        if not self.accel_exists():
            raise RuntimeError(f"Data does not exist on device, cannot reset ({self._accel_name})")
        self._accel_reset()
