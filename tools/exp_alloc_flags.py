"""Stream rate of buffers by allocation kind: plain hipMalloc vs hipExtMallocWithFlags(hipDeviceMallocContiguous),
interleaved, all alive at once (the HBM-region effect of profiles/r01_b_tuning_experiments.txt section 14)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from toast_amd import capi, synth

n_det, n_samp, rate = 1024, 720000, 200.0
D = capi.dev
lib = capi.real_lib()
torch.cuda.init()
st = torch.cuda.current_stream().cuda_stream
idx = np.arange(n_det, dtype=np.int32)
ivl = synth.make_intervals(n_samp, 1, rate)
ones = np.ones(n_det)
nbytes = n_det * n_samp * 8


def stream_ms(ptr):
    D.noise_weight(ptr, n_samp, idx, ivl, ones, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        D.noise_weight(ptr, n_samp, idx, ivl, ones, st)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 3


bufs = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    flags = 4 if i % 2 else 0
    p = C.c_void_p(0)
    rc = lib.toast_hip_device_malloc(C.c_size_t(nbytes), C.c_int(flags), C.byref(p))
    if rc != 0:
        print("alloc failed", flags, lib.toast_hip_last_error().decode())
        continue
    lib.toast_hip_memset_dev(p, C.c_int(0), C.c_size_t(nbytes), C.c_void_p(st))
    bufs.append((flags, p))
for flags, p in bufs:
    t = stream_ms(p.value)
    print("flags %d  ptr %#x  stream %.3f ms = %.2f TB/s" % (flags, p.value, t, 2 * nbytes / t / 1e9), flush=True)
# virtual-memory ranges backed by separately created physical chunks, mapped in order / shuffled
for chunk_mb, shuffled in ((2, 0), (2, 1), (64, 0), (64, 1), (1024, 0), (1024, 1), (2, 1), (64, 1)):
    p = C.c_void_p(0)
    rc = lib.toast_hip_device_malloc_vmm(C.c_size_t(nbytes), C.c_int(chunk_mb), C.c_int(shuffled), C.byref(p))
    if rc != 0:
        print("vmm alloc failed", chunk_mb, shuffled, lib.toast_hip_last_error().decode())
        continue
    lib.toast_hip_memset_dev(p, C.c_int(0), C.c_size_t(nbytes), C.c_void_p(st))
    t = stream_ms(p.value)
    print("vmm chunk %4d MB %s  ptr %#x  stream %.3f ms = %.2f TB/s" % (chunk_mb, "shuffled" if shuffled else "in order",
                                                                         p.value, t, 2 * nbytes / t / 1e9), flush=True)
# free every other one, allocate again (fragmented pool)
for flags, p in bufs[::3]:
    lib.toast_hip_device_free(p)
print("after freeing a third:")
for i in range(4):
    flags = 4 if i % 2 else 0
    p = C.c_void_p(0)
    if lib.toast_hip_device_malloc(C.c_size_t(nbytes), C.c_int(flags), C.byref(p)) != 0:
        print("alloc failed", flags)
        continue
    lib.toast_hip_memset_dev(p, C.c_int(0), C.c_size_t(nbytes), C.c_void_p(st))
    t = stream_ms(p.value)
    print("flags %d  ptr %#x  stream %.3f ms = %.2f TB/s" % (flags, p.value, t, 2 * nbytes / t / 1e9), flush=True)
