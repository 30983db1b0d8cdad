"""Statistics of the stream rate by allocation recipe (profiles/r02_d_placement_experiments.txt section 2b):
several rounds of [plain hipMalloc, VMM ranges of 2 MB / 64 MB / 1 GB chunks in order, 64 MB shuffled], all kept alive."""
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


recipes = [("plain", None), ("vmm2", (2, 0)), ("vmm64", (64, 0)), ("vmm1024", (1024, 0)), ("vmm64s", (64, 1)),
           ("vmm256", (256, 0))]
res = {k: [] for k, _ in recipes}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    for name, arg in recipes:
        p = C.c_void_p(0)
        if arg is None:
            rc = lib.toast_hip_device_malloc(C.c_size_t(nbytes), C.c_int(0), C.byref(p))
        else:
            rc = lib.toast_hip_device_malloc_vmm(C.c_size_t(nbytes), C.c_int(arg[0]), C.c_int(arg[1]), C.byref(p))
        if rc != 0:
            print("alloc failed", name, lib.toast_hip_last_error().decode())
            continue
        lib.toast_hip_memset_dev(p, C.c_int(0), C.c_size_t(nbytes), C.c_void_p(st))
        res[name].append(stream_ms(p.value))
for name, _ in recipes:
    print("%-8s " % name + " ".join("%.3f" % t for t in res[name]), flush=True)
