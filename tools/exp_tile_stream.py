"""(needs the experimental tile build: the "tile_dt" / "tile_ct" tuning keys are not in the library; results in
profiles/r02_d_placement_experiments.txt section 4.)  TLB-reach hypothesis for the slow HBM level: stream (k_noise_weight) over slow (hipDeviceMallocContiguous) and
plain allocations with the resident workgroups covering a tile of dt detectors x ct chunks."""
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


TILES = [(0, 0), (1024, 2), (256, 8), (128, 16), (64, 32), (32, 64), (16, 128), (8, 256), (4, 704), (1, 704), (64, 8), (256, 2)]
print("tiles (dt, ct):", TILES)
for i in range(6):
    flags = 4 if i % 2 else 0
    p = C.c_void_p(0)
    assert lib.toast_hip_device_malloc(C.c_size_t(nbytes), C.c_int(flags), C.byref(p)) == 0
    lib.toast_hip_memset_dev(p, C.c_int(0), C.c_size_t(nbytes), C.c_void_p(st))
    row = []
    for dt, ct in TILES:
        lib.toast_hip_set_tuning(b"tile_dt", C.c_int(dt))
        lib.toast_hip_set_tuning(b"tile_ct", C.c_int(ct))
        row.append(stream_ms(p.value))
    lib.toast_hip_set_tuning(b"tile_dt", C.c_int(0))
    lib.toast_hip_set_tuning(b"tile_ct", C.c_int(0))
    print("%-10s " % ("contiguous" if flags else "plain") + " ".join("%.3f" % t for t in row), flush=True)
