"""Map of stream bandwidth over one large allocation, in 1 GiB slices (k_noise_weight, 8 B read +
8 B write): how are the slow / fast HBM regions laid out?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from toast_amd import capi, synth
D = capi.dev
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
GiB = 1 << 30
total = int(sys.argv[1]) if len(sys.argv) > 1 else 96
big = torch.zeros(total * GiB, dtype=torch.uint8, device=dev)
n_samp = 1 << 20                      # 128 rows x 1 Mi samples x 8 B = 1 GiB
n_det = 128
idx = np.arange(n_det, dtype=np.int32)
ivl = synth.make_intervals(n_samp, 1, 1.0)
ones = np.ones(n_det)
print("base %x" % big.data_ptr())
res = []
for g in range(total):
    p = big.data_ptr() + g * GiB
    D.noise_weight(p, n_samp, idx, ivl, ones, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        D.noise_weight(p, n_samp, idx, ivl, ones, stream)
    e1.record(); e1.synchronize()
    res.append(2.0 * GiB / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e12)
for g0 in range(0, total, 16):
    print("GiB %3d..: " % g0 + " ".join("%.2f" % x for x in res[g0:g0 + 16]))
print("min %.2f max %.2f TB/s" % (min(res), max(res)))
