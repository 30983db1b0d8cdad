#!/usr/bin/env python3
"""Time toast_hip_template_offset_banded_solve_dev / _convolve_dev at the cfg3 shape of the Offset
noise prior (1024 detectors x 3600 one-second baselines, band width 20, 87-tap filter).

    python tools/exp_banded_solve.py [n_seg] [n_amp] [width] [taps]
"""
import sys

import numpy as np
import scipy.linalg
import torch

from toast_amd import capi


def main():
    n_seg = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    taps = int(sys.argv[4]) if len(sys.argv) > 4 else 87
    rng = np.random.default_rng(0)
    ab = np.zeros((w, n))
    ab[0] = 3.0 + rng.random(n) * 5.0
    ab += (0.9 ** np.arange(w))[:, None] / w
    cb = scipy.linalg.cholesky_banded(ab, lower=True)
    f = np.zeros((n, w))
    b = np.zeros((n, w))
    f[:, 0] = b[:, 0] = 1.0 / cb[0]
    for k in range(1, min(w, n)):
        f[:n - k, k] = cb[k, :n - k]
        b[k:, k] = cb[k, :n - k]
    dev = torch.device("cuda")
    fwd = torch.from_numpy(np.tile(f.ravel(), n_seg)).to(dev)
    bwd = torch.from_numpy(np.tile(b.ravel(), n_seg)).to(dev)
    seg_start = torch.arange(n_seg + 1, dtype=torch.int64, device=dev) * n
    bw = torch.full((n_seg,), w, dtype=torch.int32, device=dev)
    bs = torch.arange(n_seg, dtype=torch.int64, device=dev) * (n * w)
    x = torch.randn(n_seg * n, dtype=torch.float64, device=dev)
    flags = torch.zeros(n_seg * n, dtype=torch.uint8, device=dev)
    out = torch.empty_like(x)
    filt = torch.from_numpy(np.exp(-np.abs(np.arange(taps) - taps // 2) / 9.0)).to(dev)
    fstart = torch.zeros(n_seg, dtype=torch.int64, device=dev)
    flen = torch.full((n_seg,), taps, dtype=torch.int64, device=dev)

    def solve():
        capi.dev.offset_banded_solve(n_seg, seg_start.data_ptr(), bw.data_ptr(), w, bs.data_ptr(), fwd.data_ptr(),
                                     bwd.data_ptr(), x.data_ptr(), flags.data_ptr(), out.data_ptr())

    def conv():
        capi.dev.offset_convolve(n_seg * n, n_seg, seg_start.data_ptr(), n, fstart.data_ptr(), flen.data_ptr(), taps,
                                 filt.data_ptr(), x.data_ptr(), flags.data_ptr(), out.data_ptr(), False)

    for name, fn in (("banded_solve", solve), ("convolve", conv)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name}: {ms:.3f} ms  ({n_seg} segments x {n} amplitudes, width {w}, taps {taps}; "
              f"{1e6 * ms / (2 * n):.0f} ns per substitution step)" if name == "banded_solve" else
              f"{name}: {ms:.3f} ms  ({n_seg * n * taps / ms / 1e6:.1f} G multiply-adds/s)")
    ref = scipy.linalg.cho_solve_banded((cb, True), x[:n].cpu().numpy())
    solve()
    torch.cuda.synchronize()
    print("max rel err vs scipy:", np.max(np.abs(out[:n].cpu().numpy() - ref)) / np.max(np.abs(ref)))


if __name__ == "__main__":
    main()
