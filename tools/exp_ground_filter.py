#!/usr/bin/env python3
"""Time the ground-filter kernels at the per-GPU shape of BASELINE configs[4] (2048 detectors over
8 GPUs = 256 detectors, one hour CES at 200 Hz) and the operator end to end.

    python tools/exp_ground_filter.py [n_det] [n_samp] [trend_order] [filter_order]
"""
import sys
import time

import numpy as np
import torch

from toast_amd import capi


def main():
    n_det = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 720000
    trend = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    order = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    nt = trend + 2 * (order + 1)   # split templates
    dev = torch.device("cuda")
    x = torch.linspace(-1, 1, n, dtype=torch.float64, device=dev)
    templates = torch.empty((nt, n), dtype=torch.float64, device=dev)
    sig = torch.randn((n_det, n), dtype=torch.float64, device=dev)
    dflags = (torch.rand((n_det, n), device=dev) < 0.01).to(torch.uint8)
    sflags = (torch.rand(n, device=dev) < 0.05).to(torch.uint8)
    proj = torch.zeros((n_det, nt), dtype=torch.float64, device=dev)
    gram = torch.zeros((nt, nt), dtype=torch.float64, device=dev)
    dgram = torch.zeros((n_det, nt, nt), dtype=torch.float64, device=dev)
    nflag = torch.zeros(n_det, dtype=torch.int64, device=dev)
    coeff = torch.randn((n_det, nt), dtype=torch.float64, device=dev) * 1e-3
    idx = np.arange(n_det, dtype=np.int32)
    D = capi.dev

    def legendre():
        D.legendre_templates(x.data_ptr(), n, 0, nt, templates.data_ptr())

    def fit():
        D.template_fit(templates.data_ptr(), nt, n, idx, sig.data_ptr(), idx, dflags.data_ptr(), 1, sflags.data_ptr(), 1,
                       proj.data_ptr(), gram.data_ptr(), dgram.data_ptr(), nflag.data_ptr())

    def subtract():
        D.template_subtract(templates.data_ptr(), nt, trend, n, idx, sig.data_ptr(), coeff.data_ptr())

    nds = n_det * n
    for name, fn, nbytes in (("legendre_templates", legendre, 8 * n * (nt + 1)), ("template_fit", fit, 9 * nds),
                             ("template_subtract", subtract, 16 * nds)):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:20s} {ms:8.3f} ms   {nds / ms / 1e6:7.1f} G det-samples/s   {nbytes / ms / 1e9:6.2f} TB/s algorithmic "
              f"({n_det} det x {n} samples, {nt} templates)")
    # host side of the fit: n_det solves of nt x nt
    t0 = time.time()
    g = (gram[None] - dgram).cpu().numpy()
    p = proj.cpu().numpy()
    for d in range(n_det):
        np.linalg.cond(g[d])
        np.dot(np.linalg.inv(g[d]), p[d])
    print(f"host solves          {1e3 * (time.time() - t0):8.3f} ms")


if __name__ == "__main__":
    main()
