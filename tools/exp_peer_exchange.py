#!/usr/bin/env python3
"""Experiment: what do the kernels of TOAST_HIP_COMM_MODE=peer cost when no link is involved?

W processes on the ONE GPU of the test box (collective bootstrap and the two barriers through the shared-memory
stand-in for librccl, tests/librccl_mock.so), a replicated map of the cfg-3 size (12.6 M pixels x 3 doubles = 302 MB):
every rank's push / sum / pull run against exchange buffers that live in the same HBM, so the times are the kernels'
own cost (what remains when the links are infinitely fast), not a statement about xGMI.  The barriers of the stand-in
are host round trips; their cost is measured with a one-word all-reduce and reported next to the total.

    python tools/exp_peer_exchange.py [ranks]        (launches its ranks itself)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main():
    import numpy as np
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from toast_amd import capi
    from toast_amd.accel import accel_assign_device
    from toast_amd.data import Comm

    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    accel_assign_device(size, rank, 1.0, False)
    comm = Comm()
    assert comm.device_comm()
    n_px, nnz = 3072 * 4096, 3
    z = torch.full((n_px * nnz,), float(rank + 1), dtype=torch.float64, device="cuda")
    word = torch.zeros(1, dtype=torch.int32, device="cuda")

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    out = {}
    for mode in ("peer", "peer:flags"):      # ("owner" would need the whole map in one 96 MB slot of the stand-in)
        capi.dev.comm_set_mode(mode)
        z.fill_(float(rank + 1))
        capi.dev.comm_map_reduce_apply(n_px, nnz, 0, z.data_ptr(), reduce=True)
        torch.cuda.synchronize()
        want = size * (size + 1) / 2
        assert float(z.min()) == want and float(z.max()) == want, (mode, float(z.min()), float(z.max()), want)
        out[mode] = timed(lambda: capi.dev.comm_map_reduce_apply(n_px, nnz, 0, z.data_ptr(), reduce=True), 5)
    out["one_word_allreduce"] = timed(lambda: capi.dev.comm_allreduce(word.data_ptr(), 1, np.int32, "max"), 20)
    if rank == 0:
        print("ranks %d  map %.0f MB   peer %.3f ms per reduction (of which 2 barriers of the stand-in: %.3f ms)   "
              "peer:flags %.3f ms   exchange buffer %.0f MB per rank"
              % (size, n_px * nnz * 8 / 1e6, out["peer"], 2 * out["one_word_allreduce"], out["peer:flags"],
                 capi.dev.comm_peer_stats()[2] / 1e6), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if "RANK" in os.environ:
        return rank_main()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mock = os.path.join(ROOT, "tests", "librccl_mock.so")
    src = os.path.join(ROOT, "tests", "rccl_mock.cpp")
    if not os.path.exists(mock) or os.path.getmtime(mock) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        src, "-o", mock, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TOAST_HIP_COMM="rccl",
               TOAST_HIP_RCCL_LIB=mock, TOAST_GPU_MEM_GB="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", "29631", os.path.abspath(__file__)]
    raise SystemExit(subprocess.run(cmd, env=env, cwd=ROOT).returncode)


if __name__ == "__main__":
    main()
