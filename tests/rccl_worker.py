"""Worker for tests/test_gpu_rccl.py: the RCCL (backend "nccl") branch of the map reductions.

Run either as ONE process (world_size 1: the collectives still go through RCCL's enqueue path on
the device tensor that wraps the memory manager's pointer -- what a single-GPU box can exercise)
or under torch.distributed.run with one process per GPU (world_size >= 2, needs that many GPUs).
Checks, for both PixelData.sync_allreduce and sync_alltoallv (reference src/toast/pixels.py:710-780,
878-970): device-resident maps written by a library kernel on the library stream right before the
collective (stream ordering in), read by a library kernel right after it (stream ordering out), the
sum over ranks, equality of the two entry points, non-divisible sizes, int64 hit maps, and the
scalar all-reduce used by the amplitude dot products."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from toast_amd.accel import accel_assign_device, native  # noqa: E402
from toast_amd.data import Comm  # noqa: E402
from toast_amd.pixels import PixelData, PixelDistribution, covariance_apply  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=size, device_id=torch.device("cuda", local))
    accel_assign_device(size, rank, 1.0, False)
    comm = Comm(single_rank_collectives=True)
    assert comm.comm_world is not None and comm._dist.get_backend() == "nccl"

    # 37 submaps x 48 pixels x 3: 5328 doubles, not divisible by 5 / 7 / 8 ranks -> padded shards
    d = PixelDistribution(n_pix=64 * 48, n_submap=64, local_submaps=np.arange(3, 40), comm=comm)
    total = np.zeros(37 * 48 * 3)
    parts = [np.random.default_rng(500 + r).standard_normal(total.size) for r in range(size)]
    for p in parts:
        total += p
    cov = PixelData(d, np.float64, n_value=6)
    cov.raw[:] = np.random.default_rng(9).random(cov.raw.size)
    results = {}
    for entry in ("sync_allreduce", "sync_alltoallv"):
        pd = PixelData(d, np.float64, n_value=3)
        pd.raw[:] = parts[rank]
        pd.accel_create("zmap")
        pd.accel_update_device()
        # a library kernel on the library stream immediately before the collective ...
        ident = PixelData(d, np.float64, n_value=6)
        ident.raw.reshape(-1, 6)[:] = np.array([2.0, 0, 0, 2.0, 0, 2.0])
        covariance_apply(ident, pd)                   # zmap *= 2 on the device
        assert pd.accel_in_use()
        getattr(pd, entry)()
        # ... and one immediately after it
        covariance_apply(ident, pd)                   # zmap *= 2 again
        pd.accel_update_host()
        np.testing.assert_allclose(pd.raw, 4.0 * total, rtol=0, atol=1e-13 * np.max(np.abs(total)))
        results[entry] = pd.raw.copy()
        # every rank holds the same bits
        t = torch.from_numpy(pd.raw.copy()).cuda()
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), entry
        pd.accel_delete()
    if size <= 2:
        # two addends (or one): the sum does not depend on the reduction order
        assert np.array_equal(results["sync_allreduce"], results["sync_alltoallv"])
    # host-resident data with the nccl backend (staged through the device)
    ph = PixelData(d, np.int64, n_value=1)
    ph.raw[:] = rank + 1
    ph.sync_allreduce()
    assert np.all(ph.raw == size * (size + 1) // 2)
    ph.raw[:] = rank + 1
    ph.sync_alltoallv()
    assert np.all(ph.raw == size * (size + 1) // 2)
    assert comm.allreduce_scalar(rank + 1, op="max") == size
    assert abs(comm.allreduce_scalar(0.5, op="sum") - 0.5 * size) < 1e-15
    native().accel_synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} of {size} OK", flush=True)


if __name__ == "__main__":
    main()
