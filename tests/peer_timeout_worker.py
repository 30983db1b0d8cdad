"""Worker for tests/test_gpu_rccl_mock.py: TOAST_HIP_COMM_MODE=peer:flags when a rank does not arrive.

Two processes on the one GPU.  Rank 1 comes 2 s late to a reduction whose flag wait gives up after 0.3 s
(TOAST_HIP_COMM_PEER_TIMEOUT_MS): rank 0's waiting kernel must END (nothing may stay parked on the GPU), and the library
must say so loudly at rank 0's next call instead of handing out a map that misses a contribution."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from toast_amd import capi  # noqa: E402
from toast_amd.accel import accel_assign_device  # noqa: E402
from toast_amd.data import Comm  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    assert size == 2
    torch.cuda.set_device(0)
    accel_assign_device(size, rank, 1.0, False)
    assert Comm().device_comm()
    capi.dev.comm_set_mode("peer:flags")
    n = 4096
    z = torch.full((n,), float(rank + 1), dtype=torch.float64, device="cuda")
    reduce = lambda: capi.dev.comm_map_reduce_apply(n, 1, 0, z.data_ptr(), reduce=True)
    reduce()                                   # both in time: buffers and flag blocks established, 1 + 2
    torch.cuda.synchronize()
    assert float(z.min()) == 3.0 and float(z.max()) == 3.0
    dist.barrier()
    if rank == 1:
        time.sleep(2.0)
    t0 = time.perf_counter()
    reduce()
    torch.cuda.synchronize()                   # rank 0: returns after ~0.3 s although rank 1 has not arrived
    waited = time.perf_counter() - t0
    if rank == 0:
        # (on a box so loaded that rank 0 itself needed more than the 2 s to get here, rank 1 was in time and there is
        # nothing to report: the kernel still ended, which is the point; otherwise the error must come out)
        assert waited < 10.0, waited
        if 0.25 < waited < 1.7:
            try:
                reduce()
            except RuntimeError as err:
                assert "waited in vain for the flag of rank 1" in str(err), err
            else:
                raise AssertionError("rank 0: the time-out of the previous reduction was not reported")
        else:
            print(f"rank 0: no time-out observed (waited {waited:.2f} s)")
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} OK")


if __name__ == "__main__":
    main()
