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

from toast_amd.accel import accel_assign_device, accel_device_ptr, native  # noqa: E402
from toast_amd.data import Comm  # noqa: E402
from toast_amd.pixels import PixelData, PixelDistribution, covariance_apply  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    size = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    # "nccl": one process per GPU.  "gloo" (with TOAST_HIP_COMM=rccl and TOAST_HIP_RCCL_LIB = tests/librccl_mock.so):
    # several ranks on ONE GPU, the library's communicator over the shared-memory stand-in -- same code in comm.cpp
    backend = os.environ.get("TOAST_TEST_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=size, device_id=torch.device("cuda", local))
    else:
        assert os.environ.get("TOAST_HIP_COMM") == "rccl" and os.environ.get("TOAST_HIP_RCCL_LIB")
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=size)
    check_dev = "cuda" if backend == "nccl" else "cpu"     # where the test's own cross-checks are reduced
    accel_assign_device(size, rank, 1.0, False)
    comm = Comm(single_rank_collectives=True)
    assert comm.comm_world is not None and comm._dist.get_backend() == backend

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
        t = torch.from_numpy(pd.raw.copy()).to(check_dev)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), entry
        pd.accel_delete()
    if size <= 2:
        # two addends (or one): the sum does not depend on the reduction order
        assert np.array_equal(results["sync_allreduce"], results["sync_alltoallv"])
    # the collectives above went through the library's own communicator (toast_hip_comm_*), created on first use
    from toast_amd import capi

    # ... enqueued on the kernels' stream: no host synchronisation on the way (kernel -> collective -> kernel)
    nat = native()
    real_sync, calls = nat.accel_synchronize, []
    nat.accel_synchronize = lambda: (calls.append(1), real_sync())[1]
    real_tsync = torch.cuda.Stream.synchronize
    torch.cuda.Stream.synchronize = lambda self: (calls.append(2), real_tsync(self))[1]
    try:
        pd = PixelData(d, np.float64, n_value=3)
        pd.raw[:] = parts[rank]
        pd.accel_create("zmap")
        pd.accel_update_device()
        covariance_apply(ident, pd)
        pd.sync_allreduce()
        covariance_apply(ident, pd)
        pd.sync_alltoallv()
        covariance_apply(ident, pd, use_alltoallv=True)
        assert calls == [], calls
    finally:
        nat.accel_synchronize = real_sync
        torch.cuda.Stream.synchronize = real_tsync
    pd.accel_update_host()
    np.testing.assert_allclose(pd.raw, 8.0 * size * total, rtol=0, atol=1e-12 * size * np.max(np.abs(total)))
    pd.accel_delete()

    assert comm.device_comm() and capi.dev.comm_info()[:2] == (size, rank) and capi.dev.comm_info()[2] > 0
    first, count = capi.dev.comm_pixel_shard(37 * 48)
    per = -(-37 * 48 // size)
    assert first == min(rank * per, 37 * 48) and count == max(0, min(37 * 48, (rank + 1) * per) - first)
    # raw collectives of the C ABI on torch tensors (library stream = the default stream)
    t = torch.arange(size * 5, dtype=torch.float64, device="cuda") * (rank + 1)
    mine = torch.empty(5, dtype=torch.float64, device="cuda")
    capi.dev.comm_reduce_scatter(t.data_ptr(), mine.data_ptr(), 5, np.float64)
    tri = size * (size + 1) // 2
    assert torch.equal(mine.cpu(), torch.arange(rank * 5, rank * 5 + 5, dtype=torch.float64) * tri)
    back = torch.empty(size * 5, dtype=torch.float64, device="cuda")
    capi.dev.comm_all_gather(mine.data_ptr(), back.data_ptr(), 5, np.float64)
    assert torch.equal(back.cpu(), torch.arange(size * 5, dtype=torch.float64) * tri)
    b = torch.full((7,), float(rank + 3), dtype=torch.float64, device="cuda")
    capi.dev.comm_broadcast(b.data_ptr(), 7, np.float64, root=size - 1)
    assert torch.all(b.cpu() == size + 2)
    u = torch.tensor([rank, 5, 250 - rank], dtype=torch.uint8, device="cuda")
    capi.dev.comm_allreduce(u.data_ptr(), 3, np.uint8, "max")
    assert u.cpu().tolist() == [size - 1, 5, 250]

    # the pybind front end of the same calls (what a maintainer binds in PixelData)
    assert nat.comm_info()[:2] == (size, rank)
    pz = PixelData(d, np.float64, n_value=3)
    pz.raw[:] = parts[rank]
    pz.accel_create("zmap")
    pz.accel_update_device()
    nat.comm_allreduce(pz.buffer, "sum")
    nat.comm_map_reduce_apply(None, pz.buffer, 3, True)          # sums the (already equal) copies again
    pz.accel_update_host()
    np.testing.assert_allclose(pz.raw, size * total, rtol=0, atol=1e-13 * size * np.max(np.abs(total)))
    pz.accel_delete()

    # the PCG's dot product summed over the ranks on the stream before the stage reads it (toast_hip_pcg_stage_dev with
    # allreduce = 1) against the single-launch form of one process: p . Ap -> alpha = delta / (size * local dot)
    n_amp = 10007
    xs = torch.from_numpy(np.random.default_rng(900 + rank).standard_normal(n_amp)).cuda()
    ys = torch.from_numpy(np.random.default_rng(950 + rank).standard_normal(n_amp)).cuda()
    local_dot = float(torch.dot(xs, ys).item())
    tot = torch.tensor([local_dot], dtype=torch.float64, device=check_dev)
    dist.all_reduce(tot)
    state = torch.zeros(capi.dev.pcg_state_bytes(5) // 8 + 1, dtype=torch.float64, device="cuda")
    capi.dev.pcg_init(state.data_ptr(), 4.0, 3.0, 1e-12, 3, 5)
    capi.dev.pcg_dot(state.data_ptr(), n_amp, xs.data_ptr(), ys.data_ptr(), 0, 0, accumulate=False, stage=0)
    capi.dev.pcg_stage(state.data_ptr(), 1, allreduce=True)
    res = torch.zeros(n_amp, dtype=torch.float64, device="cuda")
    resid = torch.zeros(n_amp, dtype=torch.float64, device="cuda")
    capi.dev.pcg_step(state.data_ptr(), n_amp, xs.data_ptr(), res.data_ptr(), ys.data_ptr(), resid.data_ptr())
    torch.cuda.synchronize()
    alpha = 3.0 / float(tot.item())
    assert torch.allclose(res, alpha * xs, rtol=1e-12, atol=0) and torch.allclose(resid, -alpha * ys, rtol=1e-12, atol=0)
    if size == 1:
        state2 = torch.zeros_like(state)
        capi.dev.pcg_init(state2.data_ptr(), 4.0, 3.0, 1e-12, 3, 5)
        capi.dev.pcg_dot(state2.data_ptr(), n_amp, xs.data_ptr(), ys.data_ptr(), 0, 0, accumulate=False, stage=1)
        res2 = torch.zeros_like(res)
        resid2 = torch.zeros_like(res)
        capi.dev.pcg_step(state2.data_ptr(), n_amp, xs.data_ptr(), res2.data_ptr(), ys.data_ptr(), resid2.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(res, res2) and torch.equal(resid, resid2)

    # owner-computes covariance operations on device-resident operands (reduce-scatter / kernel on the owned pixel
    # shard / all-gather) against the all-local kernels: covariance.py:78-131, 179-221, 262-306
    from toast_amd.pixels import covariance_invert, covariance_multiply, map_reduce_apply

    crng = np.random.default_rng(41)
    a_ = crng.standard_normal((37 * 48, 3, 3))
    spd = (a_ @ a_.transpose(0, 2, 1) + 0.5 * np.eye(3))[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]

    def dev_map(values, n_value, name):
        out = PixelData(d, np.float64, n_value=n_value)
        out.raw[:] = np.asarray(values).reshape(-1)
        out.accel_create(name)
        out.accel_update_device()
        return out

    mvals = crng.standard_normal((37 * 48, 3))
    want = {}
    for alltoallv in (False, True):
        cv, mp = dev_map(spd, 6, "cov"), dev_map(mvals, 3, "map")
        covariance_apply(cv, mp, use_alltoallv=alltoallv)
        inv, rc = dev_map(spd, 6, "inv"), PixelData(d, np.float64, n_value=1)
        covariance_invert(inv, 1e-6, rcond=rc, use_alltoallv=alltoallv)
        assert inv.accel_in_use() and rc.accel_in_use()
        prod = dev_map(spd, 6, "prod")
        covariance_multiply(prod, inv, use_alltoallv=alltoallv)
        got = dict(apply=mp.data.copy(), inv=inv.data.copy(), rc=rc.data.copy(), prod=prod.data.copy())
        if not alltoallv:
            want = got
        else:
            for key in got:
                np.testing.assert_allclose(got[key], want[key], rtol=1e-13, atol=1e-15, err_msg=key)
        for obj in (cv, mp, inv, rc, prod):
            obj.accel_delete()
    np.testing.assert_allclose(want["prod"].reshape(-1, 6), np.tile([1.0, 0, 0, 1.0, 0, 1.0], (37 * 48, 1)), atol=1e-9)
    # the middle of a PCG iteration in one pass: zmap <- C . sum over ranks (zmap)
    for sync_type in ("alltoallv", "allreduce"):
        cv, z = dev_map(spd, 6, "cov"), dev_map(parts[rank], 3, "zmap")
        map_reduce_apply(cv, z, sync_type=sync_type)
        ref = PixelData(d, np.float64, n_value=3)
        ref.raw[:] = total
        covariance_apply(dev_map(spd, 6, "cov2"), ref)       # host map: staged through the same kernel
        np.testing.assert_allclose(z.data, ref.data, rtol=1e-13, atol=1e-13 * np.max(np.abs(total)))
        cv.accel_delete()
        z.accel_delete()
    # the three implementations of the owner-computes pass (toast_hip_comm_set_mode): same sums, same product
    # ("peer": no RCCL on the data path -- slices written into / read from the owners' hipIpc-opened exchange buffers;
    # between processes that share this GPU here, between GPUs over xGMI on a node)
    # ("peer:flags": the same with the two barriers of a reduction done by device flags in each other's memory)
    # (both lane widths of the exchange kernels: 8-byte system-scope atomics and ordinary 16-byte accesses)
    first_peer = True
    for mode, width in (("owner", 8), ("allreduce", 8), ("peer", 8), ("peer:flags", 8), ("peer", 16), ("peer:flags", 16),
                        ("owner", 8)):
        capi.dev.comm_set_peer_width(width)
        capi.dev.comm_set_mode(mode)
        peer_before = capi.dev.comm_peer_stats()
        assert capi.dev.comm_get_mode() == mode
        for reduce in (True, False):
            cv, z = dev_map(spd, 6, "cov"), dev_map(parts[rank] if reduce else total, 3, "zmap")
            capi.dev.comm_map_reduce_apply(37 * 48, 3, accel_device_ptr(cv.buffer), accel_device_ptr(z.buffer), reduce=reduce)
            z.accel_used(True)
            ref = PixelData(d, np.float64, n_value=3)
            ref.raw[:] = total
            covariance_apply(dev_map(spd, 6, "cov2"), ref)
            np.testing.assert_allclose(z.data, ref.data, rtol=1e-13, atol=1e-13 * np.max(np.abs(total)), err_msg=mode)
            cv.accel_delete()
            z.accel_delete()
        # a map large enough for the slices to be used (>= S * ranks * 64 pixels): 40 000 pixels, one value
        big = np.random.default_rng(77 + rank).standard_normal(40000)
        tot = np.zeros(40000)
        for r in range(size):
            tot += np.random.default_rng(77 + r).standard_normal(40000)
        t = torch.from_numpy(big).cuda()
        capi.dev.comm_map_reduce_apply(40000, 1, 0, t.data_ptr(), reduce=True)
        torch.cuda.synchronize()
        np.testing.assert_allclose(t.cpu().numpy(), tot, rtol=0, atol=1e-13 * np.max(np.abs(tot)), err_msg=mode)
        if mode.startswith("peer"):
            # an odd number of values (8-byte lane accesses, a short last slice), then twice in a row on a smaller map
            # (the exchange buffers are reused: the second reduction must not see anything of the first)
            for n_odd in (40101, 997, 997):
                vals = {r: np.random.default_rng(500 + r + n_odd).standard_normal(n_odd) for r in range(size)}
                t = torch.from_numpy(vals[rank]).cuda()
                capi.dev.comm_map_reduce_apply(n_odd, 1, 0, t.data_ptr(), reduce=True)
                capi.dev.comm_map_reduce_apply(n_odd, 1, 0, t.data_ptr(), reduce=True)     # sum of sums: x size
                torch.cuda.synchronize()
                tot = np.zeros(n_odd)
                for r in range(size):
                    tot += vals[r]
                np.testing.assert_allclose(t.cpu().numpy(), size * tot, rtol=0, atol=1e-12 * np.max(np.abs(tot)), err_msg=mode)
            # bit-identical on every rank, whatever the rank order of arrival: only the owner adds, in rank order
            got = t.cpu().numpy().copy()
            ref_bits = torch.from_numpy(got).cuda()
            capi.dev.comm_broadcast(ref_bits.data_ptr(), got.size, np.float64, 0)
            torch.cuda.synchronize()
            assert np.array_equal(ref_bits.cpu().numpy(), got), "peer mode: ranks disagree in the last bit"
            n_red, n_est, n_bytes = capi.dev.comm_peer_stats()
            n_red, n_est = n_red - peer_before[0], n_est - peer_before[1]
            if size > 1:
                # 2 small maps + 1 of 40 000 + 3 x 2 odd ones went through the exchange buffers, which grew twice
                # (1776 x 3 values -> 40 000 -> 40 101; the second mode finds them large enough) and hold (1 + size)
                # slots of ceil(40 101 / size) values
                assert n_red == 9 and n_est == (3 if first_peer else 0), (mode, width, n_red, n_est)
                assert n_bytes >= (1 + size) * 8 * (40101 // size), n_bytes
                # the memory kind of the exchange buffers is the one asked for (TOAST_HIP_COMM_PEER_MEM; default coarse)
                want_mem = "fine" if os.environ.get("TOAST_HIP_COMM_PEER_MEM", "coarse") == "fine" else "coarse"
                assert capi.dev.comm_peer_mem() == want_mem, (capi.dev.comm_peer_mem(), want_mem)
            else:
                assert n_red == 0 and n_est == 0 and n_bytes == 0
            first_peer = False
            # a call that sums nothing (reduce = 0: covariance_apply(use_alltoallv=True)) right behind
            # another call.  Without the first barrier in the reduce = 0 call a fast rank rewrites the slice it owns while a
            # slower peer still reads the previous result from it.  The last rank is made slow (a long kernel queue on its
            # stream before every pair of calls); every rank checks BOTH results, five rounds, on a map large enough for
            # the pull to take a while.
            n_big = 1 << 20
            for rnd in range(5):
                vals = {r: np.random.default_rng(7000 + 10 * rnd + r).standard_normal(n_big) for r in range(size)}
                same = np.random.default_rng(7900 + rnd).standard_normal(n_big)
                ta, tb = torch.from_numpy(vals[rank]).cuda(), torch.from_numpy(same).cuda()
                if rank == size - 1:
                    busy = torch.ones(1 << 22, device="cuda")
                    for _ in range(40):
                        busy = busy * 1.0000001
                capi.dev.comm_map_reduce_apply(n_big, 1, 0, ta.data_ptr(), reduce=True)
                capi.dev.comm_map_reduce_apply(n_big, 1, 0, tb.data_ptr(), reduce=False)
                capi.dev.comm_check(0)
                tot = np.zeros(n_big)
                for r in range(size):
                    tot += vals[r]
                np.testing.assert_allclose(ta.cpu().numpy(), tot, rtol=0, atol=1e-12 * np.max(np.abs(tot)), err_msg=mode)
                assert np.array_equal(tb.cpu().numpy(), same), (mode, width, rnd)
            # the plain all-reduce of a map of doubles takes the same exchange in these modes (bench.py's timed step)
            before = capi.dev.comm_peer_stats()[0]
            t = torch.from_numpy(vals[rank]).cuda()
            capi.dev.comm_allreduce(t.data_ptr(), n_big, np.float64, "sum")
            torch.cuda.synchronize()
            np.testing.assert_allclose(t.cpu().numpy(), tot, rtol=0, atol=1e-12 * np.max(np.abs(tot)), err_msg=mode)
            assert capi.dev.comm_peer_stats()[0] == before + (1 if size > 1 else 0)
    # ranks that hold DIFFERENT local submaps (the reference's general case): the default exchange on the device through
    # the union of all ranks' submaps, against the sums computed by hand; rank 0 also tries it with its copy on the host
    for dtype, n_value in ((np.float64, 3), (np.int64, 1)):
        mine = np.arange(2 + 3 * rank, 2 + 3 * rank + 7) % 64          # 7 submaps, shifted by 3 per rank: neighbours overlap
        mine = np.unique(mine)
        d2 = PixelDistribution(n_pix=64 * 48, n_submap=64, local_submaps=mine, comm=comm)
        assert not d2.replicated or size == 1
        vals = {r: (np.random.default_rng(900 + r).standard_normal((7, 48, n_value)) * 100).astype(dtype) for r in range(size)}
        subs = {r: np.unique(np.arange(2 + 3 * r, 2 + 3 * r + 7) % 64) for r in range(size)}
        want_u = np.zeros((mine.size, 48, n_value), dtype=dtype)
        for i, sm in enumerate(mine):
            for r in range(size):
                hit = np.flatnonzero(subs[r] == sm)
                if hit.size:
                    want_u[i] += vals[r][hit[0]]
        for on_device in (True, False):
            pd2 = PixelData(d2, dtype, n_value=n_value)
            pd2.raw[:] = vals[rank][: mine.size].reshape(-1)
            if on_device or rank != 0:
                pd2.accel_create("different_submaps")
                pd2.accel_update_device()
            was = pd2.accel_in_use()
            pd2.sync_alltoallv()
            assert pd2.accel_in_use() == was
            if dtype == np.float64:
                np.testing.assert_allclose(pd2.data, want_u, rtol=1e-13, atol=1e-11)
            else:
                assert np.array_equal(pd2.data, want_u)
            if pd2.accel_exists():
                pd2.accel_delete()
    # MIXED residency: rank 0 has "evicted" its map -- it holds it on the host -- while the others hold
    # theirs on the device.  The route does not depend on where a rank's copy lives, so every rank enters the same
    # collective on the same communicator: no hang, the right sums, and each copy ends where it started.
    for entry in ("sync_allreduce", "sync_alltoallv", "map_reduce_apply", "covariance_apply", "covariance_invert"):
        z = PixelData(d, np.float64, n_value=3)
        z.raw[:] = parts[rank]
        cvm = dev_map(spd, 6, "cov_mixed")
        on_device = rank != 0
        if on_device:
            z.accel_create("zmap_mixed")
            z.accel_update_device()
        if entry in ("sync_allreduce", "sync_alltoallv"):
            getattr(z, entry)()
            expect = total
        elif entry == "map_reduce_apply":
            map_reduce_apply(cvm, z, sync_type="alltoallv")
            ref = PixelData(d, np.float64, n_value=3)
            ref.raw[:] = total
            covariance_apply(dev_map(spd, 6, "cov_mixed_ref"), ref)
            expect = ref.raw
        elif entry == "covariance_apply":
            z.raw[:] = total
            if on_device:
                z.accel_update_device()
            covariance_apply(cvm, z, use_alltoallv=True)
            ref = PixelData(d, np.float64, n_value=3)
            ref.raw[:] = total
            covariance_apply(dev_map(spd, 6, "cov_mixed_ref"), ref)
            expect = ref.raw
        else:
            inv = PixelData(d, np.float64, n_value=6)
            inv.raw[:] = np.asarray(spd).reshape(-1)
            if on_device:
                inv.accel_create("inv_mixed")
                inv.accel_update_device()
            covariance_invert(inv, 1e-6, use_alltoallv=True)
            assert inv.accel_in_use() == on_device, entry
            np.testing.assert_allclose(inv.data, want["inv"], rtol=1e-13, atol=1e-15, err_msg=entry)
            if on_device:
                inv.accel_delete()
            cvm.accel_delete()
            continue
        assert z.accel_in_use() == on_device, entry
        np.testing.assert_allclose(z.data.reshape(-1), np.asarray(expect).reshape(-1), rtol=1e-13,
                                   atol=1e-13 * np.max(np.abs(total)), err_msg=entry)
        if on_device:
            z.accel_delete()
        cvm.accel_delete()
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
