"""Worker for tests/test_dist_gloo.py: run under torch.distributed.run with world_size 2 and
the gloo backend (CPU).  Exercises every collective of the N>1 path: the union of hit
submaps, the map all-reduce of PixelData, scalar reductions of amplitude dot products and the
detector-sharded data layout.  Exits non-zero on any mismatch."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from toast_amd.data import Comm  # noqa: E402
from toast_amd.pixels import PixelData, PixelDistribution, unify_local_submaps  # noqa: E402
from toast_amd.sim import create_satellite_data  # noqa: E402
from toast_amd.templates import Amplitudes  # noqa: E402


def main():
    dist.init_process_group("gloo")
    comm = Comm()
    rank, size = comm.world_rank, comm.world_size
    assert size == 2 and comm.comm_world is not None

    # 1. union of hit submaps -> identical distribution everywhere
    hits = np.zeros(16, dtype=np.uint8)
    hits[[1, 4] if rank == 0 else [4, 9, 15]] = 1
    union = unify_local_submaps(hits, comm)
    assert list(np.flatnonzero(union)) == [1, 4, 9, 15], union
    d = PixelDistribution(n_pix=16 * 48, n_submap=16, local_submaps=np.flatnonzero(union), comm=comm)

    # 2. map all-reduce (host path; the device path is the same call on an RCCL tensor)
    pd = PixelData(d, np.float64, n_value=3)
    rng = np.random.default_rng(100 + rank)
    mine = rng.standard_normal(pd.raw.size)
    pd.raw[:] = mine
    pd.sync_allreduce()
    other = np.random.default_rng(100 + (1 - rank)).standard_normal(pd.raw.size)
    np.testing.assert_allclose(pd.raw, mine + other, rtol=0, atol=1e-15)
    ph = PixelData(d, np.int64, n_value=1)
    ph.raw[:] = rank + 1
    ph.sync_allreduce()
    assert np.all(ph.raw == 3)
    # sync_alltoallv: same result (reference tests/ops_mapmaker_utils.py:211-397)
    pa = PixelData(d, np.float64, n_value=3)
    pa.raw[:] = mine
    pa.sync_alltoallv()
    assert np.array_equal(pa.raw, pd.raw)

    # ... also when the flat map does not divide by the number of ranks (padded shards of the reduce-scatter)
    d_odd = PixelDistribution(n_pix=16 * 45, n_submap=16, local_submaps=np.array([1, 4, 9]), comm=comm)
    po = PixelData(d_odd, np.float64, n_value=1)
    assert po.raw.size % size == 1
    mine_o = np.random.default_rng(300 + rank).standard_normal(po.raw.size)
    po.raw[:] = mine_o
    po.sync_alltoallv()
    other_o = np.random.default_rng(300 + (1 - rank)).standard_normal(po.raw.size)
    np.testing.assert_allclose(po.raw, mine_o + other_o, rtol=0, atol=1e-15)

    # 2b. ownership, statistics and broadcast of a distribution whose ranks hold DIFFERENT submaps
    # (reference src/toast/pixels.py:176-315, :972-1184)
    mine_sm = np.array([1, 4, 9] if rank == 0 else [4, 9, 12, 15])
    dd = PixelDistribution(n_pix=16 * 48 - 5, n_submap=16, local_submaps=mine_sm, comm=comm)
    assert list(dd.all_hit_submaps) == [1, 4, 9, 12, 15]
    owners = dd.submap_owners
    assert list(owners[[1, 4, 9, 12, 15]]) == [0, 0, 0, 1, 1] and np.all(owners[[0, 2, 3, 5]] == -1)
    assert list(dd.owned_submaps) == ([1, 4, 9] if rank == 0 else [12, 15])
    full = [np.arange(dd.n_pix, dtype=np.float64) * (k + 1) for k in range(2)]
    pm = PixelData(dd, np.float64, n_value=2)
    pm.broadcast_map(full if rank == 0 else None, comm_bytes=3 * 48 * 2 * 8)
    for loc, sm in enumerate(mine_sm):
        want = np.zeros((48, 2))
        n_in = min(48, dd.n_pix - sm * 48)
        for k in range(2):
            want[:n_in, k] = full[k][sm * 48:sm * 48 + n_in]
        assert np.array_equal(pm.data[loc], want), (rank, sm)
    st = pm.stats()
    # every hit submap counted once; the reference divides by ALL n_submap * n_pix_submap pixels
    ref = np.zeros((16, 48, 2))
    for sm in (1, 4, 9, 12, 15):
        n_in = min(48, dd.n_pix - sm * 48)
        for k in range(2):
            ref[sm, :n_in, k] = full[k][sm * 48:sm * 48 + n_in]
    if rank == 0:
        for k in range(2):
            assert abs(st["sum"][k] - ref[:, :, k].sum()) < 1e-9 * abs(ref[:, :, k].sum())
            assert abs(st["mean"][k] - ref[:, :, k].mean()) < 1e-12 * abs(ref[:, :, k].mean())
            assert abs(st["rms"][k] - np.std(ref[:, :, k], ddof=1)) < 1e-12 * np.std(ref[:, :, k])
    else:
        assert st is None
    assert list(dd.global_pixel_to_local(np.array([48, 4 * 48 + 7]))) == ([0, 48 + 7] if rank == 0 else [-48, 7])

    # 2c. owner-computes exchange with DIFFERENT local submaps per rank (reference pixels.py:317-414, 780-967)
    sc, sd, rc, rd, rloc = dd.alltoallv_info
    if rank == 0:       # holds 1, 4, 9 (all owned by rank 0); owns 1, 4, 9, of which rank 1 holds 4 and 9
        assert list(sc) == [3, 0] and list(sd) == [0, 3] and list(rc) == [3, 2] and list(rd) == [0, 3]
        assert {k: list(v) for k, v in rloc.items()} == {1: [0], 4: [1, 3], 9: [2, 4]}
    else:               # holds 4, 9 (owner 0) and 12, 15 (its own)
        assert list(sc) == [2, 2] and list(sd) == [0, 2] and list(rc) == [0, 2] and list(rd) == [0, 0]
        assert {k: list(v) for k, v in rloc.items()} == {12: [0], 15: [1]}
    assert not dd.replicated and d.replicated
    pv = PixelData(dd, np.float64, n_value=2)
    for loc, sm in enumerate(mine_sm):
        pv.data[loc] = 100.0 * sm + (rank + 1)
    pv.sync_alltoallv()                               # default local_func: every submap = sum of its copies
    for loc, sm in enumerate(mine_sm):
        holders = [r for r, sms in enumerate(([1, 4, 9], [4, 9, 12, 15])) if sm in sms]
        assert np.all(pv.data[loc] == sum(100.0 * sm + (r + 1) for r in holders)), (rank, sm)
    seen = {}

    def negate_first_copy(n_submap_value, receive_locations, receive, reduce_buf):
        # the reference's calling convention: every owned submap with the offsets of its copies in `receive`
        for sm, locs in receive_locations.items():
            seen[sm] = len(locs)
            reduce_buf[:] = -receive[locs[0]:locs[0] + n_submap_value]
            for lc in locs:
                receive[lc:lc + n_submap_value] = reduce_buf

    before = pv.data.copy()
    pv.sync_alltoallv(local_func=negate_first_copy)
    assert np.array_equal(pv.data, -before)
    assert seen == ({1: 1, 4: 2, 9: 2} if rank == 0 else {12: 1, 15: 1})

    try:
        pv.sync_alltoallv(bogus=1)
        raise SystemExit("unknown keyword was accepted")
    except TypeError:
        pass

    # 3. amplitude dot products: local dot + scalar all-reduce
    a = Amplitudes(comm, 10, 5)
    a.local[:] = np.arange(5) + 5 * rank
    val = a.dot(a)
    assert abs(val - float(np.sum(np.arange(10) ** 2))) < 1e-12, val
    # full copies on every process (n_local == n_global): the dot product is not reduced, sync() sums the copies
    # with flagged values contributing zero (reference templates/amplitudes.py:357-470, :545-554)
    f = Amplitudes(comm, 6, 6)
    g = Amplitudes(comm, 6, 6)
    f.local[:] = np.arange(6) + 1.0
    g.local[:] = 2.0
    f.local_flags[1] = 1
    assert abs(f.dot(g) - 2.0 * (21.0 - 2.0)) < 1e-12
    assert f.n_local_flagged == 1
    f.sync()
    assert np.allclose(f.local, [2.0, 0.0, 6.0, 8.0, 10.0, 12.0])
    f.reset_flags()
    assert f.n_local_flagged == 0 and f.use_group is False and f.local_indices is None
    try:
        Amplitudes(comm, 10, 4)
        raise SystemExit("inconsistent n_local was accepted")
    except RuntimeError:
        pass
    assert comm.allreduce_scalar(rank + 1, op="max") == 2
    assert comm.allreduce_scalar(3, op="sum") == 6

    # 4. detector sharding: each rank gets its own detectors of one focalplane, same scan
    data = create_satellite_data(comm=comm, n_det=2, total_det=4, first_det=2 * rank, n_samp=200)
    names = data.obs[0].local_detectors
    all_names = [None, None]
    dist.all_gather_object(all_names, names)
    assert len(set(all_names[0]) | set(all_names[1])) == 4 and not (set(all_names[0]) & set(all_names[1]))
    bore = data.obs[0].shared["boresight_radec"].data.copy()
    comm.allreduce_array_(bore)
    np.testing.assert_allclose(bore, 2 * data.obs[0].shared["boresight_radec"].data)

    comm.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} OK")


if __name__ == "__main__":
    main()
