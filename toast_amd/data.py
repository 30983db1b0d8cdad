"""Minimal data model honouring the reference's buffer layouts -- only what the map-making
operators touch (SURVEY.md §2.1 marks the full data model out of scope):

* ``DetectorData`` / ``DetDataManager``  src/toast/observation_data.py:25-860
  (one contiguous ``[n_det, n_samp, ...]`` buffer per key, row lookup with ``indices()``,
  ``ensure()`` re-using allocations, accel_* staging keyed by the buffer's base pointer)
* ``SharedData``                         src/toast/observation_data.py:1200-  (per-observation arrays)
* ``IntervalList``                       src/toast/intervals.py:26-300 (32-byte records)
* ``Observation`` / ``Data``             src/toast/observation.py, src/toast/data.py
* ``Comm``                               src/toast/mpi.py:113-275, backed by torch.distributed
  (RCCL on GPUs, gloo on CPU) instead of mpi4py.
"""

import os
import types
from collections.abc import MutableMapping

import numpy as np

from .accel import (
    AcceleratorObject,
    accel_data_create,
    accel_data_delete,
    accel_data_present,
    accel_data_reset,
    accel_data_update_device,
    accel_data_update_host,
    accel_enabled,
)
from .synth import interval_dtype

#: default names and flag masks (reference: src/toast/observation.py:44-111)
defaults = types.SimpleNamespace(
    times="times",
    shared_flags="flags",
    det_data="signal",
    det_flags="flags",
    hwp_angle="hwp_angle",
    boresight_radec="boresight_radec",
    boresight_azel="boresight_azel",
    azimuth="azimuth",
    elevation="elevation",
    scanning_interval="scanning",
    turnaround_interval="turnaround",
    throw_leftright_interval="throw_leftright",
    throw_rightleft_interval="throw_rightleft",
    elnod_interval="elnod",
    pixels="pixels",
    weights="weights",
    quats="quats",
    noise_model="noise_model",
    shared_mask_invalid=1,
    shared_mask_processing=2,
    shared_mask_unstable_scanrate=4,
    shared_mask_irregular=8,
    shared_mask_nonscience=1 | 2 | 4 | 8,
    det_mask_invalid=1,
    det_mask_processing=2,
    det_mask_sso=4,
    det_mask_nonscience=1 | 2 | 4,
    det_data_units="K",
)


# ----------------------------------------------------------------------------- communicator
class Comm:
    """Process layout.  One process per GPU; ``comm_world`` is None for a single process,
    else this object (it provides the few collectives the path needs)."""

    def __init__(self, use_dist=None, single_rank_collectives=False):
        import torch.distributed as dist

        self._dist = dist if (dist.is_available() and dist.is_initialized()) else None
        if use_dist is False:
            self._dist = None
        # tests only: issue the collectives even in a one-rank process group, so that the RCCL code
        # path (device tensors, stream ordering) runs on a single-GPU box
        self._single_rank_collectives = bool(single_rank_collectives)
        self.world_rank = self._dist.get_rank() if self._dist else 0
        self.world_size = self._dist.get_world_size() if self._dist else 1
        self.group_rank = self.world_rank
        self.group_size = self.world_size
        self.group = 0
        self.ngroups = 1

    @property
    def comm_world(self):
        if self._dist is None:
            return None
        return self if (self.world_size > 1 or self._single_rank_collectives) else None

    comm_group = comm_world

    # -- collectives
    def allreduce_scalar(self, value, op="sum"):
        """All-reduce one Python number (``Amplitudes.dot`` / nnz agreement)."""
        if self.comm_world is None:
            return value
        import torch

        ops = {"sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX, "min": self._dist.ReduceOp.MIN}
        dev = self._collective_device()
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        self._dist.all_reduce(t, op=ops[op])
        return type(value)(t.item()) if isinstance(value, (int, np.integer)) else float(t.item())

    def _collective_device(self):
        import torch

        backend = self._dist.get_backend()
        return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    def allreduce_tensor_(self, tensor, op="sum"):
        """In-place all-reduce of a torch tensor (device tensor -> RCCL over xGMI)."""
        if self.comm_world is None:
            return tensor
        ops = {"sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX, "min": self._dist.ReduceOp.MIN}
        self._dist.all_reduce(tensor, op=ops[op])
        return tensor

    def allreduce_array_(self, arr, op="sum"):
        """In-place all-reduce of a host NumPy array."""
        if self.comm_world is None:
            return arr
        import torch

        t = torch.from_numpy(arr)
        if self._dist.get_backend() == "nccl":
            d = t.to(self._collective_device())
            self.allreduce_tensor_(d, op)
            t.copy_(d.cpu())
        else:
            self.allreduce_tensor_(t, op)
        return arr

    def bcast_array_(self, arr, root=0):
        """In-place broadcast of a host NumPy array from ``root``."""
        if self.comm_world is None:
            return arr
        import torch

        t = torch.from_numpy(arr)
        if self._dist.get_backend() == "nccl":
            d = t.to(self._collective_device())
            self._dist.broadcast(d, src=root)
            t.copy_(d.cpu())
        else:
            self._dist.broadcast(t, src=root)
        return arr

    def allgather_array(self, arr):
        """Host NumPy array of the same shape on every rank -> array [world_size, ...] of all of them."""
        if self.comm_world is None:
            return np.asarray(arr)[None]
        import torch

        dev = self._collective_device()
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        out = [torch.empty_like(t) for _ in range(self.world_size)]
        self._dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])

    def alltoallv_array(self, send, send_counts, send_displ, recv, recv_counts, recv_displ):
        """MPI Alltoallv on host NumPy arrays (element counts / displacements per rank); the pieces of ``send`` and of
        ``recv`` must lie in rank order without gaps, which is how PixelDistribution.alltoallv_info builds them."""
        import torch

        for cnt, dsp in ((send_counts, send_displ), (recv_counts, recv_displ)):
            if not np.array_equal(np.cumsum(cnt) - cnt, dsp):
                raise RuntimeError("alltoallv_array: pieces must be contiguous and in rank order")
        n_send, n_recv = int(np.sum(send_counts)), int(np.sum(recv_counts))
        dev = self._collective_device()
        s = torch.from_numpy(send[:n_send]).to(dev)
        r = torch.empty(n_recv, dtype=s.dtype, device=dev)
        self._dist.all_to_all_single(r, s, [int(c) for c in recv_counts], [int(c) for c in send_counts])
        recv[:n_recv] = r.cpu().numpy()
        return recv

    def device_comm(self):
        """True when the collectives of device-resident data go through the library's own RCCL communicator
        (``toast_hip_comm_*``: enqueued on the kernels' stream, no host synchronisation).  That is the case when the
        process group says one GPU per rank (backend "nccl"; ``TOAST_HIP_COMM=rccl`` forces it, ``=torch`` turns it
        off).  The communicator is created on first use: rank 0 draws the RCCL unique id, the process group carries it
        to the other ranks."""
        if self.comm_world is None:
            return False
        state = getattr(self, "_device_comm", None)
        if state is not None:
            return state
        import os

        from . import capi
        from .accel import accel_enabled

        want = os.environ.get("TOAST_HIP_COMM", "")
        ok = accel_enabled() and want != "torch" and (self._dist.get_backend() == "nccl" or want == "rccl")
        if ok:
            n, r, _ = capi.dev.comm_info()
            if n == 0:
                # every rank must be able to load RCCL before anybody enters the collective initialisation
                able = int(self.allreduce_scalar(1 if capi.dev.comm_available() else 0, op="min"))
                uid = np.zeros(129, dtype=np.uint8)
                if able and self.world_rank == 0:
                    try:
                        uid[:128] = np.frombuffer(capi.dev.comm_unique_id(), dtype=np.uint8)
                        uid[128] = 1
                    except RuntimeError:
                        pass
                self.bcast_array_(uid, root=0)
                mine = 0
                if uid[128] == 1:
                    try:
                        capi.dev.comm_init(uid[:128].tobytes(), self.world_size, self.world_rank)
                        mine = 1
                    except RuntimeError:
                        mine = 0
                # used only if EVERY rank has a communicator; otherwise the collectives of device-resident data go
                # through the process group (host staging for gloo, torch's own RCCL for nccl)
                ok = bool(int(self.allreduce_scalar(mine, op="min")))
                if not ok and mine:
                    capi.dev.comm_destroy()
            elif (n, r) != (self.world_size, self.world_rank):
                raise RuntimeError(f"the library's communicator is rank {r} of {n}, the process group says "
                                   f"{self.world_rank} of {self.world_size}")
        self._device_comm = ok
        return ok

    def barrier(self):
        if self.comm_world is not None:
            self._dist.barrier()


# ----------------------------------------------------------------------------- intervals
def build_interval_dtype():
    """The Interval record (reference src/toast/intervals.py:26-45; same layout as ``synth.interval_dtype``)."""
    return interval_dtype


def regular_intervals(n, start, first, rate, duration, gap):
    """``n`` intervals of ``duration`` seconds separated by ``gap`` seconds, as a raw record array (reference
    src/toast/intervals.py:449-520): both lengths are rounded down to whole samples, except that a span that is an
    exact multiple of the sampling excludes its final sample."""
    invrate = 1.0 / rate

    def whole(span):
        lower = int(span * rate)
        return lower + 1 if np.absolute(lower * invrate - span) > 1.0e-12 else lower

    totsamples = whole(duration + gap)
    dursamples = whole(duration)
    out = np.zeros(n, dtype=interval_dtype).view(np.recarray)
    for i in range(n):
        out[i].first = first + i * totsamples
        out[i].last = out[i].first + dursamples
        out[i].start = start + i * (totsamples * invrate)
        out[i].stop = out[i].start + (dursamples * invrate)
    return out


class IntervalList:
    """Sorted, disjoint sample spans ``[first, last)`` with their times, valid inside the local ``timestamps``; ``.data``
    is the record array the kernels consume.  Construction from an existing list (``intervals``), from time spans or
    from sample spans, set algebra (``~ & |``) and ``simplify`` follow the reference (src/toast/intervals.py:48-403);
    ``data=`` adopts a ready record array (used by the synthetic inputs)."""

    def __init__(self, timestamps=None, intervals=None, timespans=None, samplespans=None, data=None):
        self.timestamps = timestamps
        empty = np.zeros(0, dtype=interval_dtype).view(np.recarray)
        if data is not None:
            self.data = np.ascontiguousarray(data, dtype=interval_dtype).view(np.recarray)
        elif intervals is not None:
            if timespans is not None or samplespans is not None:
                raise RuntimeError("If constructing from intervals, other spans should be None")
            if len(intervals) == 0:
                self.data = empty
            else:
                self.data = self._from_timespans([(x.start, x.stop) for x in intervals])
        elif timespans is not None:
            if samplespans is not None:
                raise RuntimeError("Cannot construct from both time and sample spans")
            if len(timespans) == 0:
                self.data = empty
            else:
                spans = np.vstack(timespans).astype(np.float64)
                for k in range(len(spans) - 1):
                    if np.isclose(spans[k][1], spans[k + 1][0], rtol=1e-12):
                        spans[k][1] = spans[k + 1][0]      # nearly equal times are made equal
                    if spans[k][1] > spans[k + 1][0]:
                        raise RuntimeError("Timespans must be sorted and disjoint")
                self.data = self._from_timespans(spans)
        elif samplespans is not None:
            spans = list(samplespans)
            for k in range(len(spans) - 1):
                if spans[k][1] > spans[k + 1][0]:
                    raise RuntimeError("Sample spans must be sorted and disjoint")
            if timestamps is None or len(timestamps) == 0:
                # no time axis (kernel-level callers): sample spans only
                self.data = np.zeros(len(spans), dtype=interval_dtype).view(np.recarray)
                for k, (first, last) in enumerate(spans):
                    self.data[k].first, self.data[k].last = first, last
            else:
                rows = []
                n = len(timestamps)
                for first, last in spans:
                    if last < 0 or first >= n:
                        continue
                    first = max(first, 0)
                    last = min(last, n)
                    rows.append((self._sample_time(first), self._sample_time(last), first, last))
                self.data = np.array(rows, dtype=interval_dtype).view(np.recarray) if rows else empty
        else:
            self.data = empty

    def _sample_time(self, sample):
        n = len(self.timestamps)
        if sample < 0 or sample > n:
            raise RuntimeError(f"Invalid sample index: {sample} not in [0, {n}]")
        return self.timestamps[sample - 1] if sample == n else self.timestamps[sample]

    def _from_timespans(self, timespans):
        """start <= t < stop, the last timestamp included when an interval stops exactly there (intervals.py:150-175);
        spans outside the local timestamps are dropped."""
        t = self.timestamps
        start, stop = np.vstack(timespans).T
        good = np.logical_and(start < t[-1], stop > t[0])
        start, stop = start[good], stop[good]
        first = np.searchsorted(t, start, side="left")
        last = np.searchsorted(t, stop, side="left")
        last[last == len(t) - 1] = len(t)
        out = np.zeros(len(start), dtype=interval_dtype).view(np.recarray)
        out.start, out.stop, out.first, out.last = start, stop, first, last
        return out

    def __getitem__(self, key):
        return self.data[key]

    def __delitem__(self, key):
        raise RuntimeError("Cannot delete individual elements from an IntervalList")

    def __contains__(self, item):
        return any(ival == item for ival in self.data)

    def __len__(self):
        return len(self.data)

    def __iter__(self):
        return iter(self.data)

    def __repr__(self):
        s = "<IntervalList [\n"
        for ival in self.data:
            s += f" {ival.start:15.3f} - {ival.stop:15.3f} ({ival.first:9} - {ival.last:9}),\n"
        return s + "]>"

    def _same_times(self, other):
        a, b = self.timestamps, other.timestamps
        if a is None or b is None:
            return a is None and b is None
        return (len(a) == len(b) and bool(np.isclose(a[0], b[0], rtol=1e-12))
                and bool(np.isclose(a[-1], b[-1], rtol=1e-12)))

    def __eq__(self, other):
        if not isinstance(other, IntervalList) or len(self.data) != len(other) or not self._same_times(other):
            return False
        return np.array_equal(self.data.first, other.data.first) and np.array_equal(self.data.last, other.data.last)

    def __ne__(self, other):
        return not self.__eq__(other)

    def simplify(self):
        """Merge intervals that touch (intervals.py:225-252)."""
        if len(self.data) == 0:
            return
        rows = [list(self.data[0].tolist())]
        for cur in self.data[1:]:
            if cur.first == rows[-1][3]:
                rows[-1][1], rows[-1][3] = cur.stop, cur.last
            else:
                rows.append(list(cur.tolist()))
        if len(rows) < len(self.data):
            self.data = np.array([tuple(r) for r in rows], dtype=interval_dtype).view(np.recarray)

    def __invert__(self):
        """The gaps, including the ranges before the first and after the last interval (intervals.py:254-283)."""
        if len(self.data) == 0:
            return
        t, d = self.timestamps, self.data
        neg = []
        if not np.isclose(t[0], d[0].start, rtol=1e-12):
            neg.append((t[0], d[0].start, 0, d[0].first))
        for k in range(len(d) - 1):
            if d[k + 1].first != d[k].last + 1:
                neg.append((d[k].stop, d[k + 1].start, d[k].last, d[k + 1].first))
        if not np.isclose(t[-1], d[-1].stop, rtol=1e-12):
            neg.append((d[-1].stop, t[-1], d[-1].last, len(t)))
        return IntervalList(t, intervals=np.array(neg, dtype=interval_dtype).view(np.recarray))

    def _check_times(self, other, what):
        if not self._same_times(other):
            raise RuntimeError(f"Cannot do {what} operation on intervals with different timestamps")

    def __and__(self, other):
        """Intersection (intervals.py:285-318)."""
        self._check_times(other, "AND")
        if len(self.data) == 0 or len(other) == 0:
            return IntervalList(self.timestamps)
        result = []
        a = b = 0
        while a < len(self.data) and b < len(other):
            x, y = self.data[a], other[b]
            start, stop = max(x.start, y.start), min(x.stop, y.stop)
            if start < stop:
                result.append((start, stop, max(x.first, y.first), min(x.last, y.last)))
            if x.stop < y.stop:
                a += 1
            else:
                b += 1
        return IntervalList(self.timestamps, intervals=np.array(result, dtype=interval_dtype).view(np.recarray))

    def __or__(self, other):
        """Union; intervals that merely touch stay separate, ``simplify`` joins them (intervals.py:320-402)."""
        self._check_times(other, "OR")
        if len(self.data) == 0:
            return IntervalList(self.timestamps, intervals=other.data)
        if len(other) == 0:
            return IntervalList(self.timestamps, intervals=self.data)
        # merge the two sorted sequences by first sample (ties: the other list first, like the reference's walk)
        a = b = 0
        merged = []
        while a < len(self.data) or b < len(other):
            if b >= len(other) or (a < len(self.data) and self.data[a].first < other[b].first):
                merged.append(self.data[a])
                a += 1
            else:
                merged.append(other[b])
                b += 1
        result = []
        cur = list(merged[0].tolist())
        for nxt in merged[1:]:
            if nxt.first < cur[3]:
                if nxt.last > cur[3]:
                    cur[1], cur[3] = nxt.stop, nxt.last
            else:
                result.append(tuple(cur))
                cur = list(nxt.tolist())
        result.append(tuple(cur))
        return IntervalList(self.timestamps, intervals=np.array(result, dtype=interval_dtype).view(np.recarray))


class IntervalsManager(dict):
    """``ob.intervals[name]``; the key ``None`` is the whole observation."""

    def __init__(self, n_samp, timestamps=None):
        super().__init__()
        self._n_samp = n_samp
        self._times = timestamps
        self[None] = IntervalList(timestamps, samplespans=[(0, n_samp)])

    def create(self, name, samplespans):
        self[name] = IntervalList(self._times, samplespans=samplespans)


# ----------------------------------------------------------------------------- detector data
class DetectorData(AcceleratorObject):
    """One contiguous buffer ``[n_det, n_samp] + sample_shape`` for a list of detectors."""

    def __init__(self, detectors, shape, dtype, units=None):
        super().__init__("DetectorData")
        self._set_detectors(detectors)
        self._sample_shape = tuple(shape[1:])
        self._n_samp = int(shape[0])
        self._dtype = np.dtype(dtype)
        self.units = units
        full = (len(self._detectors), self._n_samp) + self._sample_shape
        self._raw = np.zeros(int(np.prod(full)) if len(self._detectors) else 0, dtype=self._dtype)
        self._capacity = self._raw.size
        self._data = self._raw[: int(np.prod(full))].reshape(full)

    def _set_detectors(self, detectors):
        self._detectors = list(detectors)
        self._name2idx = {d: i for i, d in enumerate(self._detectors)}

    @property
    def detectors(self):
        return list(self._detectors)

    keys = detectors.fget

    @property
    def dtype(self):
        return self._dtype

    @property
    def shape(self):
        return self._data.shape

    @property
    def detector_shape(self):
        return self._data.shape[1:]

    @property
    def data(self):
        """The full buffer on the HOST.  Lazy coherence: when the device copy is the current
        one (Pipelines leave their detector-data products resident), it is copied back first and
        the host becomes the current side again."""
        if self._accel_used:
            self.accel_update_host()
        return self._data

    @property
    def buffer(self):
        """The same view without synchronisation: its base pointer is the accelerator key.  For
        device-side code paths only -- the contents may be stale."""
        return self._data

    def arg(self, use_accel):
        """Array to hand to a kernel call: the key view when the kernel runs on registered
        device memory, the (synchronised) host contents when the call is host-staged."""
        return self._data if use_accel else self.data

    def indices(self, names):
        """Rows of the given detectors, int32 (observation_data.py:171-195)."""
        return np.array([self._name2idx[x] for x in names], dtype=np.int32)

    def change_detectors(self, detectors):
        """Re-use the allocation for a different detector list when it is not longer, else
        reallocate (observation_data.py:248-325).  Host (and device) contents are zeroed."""
        detectors = list(detectors)
        if detectors == self._detectors:
            return
        n_new = len(detectors)
        need = n_new * self._n_samp * int(np.prod(self._sample_shape, dtype=np.int64))
        if need > self._capacity:
            if self.accel_exists():
                self.accel_delete()
            self._raw = np.zeros(need, dtype=self._dtype)
            self._capacity = need
        elif need == self._data.size:
            # same footprint (e.g. one-detector buffers recycled by SINGLE pipelines): keep the
            # host allocation and the device copy, zero both (the host side only when it is the
            # current one: a stale host buffer is overwritten by the next update_host anyway)
            if not self._accel_used:
                self._raw[:] = 0
            if self.accel_exists():
                self.accel_reset()
        else:
            # the device copy is keyed by the *view*, whose size changes: drop it
            if self.accel_exists():
                self.accel_delete()
            self._raw[:] = 0
        self._set_detectors(detectors)
        self._data = self._raw[:need].reshape((n_new, self._n_samp) + self._sample_shape)

    def reset(self, dets=None):
        if dets is None:
            if self._accel_used:
                self.accel_reset()   # the host side is stale anyway
                return
            self._data[:] = 0
            if self.accel_exists():
                self.accel_reset()
        elif set(dets) >= set(self._detectors):
            self.reset(None)
        else:
            on_dev = self.accel_in_use()
            if on_dev:
                self.accel_update_host()
            for d in dets:
                self._data[self._name2idx[d]] = 0
            if on_dev:
                self.accel_update_device()

    def _row(self, key):
        if isinstance(key, (str, np.str_)):
            return self._name2idx[key]
        return key

    def __getitem__(self, key):
        if self._accel_used:
            self.accel_update_host()
        if isinstance(key, tuple):
            first = key[0]
            if isinstance(first, (list, tuple)) and first and isinstance(first[0], (str, np.str_)):
                first = [self._name2idx[x] for x in first]
            else:
                first = self._row(first)
            return self._data[(first,) + tuple(key[1:])]
        return self._data[self._row(key)]

    def __setitem__(self, key, value):
        if self._accel_used:
            self.accel_update_host()
        if isinstance(key, tuple):
            self._data[(self._row(key[0]),) + tuple(key[1:])] = value
        else:
            self._data[self._row(key)] = value

    # small accessors of the reference class (src/toast/observation_data.py:181-246, :462-500)
    def keys(self):
        return list(self._detectors)

    @property
    def sample_shape(self):
        return tuple(self._sample_shape)

    @property
    def flatdata(self):
        """1-D view of the buffer holding the current detectors."""
        if self._accel_used:
            self.accel_update_host()
        return self._data.reshape(-1)

    def memory_use(self):
        return int(self._raw.nbytes)

    def update_units(self, new_units):
        self.units = new_units

    def view(self, key):
        """Array view for a detector name, index, slice or list of names (no copy for the first three)."""
        return self[key]

    def __delitem__(self, key):
        raise NotImplementedError("Cannot delete individual elements")

    def __iter__(self):
        return iter(self._detectors)

    def __len__(self):
        return len(self._detectors)

    def clear(self):
        """Release the device copy and the host buffer."""
        if self.accel_exists():
            self.accel_delete()
        self._data = self._raw = None

    # accelerator protocol
    def _accel_exists(self):
        return self._data.size > 0 and accel_data_present(self._data, self._accel_name)

    def _accel_create(self, zero_out=False):
        # a two-dimensional float64 array per detector is a timestream: scan_map, noise_weight, the FFT passes and the
        # template projections read and write it in one sweep -- the arena keeps such blocks in rank-interleaved slabs
        streamed = self._data.ndim == 2 and self._data.dtype == np.float64
        accel_data_create(self._data, self._accel_name, zero_out=zero_out, owner=self, kind=1 if streamed else 0)

    def _accel_update_device(self):
        accel_data_update_device(self._data, self._accel_name)

    def _accel_update_host(self):
        accel_data_update_host(self._data, self._accel_name)

    def _accel_delete(self):
        accel_data_delete(self._data, self._accel_name)

    def _accel_reset(self):
        accel_data_reset(self._data, self._accel_name)


class _KeyAccel:
    """Key-wise accelerator calls of the reference's DetDataManager / SharedDataManager
    (src/toast/observation_data.py:881-1039): thin wrappers around the objects' own methods."""

    def _obj(self, key):
        return self[key]

    def accel_exists(self, key):
        return self._obj(key).accel_exists()

    def accel_in_use(self, key):
        return self._obj(key).accel_in_use()

    def accel_used(self, key, state):
        self._obj(key).accel_used(state)

    def accel_create(self, key, zero_out=False):
        self._obj(key).accel_create(key, zero_out=zero_out)

    def accel_update_device(self, key):
        self._obj(key).accel_update_device()

    def accel_update_host(self, key):
        self._obj(key).accel_update_host()

    def accel_delete(self, key):
        self._obj(key).accel_delete()

    def accel_reset(self, key):
        self._obj(key).accel_reset()

    def accel_clear(self):
        for key in list(self.keys()):
            if self._obj(key).accel_exists():
                self._obj(key).accel_delete()

    def memory_use(self):
        """Bytes held on the host.  A size query: it must not go through ``.data`` (the lazy-coherence property would
        copy every device-resident buffer back and make the host the current side)."""
        total = 0
        for k in self.keys():
            obj = self._obj(k)
            if hasattr(obj, "memory_use"):
                total += int(obj.memory_use())
            else:
                raw = getattr(obj, "_raw", None)
                total += int(raw.nbytes) if raw is not None else 0
        return total


class DetDataManager(_KeyAccel, MutableMapping):
    """``ob.detdata`` (reference: observation_data.py:620-1190)."""

    def __init__(self, n_samp, local_detectors):
        self._n_samp = n_samp
        self._local_detectors = list(local_detectors)
        self._store = {}

    def create(self, name, sample_shape=(), dtype=np.float64, detectors=None, units=None):
        if name in self._store:
            raise RuntimeError(f"detdata '{name}' already exists")
        dets = self._local_detectors if detectors is None else list(detectors)
        self._store[name] = DetectorData(dets, (self._n_samp,) + tuple(sample_shape), dtype, units=units)
        self._store[name]._accel_name = name
        return self._store[name]

    def ensure(self, name, sample_shape=(), dtype=np.float64, detectors=None, accel=False, create_units=None,
               zero_new=True):
        """Make sure ``name`` exists with this shape/dtype and holds ``detectors``.

        Returns True when it already held all requested detectors (callers then skip
        recomputation), False when it was created or its detector list changed
        (observation_data.py:725-860).  With ``accel=True`` the buffer also exists on the
        device afterwards and is marked as in use there.  ``zero_new=False``: the caller is about to write every
        sample of every row, a new device buffer need not be cleared first (17.7 GB of Stokes weights: 3 ms)."""
        dets = self._local_detectors if detectors is None else list(detectors)
        existing = True
        if name not in self._store:
            self.create(name, sample_shape, dtype, dets, units=create_units)
            existing = False
        else:
            cur = self._store[name]
            if cur.detector_shape[1:] != tuple(sample_shape) or cur.dtype != np.dtype(dtype):
                raise RuntimeError(f"detdata '{name}' exists with a different sample shape or dtype")
            have = set(cur.detectors)
            if not all(d in have for d in dets):
                cur.change_detectors(dets)
                existing = False
        obj = self._store[name]
        if accel and accel_enabled() and obj.buffer.size > 0:   # (no detectors: nothing to hold on the device)
            if not obj.accel_exists():
                obj.accel_create(name, zero_out=(not existing) and zero_new)
                if existing:
                    obj.accel_update_device()
            elif not obj.accel_in_use():
                obj.accel_update_device()
            obj.accel_used(True)
        return existing

    def rename(self, original, new_name):
        """Move the object stored under ``original`` to ``new_name`` (observation_data.py:861-879)."""
        if new_name in self._store:
            raise RuntimeError(f"detdata key '{new_name}' already exists")
        self._store[new_name] = self._store.pop(original)

    def __getitem__(self, name):
        return self._store[name]

    def __setitem__(self, name, value):
        self._store[name] = value

    def __delitem__(self, name):
        obj = self._store.pop(name)
        if obj.accel_exists():
            obj.accel_delete()

    def __iter__(self):
        return iter(self._store)

    def __len__(self):
        return len(self._store)


class SharedData(AcceleratorObject):
    """A per-observation array shared by all detectors (boresight, flags, HWP angle, times)."""

    def __init__(self, array, name="shared"):
        super().__init__(name)
        self.data = np.ascontiguousarray(array)

    def _accel_exists(self):
        return accel_data_present(self.data, self._accel_name)

    def _accel_create(self, zero_out=False):
        accel_data_create(self.data, self._accel_name, zero_out=zero_out, owner=self)

    def _accel_update_device(self):
        accel_data_update_device(self.data, self._accel_name)

    def _accel_update_host(self):
        accel_data_update_host(self.data, self._accel_name)

    def _accel_delete(self):
        accel_data_delete(self.data, self._accel_name)

    def _accel_reset(self):
        accel_data_reset(self.data, self._accel_name)


class SharedDataManager(_KeyAccel, dict):
    def create(self, name, array):
        self[name] = SharedData(array, name)
        return self[name]


# ----------------------------------------------------------------------------- instrument
class Focalplane:
    """Detector table: ``detector_data[det]`` has ``quat``, ``gamma``, ``pol_leakage``
    (epsilon) and ``cal`` (reference: src/toast/instrument.py Focalplane)."""

    def __init__(self, detectors, quats, gamma=None, epsilon=None, cal=None, sample_rate=1.0, columns=None):
        self.detectors = list(detectors)
        n = len(self.detectors)
        self.sample_rate = float(sample_rate)
        gamma = np.zeros(n) if gamma is None else np.asarray(gamma, dtype=np.float64)
        epsilon = np.zeros(n) if epsilon is None else np.asarray(epsilon, dtype=np.float64)
        cal = np.ones(n) if cal is None else np.asarray(cal, dtype=np.float64)
        self._table = {
            d: dict(quat=np.asarray(quats[i], dtype=np.float64), gamma=float(gamma[i]),
                    pol_leakage=float(epsilon[i]), cal=float(cal[i]))
            for i, d in enumerate(self.detectors)
        }

        # extra per-detector columns (wafer, band, pixel ... : the keys of split map-making)
        for name, values in (columns or {}).items():
            for d, v in zip(self.detectors, values):
                self._table[d][name] = v

    def __getitem__(self, det):
        return self._table[det]

    def detector_groups(self, column):
        """Detectors grouped by the value of a focalplane column (instrument.py
        Focalplane.detector_groups)."""
        groups = {}
        for d in self.detectors:
            if column not in self._table[d]:
                raise RuntimeError(f"focalplane has no column '{column}'")
            groups.setdefault(self._table[d][column], []).append(d)
        return groups


class Telescope:
    def __init__(self, name, focalplane):
        self.name = name
        self.focalplane = focalplane


class Observation(MutableMapping):
    """One observation: telescope, ``n_local_samples``, ``detdata``, ``shared``, ``intervals``
    and a dict of metadata (e.g. the noise model)."""

    def __init__(self, comm, telescope, n_samples, name="obs", detectors=None):
        self.comm = comm
        self.telescope = telescope
        self.name = name
        self.n_local_samples = int(n_samples)
        self.local_detectors = list(telescope.focalplane.detectors if detectors is None else detectors)
        self.local_detector_flags = {d: 0 for d in self.local_detectors}
        times = None
        self.detdata = DetDataManager(self.n_local_samples, self.local_detectors)
        self.shared = SharedDataManager()
        self.intervals = IntervalsManager(self.n_local_samples, times)
        self._meta = {}

    def set_times(self, times):
        self.shared.create(defaults.times, np.asarray(times, dtype=np.float64))
        self.intervals._times = self.shared[defaults.times].data
        self.intervals[None] = IntervalList(self.intervals._times, samplespans=[(0, self.n_local_samples)])

    def update_local_detector_flags(self, flags):
        self.local_detector_flags.update(flags)

    def select_local_detectors(self, selection=None, flagmask=0):
        """Local detectors restricted to ``selection`` and with ``(flags & flagmask) == 0``
        (reference: src/toast/observation.py select_local_detectors)."""
        sel = None if selection is None else set(selection)
        out = []
        for d in self.local_detectors:
            if sel is not None and d not in sel:
                continue
            if self.local_detector_flags.get(d, 0) & flagmask:
                continue
            out.append(d)
        return out

    def __getitem__(self, key):
        return self._meta[key]

    def __setitem__(self, key, value):
        self._meta[key] = value

    def __delitem__(self, key):
        del self._meta[key]

    def __iter__(self):
        return iter(self._meta)

    def __len__(self):
        return len(self._meta)


class Data(MutableMapping):
    """Observations + global objects (pixel distributions, maps, amplitudes)."""

    def __init__(self, comm=None):
        self.comm = Comm() if comm is None else comm
        self.obs = []
        self._internal = {}
        self._pinned = {k: set() for k in ("global", "detdata", "shared")}
        # Lazy host coherence (MI355X: 288 GB of HBM): Pipelines leave detector data resident
        # and device-current at finalize; ``DetectorData.data`` copies back on first host access.
        # ``_protected``: detdata keys staged by a running Pipeline (never evicted).
        # (TOAST_HIP_LAZY_HOST=0: the reference's eager copy-back / delete at the end of every Pipeline)
        self.lazy_host = os.environ.get("TOAST_HIP_LAZY_HOST", "1") != "0"
        self._protected = []   # one set per running Pipeline
        import weakref

        from .accel import add_eviction_handler

        ref = weakref.ref(self)
        add_eviction_handler(lambda: (ref().accel_evict() if ref() is not None else 0))

    def all_detector_groups(self, column=None, selection=None, flagmask=0):
        """Valid detectors of all observations (and all processes), split by the value of a
        focalplane column; ``{"ALL": None}``-style single group without a column
        (src/toast/data.py:141-230)."""
        splits = {}
        for ob in self.obs:
            groups = {"ALL": ob.local_detectors} if column is None else ob.telescope.focalplane.detector_groups(column)
            dets = set(ob.select_local_detectors(selection, flagmask=flagmask))
            for k, v in groups.items():
                for d in v:
                    if d in dets and d not in splits.setdefault(k, []):
                        splits[k].append(d)
        splits = {k: v for k, v in splits.items() if len(v) > 0}
        if self.comm.comm_world is not None:
            import torch.distributed as dist

            gathered = [None] * self.comm.world_size
            dist.all_gather_object(gathered, splits)
            merged = {}
            for part in gathered:
                for k, v in part.items():
                    for d in v:
                        if d not in merged.setdefault(k, []):
                            merged[k].append(d)
            splits = {k: merged[k] for k in sorted(merged, key=str)}
        return splits

    def accel_pin(self, names):
        """Keep these objects resident on the device across Pipelines: ``accel_update_host`` and
        ``accel_delete`` driven by Pipeline staging skip them while they are in use there.  For
        buffers that are written once and then only read (cached pointing), on a 288 GB device."""
        for k, v in names.items():
            self._pinned[k] |= set(v)

    def accel_unpin(self, names=None, update_host=True):
        """Release pinned objects (all when ``names`` is None): bring the host copies up to date
        and free the device copies (the state a Pipeline would have left)."""
        names = {k: set(v) for k, v in self._pinned.items()} if names is None else names
        for k, v in names.items():
            self._pinned[k] -= set(v)
        for key, obj in self._each({k: list(v) for k, v in names.items()}, include_pinned=True):
            if obj.accel_exists():
                if self.lazy_host and isinstance(obj, DetectorData):
                    continue  # stays resident; copied back on host access or eviction
                if update_host and obj.accel_in_use():
                    obj.accel_update_host()
                obj.accel_delete()

    def accel_evict(self):
        """Write back and free every resident detector-data buffer and every lazily retained map that is neither
        pinned nor staged by a running Pipeline.  Returns the number of bytes released."""
        freed = 0
        for ob in self.obs:
            for key in list(ob.detdata.keys()):
                obj = ob.detdata[key]
                if (key in self._pinned["detdata"] or any(key in s for s in self._protected)
                        or getattr(obj, "_accel_hold", 0) > 0 or not obj.accel_exists()):
                    continue
                if obj.accel_in_use():
                    obj.accel_update_host()
                freed += obj.buffer.nbytes
                obj.accel_delete()
        # lazily retained maps (PixelData): the same treatment
        from .pixels import PixelData

        for key, obj in list(self._internal.items()):
            if (not isinstance(obj, PixelData) or key in self._pinned["global"]
                    or any(("global:" + key) in s for s in self._protected) or getattr(obj, "_accel_hold", 0) > 0
                    or not obj.accel_exists()):
                continue
            if obj.accel_in_use():
                obj.accel_update_host()
            freed += obj.buffer.nbytes
            obj.accel_delete()
        return freed

    def all_local_detectors(self, selection=None, flagmask=0):
        seen, out = set(), []
        for ob in self.obs:
            for d in ob.select_local_detectors(selection, flagmask):
                if d not in seen:
                    seen.add(d)
                    out.append(d)
        return out

    # dict interface for global objects
    def __getitem__(self, key):
        return self._internal[key]

    def __setitem__(self, key, value):
        self._internal[key] = value

    def __delitem__(self, key):
        obj = self._internal.pop(key)
        if isinstance(obj, AcceleratorObject) and obj.accel_exists():
            obj.accel_delete()

    def __iter__(self):
        return iter(self._internal)

    def __len__(self):
        return len(self._internal)

    # -- accelerator staging by ``requires()`` / ``provides()`` dictionaries
    #    (reference: src/toast/data.py accel_create / accel_update_device / ...)
    def _each(self, names, include_pinned=True):
        for key in names.get("global", []):
            obj = self._internal.get(key)
            if isinstance(obj, AcceleratorObject) and (include_pinned or key not in self._pinned["global"]):
                yield key, obj
        for ob in self.obs:
            for key in names.get("detdata", []):
                if key in ob.detdata and (include_pinned or key not in self._pinned["detdata"]):
                    yield key, ob.detdata[key]
            for key in names.get("shared", []):
                if key in ob.shared and (include_pinned or key not in self._pinned["shared"]):
                    yield key, ob.shared[key]

    def accel_create(self, names):
        for key, obj in self._each(names):
            if not obj.accel_exists():
                obj.accel_create(key)

    def accel_update_device(self, names):
        for key, obj in self._each(names):
            if obj.accel_exists() and not obj.accel_in_use():
                obj.accel_update_device()

    def accel_update_host(self, names):
        for key, obj in self._each(names, include_pinned=False):
            if obj.accel_exists() and obj.accel_in_use():
                obj.accel_update_host()

    def accel_delete(self, names):
        for key, obj in self._each(names, include_pinned=False):
            if obj.accel_exists():
                obj.accel_delete()
