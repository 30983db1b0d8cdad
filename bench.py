#!/usr/bin/env python3
"""Headline benchmark: detector-samples/s through one PCG (A^T N^-1 A) iteration.

One "step" = one pass of the PCG's pointing-matrix projections over every local
detector-sample, exactly the sequence of SolverLHS (reference:
src/toast/ops/mapmaker_solve.py:342-506):

    zmap = 0
    zmap += A^T N^-1 tod         build_noise_weighted      (41 B / det-sample)
    [N > 1]  all-reduce(zmap)    RCCL sum over the detector shards, enqueued on the kernels' stream by the library's
                                 own communicator (toast_hip_comm_*) once its results have been checked against
                                 torch.distributed on the job itself; else torch.distributed's all-reduce
    zmap  = C zmap               cov_apply_diag            (map sized)
    tod2 -= A zmap ; tod2 *= w   scan_map(subtract) + fused noise_weight  (48 B / det-sample)

Workload (config.workload = "cfg3"): BASELINE.json configs[2] -- 1024 detectors x 1 h @ 200 Hz
(720 000 samples) per GPU, Nside 1024 NEST, IQU, synthetic satellite scan (spin 10 min,
precession 50 min, 30 deg / 65 deg opening angles), white-noise TOD, 0.5 % random detector
flags, a 1 % shared-flag block.  Pointing (quaternions -> pixels, Stokes weights) is expanded
on the GPU by the library's own kernels before the timed region; all inputs are resident in
HBM when timing starts.  With N GPUs every rank holds its own 1024 detectors of a 1024 N
detector focalplane observing the same scan (weak scaling, detector-sharded like configs[3]).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2]
        (N > 1: one rank per GPU -- either launched by torch.distributed.run, or, when called
        as plain `python bench.py --gpus N`, bench.py starts that launcher itself as a child process)
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_det per GPU, n_samp, rate Hz, nside)
    "cfg3": (1024, 720000, 200.0, 1024),
    "cfg2": (64, 360000, 100.0, 512),
    "mini": (16, 50000, 100.0, 256),
    # configs[3]: 4096 detectors x 4 h @200 Hz over 8 GPUs = 512 detectors x 2 880 000 samples each
    "cfg4": (512, 2880000, 200.0, 1024),
    # configs[4] structure (ground CES: ~100 sweep intervals, flagged turnarounds, Nside 2048;
    # 2048 detectors over 8 GPUs = 256 each); the atmosphere / ground-template inputs are upstream
    "cfg5g": (256, 720000, 200.0, 2048),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_BNW = 41.0       # pixel 8 + weights 24 + tod 8 + det flag 1   
BYTES_SCAN = 48.0      # pixel 8 + weights 24 + tod read 8 + tod write 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dets", type=int, default=32, help="detectors in the CPU baseline sample")
    ap.add_argument("--no-operator-level", action="store_true",
                    help="skip the operator-level entry (NoiseFilter + MapMaker of workflows/mapmaker_pcg.py at cfg3; "
                         "about 10 s, most of it the host-side simulation of its inputs)")
    ap.add_argument("--no-lhs", action="store_true",
                    help="skip the offset-template left-hand side (operator sequence / fused / packed) on the bench buffers")
    ap.add_argument("--pcg-extra", action="store_true",
                    help="also time the full PCG LHS with offset templates (operator sequence vs fused kernels)")
    ap.add_argument("--torch-alloc", action="store_true",
                    help="EXPERIMENT: allocate the TOD-domain buffers with torch instead of the library's memory manager")
    ap.add_argument("--no-fft", action="store_true", help="skip the FFT noise-weighting measurement")
    ap.add_argument("--no-fft-long", action="store_true", help="skip the FFT measurement at the configs[3] shard's length")
    ap.add_argument("--no-cfg4", action="store_true",
                    help="with --gpus 8 and the default workload: do not also time the configs[3] shard (cfg4)")
    ap.add_argument("--shard-workload", default=None, choices=sorted(WORKLOADS),
                    help="also time this workload after the headline one and attach it as `configs3_shard` "
                         "(default: cfg4 when --gpus 8 runs the default workload)")
    ap.add_argument("--hwp", action="store_true", help="rotating half-wave plate (88 rpm): Stokes weights with HWP angle")
    ap.add_argument("--unfused", action="store_true", help="run noise_weight as its own kernel (105 B variant)")
    return ap.parse_args()


def _map_residual_vs_reference():
    """ops.MapMaker on the small end-to-end case (4 detectors x 60 000 samples, Nside 64, 12 PCG iterations; inputs from seeds:
    tests/mapmaker_case.py) against the committed fixture tests/golden/mapmaker_e2e.npz -- amplitudes and destriped map
    produced in the build container by the reference's own compiled kernels driven through the reference's own solve()
    (tests/golden/make_golden_mapmaker.py).  max |ours - reference| / max |reference| in the default (atomic) mode: the
    north star's "map residual < 1e-10 vs reference" on the final product, measured in this very process."""
    try:
        tests = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests")
        if tests not in sys.path:
            sys.path.insert(0, tests)
        import test_gpu_mapmaker_e2e as e2e

        got, want = e2e._run("small"), e2e._fixture("small")
        sel = want["pix_index"]
        rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))      # noqa: E731
        return {
            "map_residual_vs_reference": rel(got["map"][sel], want["map"]),
            "amplitude_residual_vs_reference": rel(got["amplitudes"], want["amplitudes"]),
            "residual_history_vs_reference": float(np.max(np.abs(got["history"] - want["history"]) / want["history"])),
            "residual_case": "tests/golden/mapmaker_e2e.npz 'small' (reference kernels + reference solve()), route %s" % "/".join(got["route"]),
        }
    except Exception as err:   # noqa: BLE001 -- an extra of the line
        return {"map_residual_vs_reference": None, "residual_error": repr(err)[:200]}


def operator_level():
    """The same configuration through the operators a user calls (NoiseFilter, then MapMaker with offset templates and
    10 PCG iterations: workflows/mapmaker_pcg.py with its defaults = cfg3), timed by the workflow itself.  Not the
    headline: it shows what the Operator surface delivers with the manager's own allocations and host-resident inputs
    (uploads included)."""
    import contextlib
    import importlib.util
    import io

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "workflows", "mapmaker_pcg.py")
    spec = importlib.util.spec_from_file_location("toast_amd_workflow_mapmaker_pcg", path)
    wf = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(wf)
        with contextlib.redirect_stdout(io.StringIO()):
            wf.main([])
        st = dict(wf.LAST_STATS)
    except Exception as err:   # the headline line must not depend on this extra
        st = {"error": repr(err)}
    finally:
        # drop the workflow's data and hand every device buffer of the manager (cache of released blocks included)
        # back to the driver
        import gc

        from toast_amd.accel import accel_assign_device

        wf = None
        gc.collect()
        try:
            accel_assign_device(1, 0, 0.0, False)     # (registered arrays go back to the arena; its slabs stay)
        except Exception:
            pass
    if "error" in st:
        return st
    laps = st.pop("laps", {})
    return {
        **_map_residual_vs_reference(),
        "workload": "cfg3 through ops.NoiseFilter + ops.MapMaker (workflows/mapmaker_pcg.py: host-resident inputs, "
                    "uploads included, 3.7 M offset amplitudes, full_pointing=True)",
        "noise_filter_s": laps.get("NoiseFilter"),
        "mapmaker_s": st.get("mapmaker_s"),
        "pcg_iterations": st.get("iterations"),
        "pcg_iteration_ms": st.get("pcg_iteration_ms"),
        "pcg_Gsamp_s": st.get("pcg_Gsamp_s"),
        "relative_residual": st.get("relative_residual"),
        "phases_s": st.get("phases_s"),
        #  a phase boundary is a host-side mark: the device is not waited for there, so a phase's device work
        # is booked under whichever later phase first waits (TOAST_HIP_PHASE_SYNC=1 synchronises at the boundaries);
        # mapmaker_s and noise_filter_s are synchronised wall times
        "phases_s_are": "host enqueue time per phase (not synchronised)",
        "lhs_route": st.get("lhs_route"),
        "lhs_pack_bytes": st.get("lhs_pack_bytes"),
    }


# Multi-rank runs only.  The sections after the timed steps try things that have never run on more than one GPU (the A/B
# repetitions of the communicator's modes); a collective that hangs cannot be caught as an exception and a memory fault
# ends the process -- the headline must not be lost to an extra.  So rank 0 keeps a small child process (started before
# anything touches the GPU) that owns the real stdout: it is sent the line as it stands once the timed steps are over,
# later the complete line, and prints the LAST one it received when rank 0's end of the pipe closes -- however rank 0
# ended.  A timer on every rank turns a hang into an exit.
_GUARDIAN_SRC = (
    "import sys\n"
    "last = None\n"
    "for line in sys.stdin:\n"
    "    if line.strip():\n"
    "        last = line.rstrip('\\n')\n"
    "if last is not None:\n"
    "    sys.stdout.write(last + '\\n')\n"
    "    sys.stdout.flush()\n"
)
_LIFELINE = {"guardian": None, "timer": None}


def _guardian_start(stdout_fd):
    import subprocess

    _LIFELINE["guardian"] = subprocess.Popen([sys.executable, "-c", _GUARDIAN_SRC], stdin=subprocess.PIPE, stdout=stdout_fd,
                                             text=True, start_new_session=True)


def _guardian_send(line):
    g = _LIFELINE["guardian"]
    g.stdin.write(line + "\n")
    g.stdin.flush()


def _guardian_finish():
    g = _LIFELINE["guardian"]
    _LIFELINE["guardian"] = None
    g.stdin.close()
    g.wait()


def _lifeline_arm(out, rank):
    import threading

    if _LIFELINE["timer"] is not None:
        return
    limit = float(os.environ.get("TOAST_BENCH_EXTRAS_TIMEOUT_S", "900"))
    if rank == 0 and _LIFELINE["guardian"] is not None:
        short = dict(out)
        short["truncated"] = ("the sections after the timed steps did not finish (time limit %.0f s, or the process ended "
                              "in one of them)" % limit)
        _guardian_send(json.dumps(short))

    def expire():
        if rank != 0:
            time.sleep(2.0)
        # non-zero: a hung collective or a wedged process is not a success; the guardian still prints the headline it was
        # sent, with its `truncated` key, and the driver decides from both
        os._exit(3)

    t = threading.Timer(limit, expire)
    t.daemon = True
    t.start()
    _LIFELINE["timer"] = t


def _lifeline_disarm():
    if _LIFELINE["timer"] is not None:
        _LIFELINE["timer"].cancel()
        _LIFELINE["timer"] = None


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) as a FRESH child
    `python -m torch.distributed.run ... bench.py <same arguments>`, relay its output and return its exit code.
    This process has not touched the GPU (torch is not even imported yet) and it never replaces itself: the
    ranks are children, the parent only waits."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        # rank 0's JSON line goes to stdout; anything else the ranks print there (RCCL's version banner, launcher
        # notices) is passed on to stderr, so that stdout carries exactly the one line the contract asks for
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    # stdout carries exactly ONE line, the JSON: while the benchmark runs, file descriptor 1 points at stderr, so that
    # whatever a library prints there (RCCL's version banner at communicator set-up) cannot end up next to it
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0:
        _guardian_start(real_stdout)         # (before torch is imported: nothing has touched the GPU yet)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    # TOAST_BENCH_SHARE_GPU=1 (tests only): several ranks on the GPUs that exist, collectives over
    # gloo -- lets the N > 1 code path run on a single-GPU box.  Numbers from such a run mean nothing.
    share = os.environ.get("TOAST_BENCH_SHARE_GPU", "0") == "1"
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # TOAST_BENCH_SINGLE_RANK_COMM=1 (tests only): run the N > 1 code path -- process group, the library's RCCL
    # communicator with its self-check, the per-step collective -- with ONE rank, which a single-GPU box can do.
    single = world == 1 and os.environ.get("TOAST_BENCH_SINGLE_RANK_COMM", "0") == "1"
    if single:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29561")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    op_level = None
    if (rank == 0 and world == 1 and args.workload == "cfg3" and not args.no_operator_level
            and not args.no_cpu_baseline):
        # (like the CPU baseline an extra of the default invocation only: profiling runs pass --no-cpu-baseline.)
        # First, while the process holds no device memory: its allocations are then as fresh as a user's, and
        # everything it held is released again before the benchmark buffers are allocated.
        op_level = operator_level()
    out = run(args, args.workload, world, rank, dev, headline=True)
    shard = args.shard_workload
    if shard is None and world == 8 and args.workload == "cfg3" and not args.no_cfg4:
        shard = "cfg4"
    if shard is not None:
        # BASELINE configs[3] (4096 detectors x 4 h @ 200 Hz over 8 GPUs): the default line above is the
        # weak-scaling cfg3 shard (same per-GPU work at every N, which is what a scaling curve needs);
        # the actual configs[3] shard (512 detectors x 2 880 000 samples per GPU) is timed here as well.
        torch.cuda.empty_cache()
        sub = run(args, shard, world, rank, dev, headline=False)
        if rank == 0:
            out["configs3_shard"] = {k: sub[k] for k in ("value", "unit", "ms_per_step", "kernel_ms", "config",
                                                         "allreduce", "roofline")}
            if "pcg_lhs_offset_templates" in sub:
                out["configs3_shard"]["pcg_lhs_offset_templates"] = {
                    k: v for k, v in sub["pcg_lhs_offset_templates"].items() if k.startswith(("packed", "predicted"))}
    if op_level is not None:
        out["operator_level"] = op_level
    sys.stdout.flush()
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)      # the banner sits in the C library's buffer until somebody flushes it
    except OSError:
        pass
    _lifeline_disarm()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0 and _LIFELINE["guardian"] is not None:
        _guardian_send(json.dumps(out))
        _guardian_finish()
    elif rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or single:
        dist.destroy_process_group()


def run(args, workload, world, rank, dev, headline=True):
    """Time one workload; returns the JSON object (complete on rank 0)."""
    import torch
    import torch.distributed as dist

    from toast_amd import capi, synth

    D = capi.dev
    stream = torch.cuda.current_stream().cuda_stream
    share = os.environ.get("TOAST_BENCH_SHARE_GPU", "0") == "1"
    # the collectives of the N > 1 path run (also with one rank when the single-rank test mode built a process group)
    multi = world > 1 or (dist.is_available() and dist.is_initialized())

    n_det, n_samp, rate, nside = WORKLOADS[workload]
    nnz = 3
    nps = 3072 if nside >= 16 else 12 * nside * nside
    n_submap = 12 * nside * nside // nps

    # ------------------------------------------------------------------ synthetic inputs
    t_setup = time.time()
    if not args.torch_alloc:
        # The pool: pixels 8 + weights 24 + two timestreams 16 + flags 1 + quaternions 32 B per det-sample (the packed
        # cache later takes the quaternions' place), taken from the driver in one piece -- the reference's mem_gb
        # (accelerator.cpp:292-303).  A process that already holds that much (the operator-level run before this one)
        # takes nothing.
        capi.arena_reserve(int(74.0 * n_det * n_samp) + (1 << 30))
        # ... and the one timestream that the step reads AND writes (scan_map's tod2) in a slab whose 1 GB chunks
        # alternate between two HBM zones (csrc/vmm_slab.cpp): what ops' timestreams get (DetectorData, 2-D float64)
        capi.arena_reserve(int(8.0 * n_det * n_samp) + (2 << 30), streamed=True)
    fp_all, gamma_all = synth.hex_focalplane(n_det * world, fov_deg=10.0)
    fp = np.ascontiguousarray(fp_all[rank * n_det : (rank + 1) * n_det])
    gamma = np.ascontiguousarray(gamma_all[rank * n_det : (rank + 1) * n_det])
    if workload == "cfg5g":
        bore, ivl, sflags_h = synth.ground_scan(n_samp, rate)
    else:
        bore = synth.satellite_boresight(n_samp, rate, 600.0, 30.0, 3000.0, 65.0)
        ivl = synth.make_intervals(n_samp, 1, rate)
        sflags_h = synth.shared_flags_block(n_samp, 0.01, value=1)
    idx = np.arange(n_det, dtype=np.int32)

    # Persistent buffers first, the big temporary (quaternions, 23.6 GB) last: HBM placement
    # matters (buffers carved out of just-freed memory ran scan_map up to 15 % slower in
    # tools/exp_scan_placement.py), so nothing long-lived is allocated after a large free.
    gen = torch.Generator(device=dev)
    gen.manual_seed(20261001 + rank)
    sigma = 50.0e-6 * np.sqrt(rate)
    d_bore = torch.from_numpy(bore).to(dev)
    d_sflags = torch.from_numpy(sflags_h).to(dev)
    d_hsub = torch.zeros(n_submap, dtype=torch.uint8, device=dev)

    nds = n_det * n_samp
    sizes = {"pixels": nds * 8, "weights": nds * 24, "tod": nds * 8, "tod2": nds * 8, "dflags": nds}
    placement = None
    # Default: one allocation per buffer, which is what the product's memory manager does
    # (toast_hip::Manager::create -> hipMalloc per registered array); operators get exactly this.
    allocator = ("toast_hip::Manager arena (every TOD-domain buffer is a block from toast_hip_device_malloc, i.e. a range of "
                 "the slabs the operators' arrays live in; csrc/arena.cpp)")
    if args.torch_alloc:
        allocator = "experiment: torch caching allocator (one hipMalloc per buffer, no placement policy)"
    managed = []     # device blocks from the library's memory manager (released at the end of run())

    def manager_tensor(nbytes, dtype, shape, streamed=False, scatter=False):
        """A torch view of a block from toast_hip::Manager::device_alloc -- the arena every operator's buffers come
        from (toast_hip_device_malloc(flags = -1))."""
        ptr = capi.device_malloc(nbytes, -3 if streamed else (-4 if scatter else -1))
        managed.append(ptr)

        class _Block:
            pass

        blk = _Block()
        typestr = {torch.int64: "<i8", torch.float64: "<f8", torch.uint8: "|u1", torch.int32: "<i4", torch.float32: "<f4"}[dtype]
        blk.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(ptr, False), version=3)
        return torch.as_tensor(blk, device=dev)

    def manager_release(t):
        """Give the block behind a manager_tensor back to the arena (the caller drops its view)."""
        ptr = t.data_ptr()
        managed.remove(ptr)
        torch.cuda.synchronize()
        capi.device_free(ptr)

    def carve(name, dtype, shape):
        if args.torch_alloc:
            return torch.empty(shape, dtype=dtype, device=dev)
        return manager_tensor(sizes[name], dtype, shape, streamed=(name == "tod2"))

    def temp(dtype, shape):
        """A large temporary (quaternions, the packed cache): from the arena as well, released with drop()."""
        if args.torch_alloc:
            return torch.empty(shape, dtype=dtype, device=dev)
        return manager_tensor(int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size(), dtype, shape)

    def drop(t):
        if not args.torch_alloc:
            manager_release(t)

    if os.environ.get("TOAST_BENCH_POOL_FILLER_GB", "") != "" and not args.torch_alloc:
        # EXPERIMENT (profiles/r05_d): a read-mostly block in front of the arrays moves them along their slab, i.e.
        # relative to the zone boundary inside it
        _filler = manager_tensor(int(float(os.environ["TOAST_BENCH_POOL_FILLER_GB"]) * 2 ** 30), torch.uint8, (1,))
    d_pixels = carve("pixels", torch.int64, (n_det, n_samp))
    d_weights = carve("weights", torch.float64, (n_det, n_samp, 3))
    d_tod = carve("tod", torch.float64, (n_det, n_samp))
    d_tod2 = carve("tod2", torch.float64, (n_det, n_samp))
    d_dflags = carve("dflags", torch.uint8, (n_det, n_samp))
    d_tod.normal_(0.0, sigma, generator=gen)
    d_tod2.normal_(0.0, sigma, generator=gen)
    for d0 in range(0, n_det, 128):  # bounded temporaries
        blk = d_dflags[d0:d0 + 128]
        blk.copy_((torch.rand(blk.shape, device=dev, generator=gen) < 0.005).to(torch.uint8))
    d_quats = temp(torch.float64, (n_det, n_samp, 4))

    def timed(fn, reps=1, prime=False):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        if prime:
            # one un-timed call in front, NOT waited for: the host-side preparation of the first timed call then runs
            # while the device still works on it, as it does for every later call of a queue (device throughput)
            fn()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    n_in_view = int(np.sum(ivl["last"] - ivl["first"]))   # samples inside the intervals
    nsamp_tot = float(n_det) * n_in_view
    pd_call = lambda: D.pointing_detector(fp, d_bore.data_ptr(), idx, d_quats.data_ptr(), n_samp, ivl,
                                          d_sflags.data_ptr(), n_samp, 1, stream)
    pd_call()                     # (the first pass over a fresh block also pays its first touch: 5.7 instead of ~4 ms)
    t_pd = timed(pd_call, 2)
    pix_call = lambda: D.pixels_healpix(idx, d_quats.data_ptr(), d_sflags.data_ptr(), n_samp, 1, idx,
                                        d_pixels.data_ptr(), n_samp, ivl, d_hsub.data_ptr(), n_submap, nps,
                                        nside, True, stream)
    pix_call()
    t_pix = timed(pix_call, 3)
    d_hwp = None
    if args.hwp:
        hwp_h = np.ascontiguousarray(2 * np.pi * ((np.arange(n_samp) * (88.0 / 60.0) / rate) % 1.0))
        d_hwp = torch.from_numpy(hwp_h).to(dev)
    hwp_ptr, hwp_n = (d_hwp.data_ptr(), n_samp) if args.hwp else (0, 0)
    hwp_tab_ptr = 0
    if args.hwp:
        d_hwp_tab = torch.empty((n_samp, 2), dtype=torch.float64, device=dev)
        D.hwp_table(hwp_ptr, n_samp, d_hwp_tab.data_ptr(), stream)
        hwp_tab_ptr = d_hwp_tab.data_ptr()
    sw_call = lambda: D.stokes_weights_IQU(idx, d_quats.data_ptr(), idx, d_weights.data_ptr(), n_samp, hwp_ptr, hwp_n, ivl,
                                           np.zeros(n_det), gamma, np.ones(n_det), False, stream)
    sw_call()
    t_sw = timed(sw_call, 2)
    drop(d_quats)
    del d_quats
    # the same two outputs straight from the boresight (no quaternion buffer)
    pt_x = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sflags.data_ptr(),
                             n_shared_flags=n_samp, shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma,
                             cal=np.ones(n_det), d_hwp=hwp_ptr, n_hwp=hwp_n, d_hwp_table=hwp_tab_ptr)
    pix_x = lambda: D.otf_pixels_healpix(pt_x, idx, d_pixels.data_ptr(), n_samp, ivl, d_hsub.data_ptr(), n_submap, nps, stream)
    sw_x = lambda: D.otf_stokes_weights(pt_x, idx, d_weights.data_ptr(), n_samp, ivl, stream)
    pix_x()        # (the first launch of a kernel also loads its code object: 5.7 instead of 3.5 ms in a process that has
    sw_x()         #  not run the operator-level workflow before)
    t_pix_x = timed(pix_x, 2)
    t_sw_x = timed(sw_x, 2)

    # union of hit submaps over ranks -> one global2local for everybody
    hs = d_hsub.to(torch.int32)
    if world > 1:
        dist.all_reduce(hs, op=dist.ReduceOp.MAX)
    g2l_h, hit = synth.global_to_local(hs.cpu().numpy())
    n_local = int(hit.size)
    d_g2l = torch.from_numpy(g2l_h).to(dev)
    # the map: the target of build_noise_weighted's atomics -- a scatter block of the arena, like every ops.PixelData
    if args.torch_alloc:
        d_zmap = torch.zeros((n_local, nps, nnz), dtype=torch.float64, device=dev)
    else:
        d_zmap = manager_tensor(n_local * nps * nnz * 8, torch.float64, (n_local, nps, nnz), scatter=True)
        d_zmap.zero_()
    # packed upper-triangle "covariance": diagonally dominant, O(1)
    d_cov = torch.rand((n_local, nps, 6), dtype=torch.float64, device=dev, generator=gen) * 0.1
    d_cov[..., 0] += 1.0
    d_cov[..., 3] += 1.0
    d_cov[..., 5] += 1.0
    det_scale = np.full(n_det, 1.0 / (sigma * sigma))
    det_w = np.linspace(0.5, 0.9, n_det)
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    # ------------------------------------------------------------------ the sum over the detector shards
    # N > 1: the library's own RCCL communicator (toast_hip_comm_*: the collective is enqueued on the kernels' stream,
    # which is what the operators do) after a check of its results against torch.distributed on this very job; any
    # failure on any rank makes every rank use torch.distributed's all-reduce instead, and the JSON line says which.
    comm_impl, comm_note = None, None
    if multi:
        comm_impl, ok = "torch.distributed", 0
        # (share mode = several test ranks on one GPU over gloo: RCCL would refuse them, so the communicator is tried
        # there only when TOAST_HIP_RCCL_LIB points the library at the tests' shared-memory stand-in for librccl)
        cdev = "cpu" if share else dev       # where the protocol's own small tensors live (gloo: host)
        if (not share or os.environ.get("TOAST_HIP_RCCL_LIB")) and os.environ.get("TOAST_BENCH_COMM", "") != "torch":
            # Every step below that is collective is entered by ALL ranks or by none: first agree that every rank can
            # load RCCL, then rank 0's id (with a validity byte) goes round, then the collective initialisation.
            n_r, r_r, _ = D.comm_info()
            ready = n_r == world and r_r == rank
            if n_r == 0:
                able = torch.tensor([1 if D.comm_available() else 0], dtype=torch.int32, device=cdev)
                dist.all_reduce(able, op=dist.ReduceOp.MIN)
                uid = torch.zeros(129, dtype=torch.uint8)
                if int(able.item()) == 1 and rank == 0:
                    try:
                        uid[:128] = torch.frombuffer(bytearray(D.comm_unique_id()), dtype=torch.uint8)
                        uid[128] = 1
                    except RuntimeError as err:
                        comm_note = repr(err)[:300]
                uid = uid.to(cdev)
                dist.broadcast(uid, src=0)
                uid = uid.cpu()
                if int(uid[128]) == 1:
                    try:
                        D.comm_init(bytes(uid[:128].numpy().tobytes()), world, rank)   # collective
                        ready = True
                    except RuntimeError as err:
                        comm_note = repr(err)[:300]
                elif comm_note is None:
                    comm_note = "librccl could not be loaded on every rank"
            # ... and the communicator is used only if EVERY rank has one
            every = torch.tensor([1 if ready else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(every, op=dist.ReduceOp.MIN)
            ready = int(every.item()) == 1
            try:
                if not ready:
                    raise RuntimeError(comm_note or "no communicator on some rank")
                chk = torch.arange(3 * 4096, dtype=torch.float64, device=dev) * (rank + 1.0)
                ref_chk = chk.to(cdev)
                dist.all_reduce(ref_chk)
                ref_chk = ref_chk.to(dev)
                torch.cuda.synchronize()
                a = chk.clone()
                D.comm_allreduce(a.data_ptr(), a.numel(), np.float64, "sum", stream)
                b = chk.clone()
                D.comm_map_reduce_apply(4096, 3, 0, b.data_ptr(), True, stream)
                torch.cuda.synchronize()
                ok = int(torch.equal(a, ref_chk) and torch.allclose(b, ref_chk, rtol=1e-14, atol=0.0))
                if not ok:
                    comm_note = "toast_hip_comm results differ from torch.distributed"
            except Exception as err:    # noqa: BLE001 -- the benchmark must still run
                comm_note = repr(err)[:300]
        flag = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            comm_impl = "toast_hip_comm (RCCL on the kernels' stream)"
    zmap_count = n_local * nps * nnz

    skip_reduce = [False]     # (the same launches without the collective: what the predicted scaling divides by)

    def allreduce_zmap():
        if comm_impl is None or skip_reduce[0]:
            return
        if comm_impl.startswith("toast_hip_comm"):
            D.comm_allreduce(d_zmap.data_ptr(), zmap_count, np.float64, "sum", stream)
        else:
            dist.all_reduce(d_zmap)

    # ------------------------------------------------------------------ one step
    ev = {k: [] for k in ("bnw", "allreduce", "cov", "scan")}

    def step(record):
        d_zmap.zero_()
        if record:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            e[0].record()
        D.build_noise_weighted(d_g2l.data_ptr(), d_zmap.data_ptr(), nps, nnz, idx, d_pixels.data_ptr(), idx,
                               d_weights.data_ptr(), idx, d_tod.data_ptr(), idx, d_dflags.data_ptr(), n_samp,
                               det_scale, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, stream)
        if record:
            e[1].record()
        allreduce_zmap()
        if record:
            e[2].record()
        D.cov_apply_diag(n_local, nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), stream)
        if record:
            e[3].record()
            e[4].record()
        if args.unfused:
            D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, d_tod2.data_ptr(), idx,
                       d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False, True, False,
                       None, stream)
            D.noise_weight(d_tod2.data_ptr(), n_samp, idx, ivl, det_w, stream)
        else:
            D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, d_tod2.data_ptr(), idx,
                       d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False, True, False,
                       det_w, stream)
        if record:
            e[5].record()
            ev["bnw"].append((e[0], e[1]))
            ev["allreduce"].append((e[1], e[2]))
            ev["cov"].append((e[2], e[3]))
            ev["scan"].append((e[4], e[5]))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
    value = world * nsamp_tot * args.steps / elapsed
    zone_diag = None
    if os.environ.get("TOAST_BENCH_ZONE_DIAG", "") in ("1", "2", "3") and not args.torch_alloc:
        # EXPERIMENT (profiles/r06_e): which of the streams that scan_map / build_noise_weighted read share an HBM zone
        # with the two chunk classes of the written timestream?  One read + write pass (x * 1.0: the data stays bit for
        # bit) over a 1 GB range of a read stream and a 1 GB chunk of tod2, rows dealt alternately: TB/s (slow = same zone).
        gb = 1 << 30
        cls = {}
        for k in range(int(d_tod2.numel() * 8 // gb)):
            _, own, other = capi.arena_block_zone(d_tod2.data_ptr() + k * gb, 1)
            name = "P" if own else ("Q" if other else None)
            if name is not None:
                cls.setdefault(name, d_tod2.data_ptr() + k * gb)
        zone_diag = {"tod2_first_chunk_of_class_offsets_GB": {k: (v - d_tod2.data_ptr()) / gb for k, v in cls.items()}}
        wb = d_weights.numel() * 8
        ranges = {"weights@%d%%" % pct: d_weights.data_ptr() + int((wb - gb) * pct / 100) // 4096 * 4096 for pct in (0, 25, 50, 75, 100)}
        ranges["tod@0%"] = d_tod.data_ptr()
        ranges["tod@100%"] = d_tod.data_ptr() + (d_tod.numel() * 8 - gb) // 4096 * 4096
        for rname, rptr in ranges.items():
            for cname, cptr in cls.items():
                # (the chunk may start inside tod2: 1 GB from there stays inside the block when it is not its last GB)
                n_each = min(gb, d_tod2.data_ptr() + d_tod2.numel() * 8 - cptr) // (1 << 20) * (1 << 20)
                t = capi.probe_stream_split([rptr, cptr], n_each)
                zone_diag["%s vs tod2:%s" % (rname, cname)] = round(4.0 * n_each / (t * 1e-3) / 1e12, 3)
        if len(cls) == 2:
            n_each = min(gb, d_tod2.data_ptr() + d_tod2.numel() * 8 - max(cls.values())) // (1 << 20) * (1 << 20)
            t = capi.probe_stream_split([cls["P"], cls["Q"]], n_each)
            zone_diag["tod2:P vs tod2:Q"] = round(4.0 * n_each / (t * 1e-3) / 1e12, 3)
        zb = d_zmap.numel() * 8 // (1 << 20) * (1 << 20)
        for rname, rptr in list(ranges.items()) + list(("tod2:" + k, v) for k, v in cls.items()):
            t = capi.probe_stream_split([rptr, d_zmap.data_ptr()], zb)
            zone_diag["%s vs zmap" % rname] = round(4.0 * zb / (t * 1e-3) / 1e12, 3)
        ranges["pixels@0%"] = d_pixels.data_ptr()
        for rname in ("pixels@0%",):
            for cname, cptr in list(cls.items()) + [("zmap", d_zmap.data_ptr())]:
                nb = zb if cname == "zmap" else gb
                t = capi.probe_stream_split([ranges[rname], cptr], nb)
                zone_diag["%s vs %s" % (rname, cname if cname == "zmap" else "tod2:" + cname)] = round(4.0 * nb / (t * 1e-3) / 1e12, 3)
        zone_diag["slabs_third_zone"] = capi.alloc_stats().get("slabs_third_zone")
        if os.environ["TOAST_BENCH_ZONE_DIAG"] == "3":
            # every GB of the read arrays against the map's chunk and the next chunk of its run: does a PART of the read-mostly
            # slab share a zone with one of them?
            lo = min(t.data_ptr() for t in (d_pixels, d_weights, d_tod))
            hi = max(t.data_ptr() + t.numel() * t.element_size() for t in (d_pixels, d_weights, d_tod))
            # (the slab's chunks are 1 GB from ITS base, which is only 2 MB aligned: walk down to it)
            sbase = d_zmap.data_ptr() // (2 << 20) * (2 << 20)
            while capi.arena_block_zone(sbase - (2 << 20), 1)[0]:
                sbase -= 2 << 20
            zbase = sbase + (d_zmap.data_ptr() - sbase) // gb * gb
            per_gb = {}
            for name, cptr in (("map_chunk", zbase), ("next_chunk", zbase + gb)):
                inside, own, other = capi.arena_block_zone(cptr, gb)
                if not inside or own:
                    continue
                per_gb[name] = [round(4.0 * gb / (capi.probe_stream_split([lo + k * gb, cptr], gb) * 1e-3) / 1e12, 2)
                                for k in range(int((hi - lo) // gb))]
            zone_diag["read_GBs_vs_chunk"] = per_gb
            zone_diag["read_arrays_GB_offsets"] = {n: round((t.data_ptr() - lo) / gb, 2) for n, t in
                                                   (("pixels", d_pixels), ("weights", d_weights), ("tod", d_tod))}
        if os.environ["TOAST_BENCH_ZONE_DIAG"] in ("2", "3"):
            # does the level of build_noise_weighted follow the place of ITS MAP?  The same launch into maps at other places
            # of the arena (kept alive: every next one lies elsewhere): scatter blocks, streamed blocks, the read-mostly slab
            sweep, keep = [], []
            for kind in ("scatter", "scatter", "scatter", "scatter", "streamed", "streamed", "plain", "plain"):
                try:
                    z = manager_tensor(d_zmap.numel() * 8, torch.float64, tuple(d_zmap.shape), streamed=(kind == "streamed"),
                                       scatter=(kind == "scatter"))
                except Exception:
                    break
                keep.append(z)
                z.zero_()
                bnw_z = lambda: D.build_noise_weighted(d_g2l.data_ptr(), z.data_ptr(), nps, nnz, idx, d_pixels.data_ptr(), idx,
                                                       d_weights.data_ptr(), idx, d_tod.data_ptr(), idx, d_dflags.data_ptr(),
                                                       n_samp, det_scale, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, stream)
                bnw_z()
                inside, own, other = capi.arena_block_zone(z.data_ptr(), z.numel() * 8)
                sweep.append([kind, round(timed(bnw_z, 5), 3), "P" if own and not other else ("Q" if other and not own else "PQ"),
                              round((z.data_ptr() - d_zmap.data_ptr()) / gb, 2)])
            zone_diag["bnw_ms_by_map_place"] = sweep
            for z in keep:
                manager_release(z)

    # dominant kernel -> roofline entry (algorithmic bytes / measured launch duration)
    scan_bytes = BYTES_SCAN if not args.unfused else BYTES_SCAN  # noise_weight timed inside "scan" if unfused
    cand = {
        "build_noise_weighted": (BYTES_BNW * nsamp_tot, ms["bnw"]),
        "scan_map": ((scan_bytes + (16.0 if args.unfused else 0.0)) * nsamp_tot, ms["scan"]),
    }
    dom = max(cand, key=lambda k: cand[k][1])
    ach = cand[dom][0] / (cand[dom][1] * 1e-3) / 1e9
    # HBM traffic per launch of the dominant kernel from the committed PMC profile of this same
    # command (profiles/traffic_<workload>.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes,
    # gfx950 FETCH correction calibrated in-run; profiles/README.md); null if not profiled.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload)
    knames = {"build_noise_weighted": ("k_build_noise_weighted_v2<2>", "k_build_noise_weighted_pair<3>",
                                       "k_build_noise_weighted<3>"),
              "scan_map": ("k_scan_map_v2<double>", "k_scan_map<double, 3>")}[dom]
    if os.path.isfile(tpath) and not args.unfused:
        try:
            with open(tpath) as f:
                kernels = json.load(f)["kernels"]
            for kname in knames:
                if kname in kernels:
                    traffic = kernels[kname]["hbm_bytes"]
                    break
        except (KeyError, ValueError):
            traffic = None
    # the same from the counters that count fixed 32-byte units (profiles/r04_e: exact on known byte counts, no correction)
    traffic_exact = None
    xpath = os.path.join(ROOT, "profiles", "traffic_exact_%s.json" % workload)
    if os.path.isfile(xpath) and not args.unfused:
        try:
            with open(xpath) as f:
                kernels = json.load(f)["kernels"]
            for kname in knames:
                if kname in kernels:
                    traffic_exact = kernels[kname]["hbm_bytes"]
                    break
        except (KeyError, ValueError):
            traffic_exact = None
    roofline = {
        "bound": "hbm",
        "kernel": dom,
        "achieved": ach,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_exact": traffic_exact,
        "traffic_unit": "bytes per launch (rocprofv3 PMC, committed profile)",
        "algorithmic_bytes_per_launch": cand[dom][0],
        "per_kernel_GBs": {k: cand[k][0] / (cand[k][1] * 1e-3) / 1e9 for k in cand},
        "iteration_GBs": (BYTES_BNW + BYTES_SCAN) * nsamp_tot / ((ms["bnw"] + ms["scan"]) * 1e-3) / 1e9,
    }

    # measured stream ceiling on this box ("report % of both"): a pure
    # 8 B read + 8 B write stream (k_noise_weight over the work timestream, scale 1.0)
    ones = np.ones(n_det)
    D.noise_weight(d_tod2.data_ptr(), n_samp, idx, ivl, ones, stream)
    t_rw = timed(lambda: D.noise_weight(d_tod2.data_ptr(), n_samp, idx, ivl, ones, stream), 5)
    stream_rw = 16.0 * nsamp_tot / t_rw / 1e6
    roofline["stream_ceiling"] = {
        "read_write_GBs": stream_rw,
        "frac_of_read_write_stream": ach / stream_rw,
        "note": "k_noise_weight_v2, a pure read + write stream (16 B per lane and access) over the same work buffer",
    }
    if nnz == 3 and n_samp % 2 == 0 and hasattr(capi.real_lib(), "toast_hip_probe_byte_mix_dev"):
        # ... and the two kernels' OWN byte mixes as plain streams over the very arrays they read and write, in their launch
        # shape (no gather, no scan, no atomics; flags left out): what the places of these arrays in HBM let such a kernel reach
        mix_scan = lambda: capi.probe_byte_mix(d_pixels.data_ptr(), d_weights.data_ptr(), d_tod.data_ptr(), d_tod2.data_ptr(),    # noqa: E731
                                               n_det, n_samp, stream)
        mix_bnw = lambda: capi.probe_byte_mix(d_pixels.data_ptr(), d_weights.data_ptr(), d_tod.data_ptr(), 0, n_det, n_samp, stream)    # noqa: E731
        mix_scan()
        mix_bnw()
        t_ms, t_mb = timed(mix_scan, 5), timed(mix_bnw, 5)
        roofline["stream_ceiling"].update({
            "scan_map_byte_mix_ms": t_ms, "scan_map_byte_mix_GBs": 48.0 * nsamp_tot / t_ms / 1e6,
            "build_noise_weighted_byte_mix_ms": t_mb, "build_noise_weighted_byte_mix_GBs": 40.0 * nsamp_tot / t_mb / 1e6,
            "scan_map_frac_of_its_mix": t_ms / ms["scan"], "build_noise_weighted_frac_of_its_mix": t_mb / ms["bnw"],
            "byte_mix_note": "k_probe_byte_mix: 40 B read (+ 8 B written) per det-sample from / to the arrays of the timed kernels",
        })
    # the same three kernels with one sample per lane (8-byte lane accesses) and two consecutive samples per lane
    # (16-byte lane accesses; the default), same buffers, same process
    if headline:
        lanes = {}
        bnw_only = lambda: D.build_noise_weighted(d_g2l.data_ptr(), d_zmap.data_ptr(), nps, nnz, idx, d_pixels.data_ptr(),
                                                  idx, d_weights.data_ptr(), idx, d_tod.data_ptr(), idx,
                                                  d_dflags.data_ptr(), n_samp, det_scale, 1, n_samp, ivl,
                                                  d_sflags.data_ptr(), n_samp, 1, stream)
        scan_only = lambda: D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, d_tod2.data_ptr(), idx,
                                       d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False, True,
                                       False, det_w, stream)
        rw_only = lambda: D.noise_weight(d_tod2.data_ptr(), n_samp, idx, ivl, ones, stream)
        for v2, name in ((0, "one_sample_per_lane"), (1, "two_samples_per_lane")):
            capi.set_tuning("vec2", v2)
            for fn in (bnw_only, scan_only, rw_only):
                fn()
            t_b, t_s, t_r = timed(bnw_only, 3), timed(scan_only, 3), timed(rw_only, 3)
            lanes[name] = {"build_noise_weighted_ms": t_b, "scan_map_ms": t_s, "read_write_stream_ms": t_r,
                           "build_noise_weighted_GBs": BYTES_BNW * nsamp_tot / t_b / 1e6,
                           "scan_map_GBs": BYTES_SCAN * nsamp_tot / t_s / 1e6,
                           "read_write_stream_GBs": 16.0 * nsamp_tot / t_r / 1e6}
        capi.set_tuning("vec2", 1)
        roofline["lane_width"] = lanes

    out = {
        "metric": "detector-samples/sec through one PCG (A^T N^-1 A) iteration",
        "value": value,
        "unit": "det-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "detectors_per_gpu": n_det,
            "samples_per_detector": n_samp,
            "sample_rate_hz": rate,
            "nside": nside,
            "nnz": nnz,
            "n_local_submap": n_local,
            "noise_weight": "separate kernel" if args.unfused else "fused into scan_map",
            "parallelism": "detector-sharded x%d, RCCL all-reduce of zmap per step" % world,
        },
        "roofline": roofline,
        "kernel_ms": ms,
        "zone_diag": zone_diag,
        "expansion": {
            "pointing_detector_Gsamp_s": nsamp_tot / t_pd / 1e6,
            "pixels_healpix_Gsamp_s": nsamp_tot / t_pix / 1e6,
            "pixels_healpix_GBs": 40.0 * nsamp_tot / t_pix / 1e6,
            "stokes_weights_IQU_Gsamp_s": nsamp_tot / t_sw / 1e6,
            "three_kernel_chain_ms": t_pd + t_pix + t_sw,
            "from_boresight_pixels_ms": t_pix_x,
            "from_boresight_weights_ms": t_sw_x,
        },
        "setup_s": t_setup,
        "allocator": allocator,
        "placement": placement,
        # the arena's counters for this process so far: slabs taken from the driver, wall time inside hipMalloc
        "allocator_stats": capi.alloc_stats(),
        # per-step RCCL all-reduce of the device-resident zmap (fp64 sum over the detector shards);
        # kernel_ms.allreduce is its stream time on this rank (includes waiting for the slowest rank)
        "allreduce": {
            "bytes": int(n_local) * nps * nnz * 8 if multi else 0,
            "ms": ms["allreduce"] if multi else 0.0,
            "backend": (dist.get_backend() if multi else None),
            "implementation": comm_impl,
            "owner_computes_reduce_apply_ms": None,
            # the same pass per implementation (TOAST_HIP_COMM_MODE; "@16": TOAST_HIP_COMM_PEER_WIDTH=16): {"owner", "allreduce",
            # "peer", "peer:flags", "peer@16", "peer:flags@16"} and, for the peer modes, the plain all-reduce of the timed
            # step through the same exchange ("allreduce_via_<mode>")
            "reduce_apply_ms_by_mode": None,
            "note": comm_note,
            "algorithm_GBs": (2.0 * (world - 1) / world * n_local * nps * nnz * 8 / (ms["allreduce"] * 1e-3) / 1e9
                              if multi and world > 1 and ms["allreduce"] > 0 else None),
        },
    }

    if headline and world > 1:
        _lifeline_arm(out, rank)
    if comm_impl is not None and comm_impl.startswith("toast_hip_comm"):
        # the same sum + covariance as ONE owner-computes pass (reduce-scatter, cov_apply_diag on the owned pixel
        # shard, all-gather): what ops.BinMap / the fused SolverLHS do with sync_type = "alltoallv"
        oc = lambda: D.comm_map_reduce_apply(n_local * nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), True, stream)
        oc()
        owner_ms = timed(oc, 5)
        # ... and the other ways the library can do that pass (toast_hip_comm_set_mode), for an A/B on this very job: one
        # all-reduce followed by every rank multiplying the whole map, and the exchange over hipIpc-opened buffers ("peer":
        # every link of the xGMI mesh at once, RCCL only for the two barriers; "peer:flags": device flags instead), the
        # latter with 8- and 16-byte lane accesses
        mode_ms = {"owner": owner_ms}
        for mode, width in (("allreduce", 0), ("peer", 8), ("peer:flags", 8), ("peer", 16), ("peer:flags", 16)):
            key = mode if width in (0, 8) else "%s@%d" % (mode, width)
            try:
                if width:
                    D.comm_set_peer_width(width)
                D.comm_set_mode(mode)
                oc()
                mode_ms[key] = timed(oc, 5)
                if mode.startswith("peer"):
                    # the timed step's own all-reduce through the same exchange (toast_hip_comm_allreduce_dev in a peer mode)
                    D.comm_allreduce(d_zmap.data_ptr(), d_zmap.numel(), np.float64, "sum", stream)
                    mode_ms["allreduce_via_" + key] = timed(
                        lambda: D.comm_allreduce(d_zmap.data_ptr(), d_zmap.numel(), np.float64, "sum", stream), 5)
                    D.comm_check(stream)
            except RuntimeError as err:      # noqa: PERF203 -- the headline must not depend on this extra
                mode_ms[key] = repr(err)[:200]
        D.comm_set_peer_width(8)
        D.comm_set_mode("owner")
        out["allreduce"]["owner_computes_reduce_apply_ms"] = owner_ms
        out["allreduce"]["reduce_apply_ms_by_mode"] = mode_ms
        # What this very run says about scaling (weak: the same shard per GPU at every N): the step's own kernels next to
        # the reduction + covariance pass of each mode -- N x compute / (compute + reduction), nothing overlapped (the
        # scan needs the whole reduced map).  One driver run answers "does any mode clear 6 x at N = 8" (VERDICT round 5,
        # item 7 b); the measured curve is the driver's own, from the per-N values.
        compute_ms = ms["bnw"] + ms["cov"] + ms["scan"]
        out["allreduce"]["predicted_weak_scaling_by_mode"] = {
            k: world * compute_ms / (ms["bnw"] + ms["scan"] + v) for k, v in mode_ms.items()
            if isinstance(v, float) and not k.startswith("allreduce_via_")}
        out["allreduce"]["predicted_weak_scaling_note"] = (
            "N x (bnw + cov + scan) / (bnw + scan + reduce_apply_ms_by_mode[mode]) from this run's own kernel_ms; "
            "the headline value uses the plain all-reduce (kernel_ms.allreduce)")

    if os.environ.get("TOAST_BENCH_MAP_RUNS", "") != "" and not args.torch_alloc:
        # EXPERIMENT (profiles/r05_d): build_noise_weighted with its map in block after block of the scatter class, in ONE
        # process -- is the two-level behaviour of the kernel a property of where in the slab the map lies?
        runs = []
        fillers = []
        for k in range(int(os.environ["TOAST_BENCH_MAP_RUNS"])):
            zm = manager_tensor(n_local * nps * nnz * 8, torch.float64, (n_local, nps, nnz), scatter=True)
            zm.zero_()
            zone = capi.arena_block_zone(zm.data_ptr(), zm.numel() * 8)
            call = lambda: D.build_noise_weighted(d_g2l.data_ptr(), zm.data_ptr(), nps, nnz, idx, d_pixels.data_ptr(),
                                                  idx, d_weights.data_ptr(), idx, d_tod.data_ptr(), idx,
                                                  d_dflags.data_ptr(), n_samp, det_scale, 1, n_samp, ivl,
                                                  d_sflags.data_ptr(), n_samp, 1, stream)
            call()
            runs.append({"address": hex(zm.data_ptr()), "offset_GB_from_first": (zm.data_ptr() - d_zmap.data_ptr()) / 2.0 ** 30,
                         "zone": zone, "bnw_ms": timed(call, 5)})
            fillers.append(zm)
            # what is left of this 2 GB run is taken, so that the next map lands in the next run
            for pad in (int(1.0 * 2 ** 30), int(0.5 * 2 ** 30), int(0.25 * 2 ** 30)):
                blk = manager_tensor(pad, torch.uint8, (pad,), scatter=True)
                if abs(blk.data_ptr() - zm.data_ptr()) < 2 ** 31:
                    fillers.append(blk)
                else:
                    manager_release(blk)
        out["map_runs_experiment"] = runs
        del fillers

    # ------------------------------------------------------------------ FFT noise weighting (SURVEY.md section 8d:
    # "report separately"): ops.NoiseFilter's device call on the same timestream shape -- every detector
    # row convolved with its own N_tt'^-1 = NET^2 / PSD(f) kernel (reference src/toast/ops/noise_filter.py:130-188
    # -> toast.fft.convolve, src/toast/fft.py:252-350).  Not part of the PCG step; timed once per run.
    if not args.no_fft:
        from toast_amd import fft as hipfft
        from toast_amd.noise import AnalyticNoise

        nse = AnalyticNoise(rate={"d": rate}, fmin={"d": 1.0e-5}, detectors=["d"], fknee={"d": 0.05},
                            alpha={"d": 1.0}, NET={"d": 50.0e-6})
        kfreq = nse.freq("d")
        psd = nse.psd("d")
        net_sq = (50.0e-6) ** 2
        kern = net_sq / np.maximum(psd, 1.0e-3 * net_sq)
        kern[0] = 0.0
        kernels = np.tile(kern, (n_det, 1)) * np.linspace(0.9, 1.1, n_det)[:, None]   # one kernel per detector
        n_fft = hipfft.fft_length(n_samp)
        fft_call = lambda: hipfft.convolve_dev(d_tod2.data_ptr(), idx, n_samp, rate, kfreq, kernels, stream=stream)
        t_first = time.perf_counter()
        fft_call()                      # plans, rocFFT kernel cache, scratch buffers
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t_first
        t_fft_idle = timed(fft_call, 3)          # round 2-4's number: the first call's host preparation with the device idle
        t_fft = timed(fft_call, 3, prime=True)
        t_prep = time.perf_counter()
        hipfft.kernel_coefficients(kfreq, kernels)
        t_prep = time.perf_counter() - t_prep
        tot = float(n_det) * n_samp
        out["fft_noise_weight"] = {
            "ms": t_fft,
            # how "ms" is taken: three calls behind one un-timed, un-synchronised call, i.e. back to back on a busy queue.
            # ms_from_idle_queue: the same three calls started on an idle device (rounds 2-4 reported this one): the host's
            # PCHIP coefficients of the first call (host_prep_ms per call) are then waited for by an idle device
            "timing": "3 calls back to back behind one un-timed call",
            "ms_from_idle_queue": t_fft_idle,
            "host_prep_ms": 1e3 * t_prep,
            "first_call_ms": 1e3 * t_first,
            "samples_per_s": tot / (t_fft * 1e-3),
            "n_fft": int(n_fft),
            "implementation": hipfft.implementation(n_samp),
            # compulsory traffic: read + write each timestream sample once
            "hbm_GBs": 16.0 * tot / (t_fft * 1e-3) / 1e9,
            "frac": 16.0 * tot / (t_fft * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # what the passes of the pipeline move per timestream sample
            "pipeline_bytes_per_sample": hipfft.pipeline_bytes_per_sample(n_samp),
        }
        pb = out["fft_noise_weight"]["pipeline_bytes_per_sample"]
        if pb:
            out["fft_noise_weight"]["pipeline_GBs"] = pb * tot / (t_fft * 1e-3) / 1e9
            out["fft_noise_weight"]["pipeline_frac"] = pb * tot / (t_fft * 1e-3) / 1e9 / HBM_PEAK_GBS
        # ... and at the timestream length of the configs[3] shard (512 detectors x 2 880 000 samples, n_fft 2^23: column
        # transforms of 2048 points, where the passes' pieces are narrowest), on a buffer of its own
        if workload == "cfg3" and headline and world == 1 and not args.no_fft_long:
            ld, ls = 512, 2880000
            try:
                # (a timestream block of the arena, like the signal ops.NoiseFilter registers: streamed kind)
                if args.torch_alloc:
                    d_long = torch.empty((ld, ls), dtype=torch.float64, device=dev)
                else:
                    d_long = manager_tensor(ld * ls * 8, torch.float64, (ld, ls), streamed=True)
                d_long.normal_(0.0, 1.0)
                lidx = np.arange(ld, dtype=np.int32)
                lkern = np.tile(kern, (ld, 1)) * np.linspace(0.9, 1.1, ld)[:, None]
                long_call = lambda: hipfft.convolve_dev(d_long.data_ptr(), lidx, ls, rate, kfreq, lkern, stream=stream)
                long_call()
                torch.cuda.synchronize()
                t_long_idle = timed(long_call, 3)
                t_long = timed(long_call, 3, prime=True)
                lpb = hipfft.pipeline_bytes_per_sample(ls)
                out["fft_noise_weight"]["long"] = {
                    "n_det": ld, "n_samp": ls, "n_fft": int(hipfft.fft_length(ls)), "ms": t_long, "ms_from_idle_queue": t_long_idle,
                    "samples_per_s": float(ld) * ls / (t_long * 1e-3),
                    "pipeline_bytes_per_sample": lpb,
                    "pipeline_frac": (lpb * float(ld) * ls / (t_long * 1e-3) / 1e9 / HBM_PEAK_GBS) if lpb else None,
                }
                if not args.torch_alloc:
                    manager_release(d_long)
                del d_long
            except RuntimeError as err:      # noqa: PERF203 -- an extra: the headline must not depend on it
                out["fft_noise_weight"]["long"] = {"error": repr(err)[:200]}

    # ------------------------------------------------------------------ extra: full PCG LHS with offset templates
    # (not the headline metric) the complete SolverLHS of configs[2] "full MapMaker PCG":
    # a' = M^T N^-1 (M a - A C A^T N^-1 M a) with 1 s baselines, as the reference's operator
    # sequence (8 + 41 + 8 + 48 + 9 B per det-sample, five passes) and fused (33 + 33 B, two passes).
    # The complete left-hand side with one Offset template on the same buffers (operator sequence, fused sweeps, fused
    # sweeps from the packed pointing cache): part of every run; the on-the-fly / compact variants only with --pcg-extra.
    if not args.no_lhs and nnz == 3 and not args.torch_alloc:
        # the two set-up sweeps of a destriping run that read the same pixels / weights / flags -- solver covariance + hits,
        # then the right-hand side's A^T N^-1 d -- apart and as ONE sweep (toast_hip_build_cov_hits_signal_dev)
        d_invcov = manager_tensor(n_local * nps * 6 * 8, torch.float64, (n_local, nps, 6), scatter=True)
        d_hits = manager_tensor(n_local * nps * 8, torch.int64, (n_local, nps, 1), scatter=True)
        cov_args = lambda: (nps, nnz, idx, d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, d_tod.data_ptr(), idx,    # noqa: E731
                            d_dflags.data_ptr(), n_samp, det_scale, det_scale, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1,
                            stream)
        fused_flag = [False]

        def cov_only():
            d_invcov.zero_()
            d_hits.zero_()
            D.build_cov_hits_signal(d_g2l.data_ptr(), d_invcov.data_ptr(), d_hits.data_ptr(), 0, *cov_args())

        def cov_and_signal():
            d_invcov.zero_()
            d_hits.zero_()
            d_zmap.zero_()
            fused_flag[0] = D.build_cov_hits_signal(d_g2l.data_ptr(), d_invcov.data_ptr(), d_hits.data_ptr(),
                                                    d_zmap.data_ptr(), *cov_args())

        cov_only()
        ref_cov = d_invcov.clone()
        cov_and_signal()
        out["mapmaker_setup_sweeps"] = {
            "cov_hits_ms": timed(cov_only, 3), "signal_map_ms": ms["bnw"], "cov_hits_signal_ms": timed(cov_and_signal, 3),
            "one_kernel": bool(fused_flag[0]),
            "cov_max_rel_diff": float((d_invcov - ref_cov).abs().max() / ref_cov.abs().max()),
            "note": "inverse covariance + hits (33 B per det-sample) and the right-hand side's noise-weighted map (41 B) as "
                    "two sweeps and as one (42 B)",
        }
        del ref_cov
        manager_release(d_invcov)
        manager_release(d_hits)
        del d_invcov, d_hits
    if args.pcg_extra or not args.no_lhs:
        step_len = int(rate)  # 1 s baselines
        nav = np.array([(int(v["last"]) - int(v["first"]) + step_len - 1) // step_len for v in ivl], dtype=np.int64)
        n_amp_det = int(nav.sum())
        amp_off = np.arange(n_det, dtype=np.int64) * n_amp_det
        n_amp = n_det * n_amp_det
        d_amp_in = torch.randn(n_amp, dtype=torch.float64, device=dev, generator=gen)
        if args.torch_alloc:
            d_amp_out = torch.zeros(n_amp, dtype=torch.float64, device=dev)
        else:     # (what the projections scatter into: a scatter block, like ops' Amplitudes)
            d_amp_out = manager_tensor(n_amp * 8, torch.float64, (n_amp,), scatter=True)
            d_amp_out.zero_()
        d_amp_flags = torch.zeros(n_amp, dtype=torch.uint8, device=dev)

        def lhs_unfused():
            d_tod2.zero_()
            D.offset_add_to_signal_multi(step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_flags.data_ptr(), idx,
                                         d_tod2.data_ptr(), n_samp, ivl, stream)
            d_zmap.zero_()
            D.build_noise_weighted(d_g2l.data_ptr(), d_zmap.data_ptr(), nps, nnz, idx, d_pixels.data_ptr(), idx,
                                   d_weights.data_ptr(), idx, d_tod2.data_ptr(), idx, d_dflags.data_ptr(), n_samp,
                                   det_w, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, stream)
            allreduce_zmap()
            D.cov_apply_diag(n_local, nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), stream)
            D.scan_map(np.float64, d_g2l.data_ptr(), nps, d_zmap.data_ptr(), nnz, d_tod2.data_ptr(), idx,
                       d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, n_samp, ivl, 1.0, False, True, False,
                       det_w, stream)
            d_amp_out.zero_()
            D.offset_project_signal_multi(idx, d_tod2.data_ptr(), idx, d_dflags.data_ptr(), 1, step_len, amp_off, nav,
                                          d_amp_out.data_ptr(), d_amp_flags.data_ptr(), n_samp, ivl, stream)

        def lhs_fused():
            d_zmap.zero_()
            D.offset_accumulate(step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_flags.data_ptr(),
                                d_g2l.data_ptr(), d_zmap.data_ptr(), nps, nnz, idx, d_pixels.data_ptr(), idx,
                                d_weights.data_ptr(), idx, d_dflags.data_ptr(), n_samp, det_w, 1, n_samp, ivl,
                                d_sflags.data_ptr(), n_samp, 1, stream)
            allreduce_zmap()
            D.cov_apply_diag(n_local, nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), stream)
            d_amp_out.zero_()
            D.offset_scan_project(step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_out.data_ptr(),
                                  d_amp_flags.data_ptr(), d_g2l.data_ptr(), d_zmap.data_ptr(), nps, nnz, idx,
                                  d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, d_dflags.data_ptr(), 1, det_w,
                                  n_samp, ivl, stream)

        lhs_unfused()
        ref_out = d_amp_out.clone()
        t_unf = timed(lhs_unfused, 5)
        lhs_fused()
        err = float((d_amp_out - ref_out).abs().max() / ref_out.abs().max())
        t_fus = timed(lhs_fused, 5)
        out["pcg_lhs_offset_templates"] = {
            "baseline_step_samples": step_len,
            "amplitudes": int(n_amp),
            "operator_sequence_ms": t_unf,
            "fused_ms": t_fus,
            "operator_sequence_Gsamp_s": world * nsamp_tot / t_unf / 1e6,
            "fused_Gsamp_s": world * nsamp_tot / t_fus / 1e6,
            "fused_vs_sequence_max_rel_diff": err,
        }

        # the same left-hand side from the solver's packed pointing cache (csrc/packed_pointing.hip: 18-20 B per
        # det-sample instead of 33; what ops.SolverLHS sweeps after its first application)
        pk_key = temp(torch.int32, (n_det, n_samp))
        pk_qu = temp(torch.float64, (n_det, n_samp, 2))
        pk_cal = torch.empty(n_det, dtype=torch.float64, device=dev)
        # the whole pack in one sweep (round 6: pair words, Q / U rows and the pair weight sums -- the partner's Q / U as exact
        # float sums, 14 instead of 18 B per det-sample, refused unless exact -- straight from pixels / weights / flags);
        # pack_once_ms is the complete pack now (round 5: pack + pair check + merge, the sums in a fourth pass not counted)
        pk_corr = temp(torch.float32, ((n_det + 1) // 2, n_samp, 2))
        torch.cuda.synchronize()
        t0 = time.time()
        packable, pair_words, pair_sums = D.offset_pack_pointing_onepass(
            d_g2l.data_ptr(), nps, idx, d_pixels.data_ptr(), idx, d_weights.data_ptr(), idx, d_dflags.data_ptr(), n_samp,
            1, d_sflags.data_ptr(), n_samp, 1, idx, d_dflags.data_ptr(), n_samp, 1, n_samp, ivl, pk_key.data_ptr(),
            pk_qu.data_ptr(), pk_cal.data_ptr(), pk_corr.data_ptr(), stream=stream)
        t_pack = 1e3 * (time.time() - t0)
        corr_ptr = pk_corr.data_ptr() if (packable and pair_words and pair_sums) else 0
        if packable:
            use_corr = [0]

            def lhs_packed():
                d_zmap.zero_()
                D.offset_accumulate_packed(step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_flags.data_ptr(),
                                           d_zmap.data_ptr(), pk_key.data_ptr(), pk_qu.data_ptr(), pk_cal.data_ptr(),
                                           det_w, n_samp, ivl, pair_words=pair_words, pair_corr=use_corr[0], stream=stream)
                allreduce_zmap()
                D.cov_apply_diag(n_local, nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), stream)
                d_amp_out.zero_()
                D.offset_scan_project_packed(step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_out.data_ptr(),
                                             d_amp_flags.data_ptr(), d_zmap.data_ptr(), pk_key.data_ptr(),
                                             pk_qu.data_ptr(), pk_cal.data_ptr(), det_w, n_samp, ivl,
                                             pair_words=pair_words, pair_corr=use_corr[0], stream=stream)

            lhs_packed()
            err_pk = float((d_amp_out - ref_out).abs().max() / ref_out.abs().max())
            t_pk = timed(lhs_packed, 5)
            out["pcg_lhs_offset_templates"].update({
                "packed_ms": t_pk,
                "packed_Gsamp_s": world * nsamp_tot / t_pk / 1e6,
                "packed_bytes_per_det_sample_and_sweep": 18 if pair_words else 20,
                "packed_pair_words": bool(pair_words),
                "packed_vs_sequence_max_rel_diff": err_pk,
                "pack_once_ms": t_pack,
            })
            if comm_impl is not None and comm_impl.startswith("toast_hip_comm"):
                # the packed left-hand side with the owner-computes pass in the middle (what the fused SolverLHS runs with
                # sync_type = "alltoallv"), per implementation of that pass
                def lhs_packed_oc():
                    d_zmap.zero_()
                    D.offset_accumulate_packed(step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_flags.data_ptr(),
                                               d_zmap.data_ptr(), pk_key.data_ptr(), pk_qu.data_ptr(), pk_cal.data_ptr(),
                                               det_w, n_samp, ivl, pair_words=pair_words, pair_corr=corr_ptr, stream=stream)
                    if skip_reduce[0]:
                        D.cov_apply_diag(n_local, nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), stream)
                    else:
                        D.comm_map_reduce_apply(n_local * nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), True, stream)
                    d_amp_out.zero_()
                    D.offset_scan_project_packed(step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_out.data_ptr(),
                                                 d_amp_flags.data_ptr(), d_zmap.data_ptr(), pk_key.data_ptr(),
                                                 pk_qu.data_ptr(), pk_cal.data_ptr(), det_w, n_samp, ivl,
                                                 pair_words=pair_words, pair_corr=corr_ptr, stream=stream)

                by_mode = {}
                for mode, width in (("owner", 0), ("allreduce", 0), ("peer", 8), ("peer:flags", 8), ("peer", 16),
                                    ("peer:flags", 16)):
                    key = mode if width in (0, 8) else "%s@%d" % (mode, width)
                    try:
                        if width:
                            D.comm_set_peer_width(width)
                        D.comm_set_mode(mode)
                        lhs_packed_oc()
                        by_mode[key] = timed(lhs_packed_oc, 5)
                        D.comm_check(stream)
                    except RuntimeError as err:      # noqa: PERF203
                        by_mode[key] = repr(err)[:200]
                D.comm_set_peer_width(8)
                D.comm_set_mode("owner")
                out["pcg_lhs_offset_templates"]["packed_ms_by_mode"] = by_mode
                # ... and the same two sweeps + cov_apply_diag with no exchange at all: the PCG iteration's compute part,
                # i.e. what one GPU does for its shard; N x that / packed_ms_by_mode[mode] = the scaling this run predicts
                skip_reduce[0] = True
                try:
                    lhs_packed_oc()
                    t_sweeps = timed(lhs_packed_oc, 5)
                finally:
                    skip_reduce[0] = False
                out["pcg_lhs_offset_templates"]["packed_sweeps_only_ms"] = t_sweeps
                out["pcg_lhs_offset_templates"]["predicted_weak_scaling_by_mode"] = {
                    k: world * t_sweeps / v for k, v in by_mode.items() if isinstance(v, float)}
            if corr_ptr:
                # the same sweeps with the partner's weights rebuilt from the pair sums: what ops.SolverLHS runs when the
                # pairs allow it (packed_ms above is then the 18-byte form, kept for comparison)
                use_corr[0] = corr_ptr
                lhs_packed()
                err_pw = float((d_amp_out - ref_out).abs().max() / ref_out.abs().max())
                t_pw = timed(lhs_packed, 5)
                out["pcg_lhs_offset_templates"].update({
                    "packed_pair_weights_ms": t_pw,
                    "packed_pair_weights_Gsamp_s": world * nsamp_tot / t_pw / 1e6,
                    "packed_pair_weights_bytes_per_det_sample_and_sweep": 14,
                    "packed_pair_weights_vs_sequence_max_rel_diff": err_pw,
                })
        if pk_corr is not None:
            drop(pk_corr)
        drop(pk_key)
        drop(pk_qu)
        del pk_key, pk_qu, pk_cal

    if args.pcg_extra:
        # pointing on the fly: the same two operators and the same LHS
        # without the 32 B/det-sample pointing cache -- 9 + 16 B (A^T, A) and ~2 + ~2 B (offset LHS)
        pt = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sflags.data_ptr(),
                               n_shared_flags=n_samp, shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma,
                               cal=np.ones(n_det), d_hwp=hwp_ptr, n_hwp=hwp_n, d_hwp_table=hwp_tab_ptr)

        def ata_otf():
            d_zmap.zero_()
            D.otf_build_noise_weighted(pt, d_g2l.data_ptr(), d_zmap.data_ptr(), nps, idx, d_tod.data_ptr(), idx,
                                       d_dflags.data_ptr(), n_samp, det_scale, 1, n_samp, ivl, d_sflags.data_ptr(),
                                       n_samp, 1, stream)
            allreduce_zmap()
            D.cov_apply_diag(n_local, nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), stream)
            D.otf_scan_map(pt, d_g2l.data_ptr(), d_zmap.data_ptr(), nps, d_tod2.data_ptr(), idx, n_samp, ivl, 1.0,
                           False, True, det_w, stream)

        def lhs_otf():
            d_zmap.zero_()
            D.otf_offset_accumulate(pt, step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_flags.data_ptr(),
                                    d_g2l.data_ptr(), d_zmap.data_ptr(), nps, idx, d_dflags.data_ptr(), n_samp,
                                    det_w, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, stream)
            allreduce_zmap()
            D.cov_apply_diag(n_local, nps, nnz, d_cov.data_ptr(), d_zmap.data_ptr(), stream)
            d_amp_out.zero_()
            D.otf_offset_scan_project(pt, step_len, amp_off, nav, d_amp_in.data_ptr(), d_amp_out.data_ptr(),
                                      d_amp_flags.data_ptr(), d_g2l.data_ptr(), d_zmap.data_ptr(), nps, idx,
                                      d_dflags.data_ptr(), n_samp, 1, det_w, n_samp, ivl, stream)

        lhs_otf()
        err_otf = float((d_amp_out - ref_out).abs().max() / ref_out.abs().max())
        t_lhs_otf = timed(lhs_otf, 5)
        ata_otf()
        t_ata_otf = timed(ata_otf, 5)
        t_acc = timed(lambda: D.otf_build_noise_weighted(
            pt, d_g2l.data_ptr(), d_zmap.data_ptr(), nps, idx, d_tod.data_ptr(), idx, d_dflags.data_ptr(), n_samp,
            det_scale, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, stream), 3)
        # compact mode: int32 local pixel index cache (4 B) + weights on the fly
        d_cpix = torch.empty((n_det, n_samp), dtype=torch.int32, device=dev)
        D.compact_pixels(d_g2l.data_ptr(), nps, n_local, idx, d_pixels.data_ptr(), idx, d_cpix.data_ptr(), n_samp, ivl,
                         stream)
        ptc = capi.otf_pointing(d_bore.data_ptr(), fp, nside, True, nnz, d_shared_flags=d_sflags.data_ptr(),
                                n_shared_flags=n_samp, shared_flag_mask=1, epsilon=np.zeros(n_det), gamma=gamma,
                                cal=np.ones(n_det), d_compact_pixels=d_cpix.data_ptr(), compact_index=idx,
                                d_hwp=hwp_ptr, n_hwp=hwp_n, d_hwp_table=hwp_tab_ptr)
        pt_keep = pt
        pt = ptc
        lhs_otf()
        err_c = float((d_amp_out - ref_out).abs().max() / ref_out.abs().max())
        t_lhs_c = timed(lhs_otf, 5)
        ata_otf()
        t_ata_c = timed(ata_otf, 5)
        t_acc_c = timed(lambda: D.otf_build_noise_weighted(
            ptc, d_g2l.data_ptr(), d_zmap.data_ptr(), nps, idx, d_tod.data_ptr(), idx, d_dflags.data_ptr(), n_samp,
            det_scale, 1, n_samp, ivl, d_sflags.data_ptr(), n_samp, 1, stream), 3)
        t_scan_c = timed(lambda: D.otf_scan_map(ptc, d_g2l.data_ptr(), d_zmap.data_ptr(), nps, d_tod2.data_ptr(), idx,
                                                n_samp, ivl, 1.0, False, True, det_w, stream), 3)
        pt = pt_keep
        out["compact_pixels_weights_on_the_fly"] = {
            "bytes_per_det_sample": {"accumulate": 13, "scan": 20, "offset_lhs": 10},
            "ata_ms": t_ata_c,
            "ata_Gsamp_s": world * nsamp_tot / t_ata_c / 1e6,
            "accumulate_ms": t_acc_c,
            "scan_ms": t_scan_c,
            "offset_lhs_ms": t_lhs_c,
            "offset_lhs_Gsamp_s": world * nsamp_tot / t_lhs_c / 1e6,
            "offset_lhs_vs_sequence_max_rel_diff": err_c,
        }
        out["pointing_on_the_fly"] = {
            "ata_ms": t_ata_otf,
            "ata_Gsamp_s": world * nsamp_tot / t_ata_otf / 1e6,
            "otf_build_noise_weighted_ms": t_acc,
            "offset_lhs_ms": t_lhs_otf,
            "offset_lhs_Gsamp_s": world * nsamp_tot / t_lhs_otf / 1e6,
            "offset_lhs_vs_sequence_max_rel_diff": err_otf,
            "cached_expand_plus_ata_ms": t_pd + t_pix + t_sw + 1e3 * elapsed / args.steps,
        }

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1)
    if world == 1 and rank == 0 and headline and not args.no_cpu_baseline:
        import oracle

        nd = min(args.cpu_dets, n_det)
        sub = slice(0, nd)
        pix_h = d_pixels[sub].cpu().numpy()
        w_h = d_weights[sub].cpu().numpy()
        tod_h = d_tod[sub].cpu().numpy()
        tod2_h = d_tod2[sub].cpu().numpy()
        df_h = d_dflags[sub].cpu().numpy()
        z_h = np.zeros((n_local, nps, nnz))
        cov_h = d_cov.cpu().numpy()
        idx_h = np.arange(nd, dtype=np.int32)
        ds_h = np.ascontiguousarray(det_scale[:nd])
        dw_h = np.ascontiguousarray(det_w[:nd])

        def make_step(impl, tail):
            scan = getattr(impl, "scan_map", None) or getattr(impl, "ops_scan_map_float64")

            def cpu_step():
                z_h[:] = 0.0
                impl.build_noise_weighted(g2l_h, z_h, idx_h, pix_h, idx_h, w_h, idx_h, tod_h, idx_h, df_h, ds_h, 1,
                                          ivl, sflags_h, 1, *tail)
                scan(g2l_h, nps, z_h, tod2_h, idx_h, pix_h, idx_h, w_h, idx_h, ivl, 1.0, False, True, False, *tail)
                impl.noise_weight(tod2_h, idx_h, ivl, dw_h, *tail)

            return cpu_step

        def time_cpu(cpu_step, budget_s):
            cpu_step()  # warm (page-touch)
            t0 = time.perf_counter()
            reps = 0
            while True:
                cpu_step()
                reps += 1
                if time.perf_counter() - t0 > budget_s or reps >= 5:
                    break
            return (time.perf_counter() - t0) / reps

        # the reference's own compiled host path (oracle/_ref, built from /root/reference's sources by
        # oracle/ref_build.sh and shipped as a prebuilt .so) when present, else our restatement of it
        ref = None
        try:
            ref = oracle.load_ref()
        except Exception:  # an unloadable .so must not take the bench down
            ref = None
        sample = ("%d of the %d detectors x %d samples of the same workload; build_noise_weighted + "
                  "scan_map(subtract) + noise_weight on the host path (cov_apply_diag excluded: map sized)"
                  % (nd, n_det, n_samp))
        impl, tail, kind = (ref, (False,), "reference") if ref is not None else (oracle, (), "port")
        n_all = oracle.num_threads()
        # thread sweep (SURVEY.md section 8d asks for the all-core and the 1-thread number; the reference's
        # build_noise_weighted host path lets every thread scan all samples -- T-fold redundant reads -- so
        # more threads are not faster: report the whole curve and the best point)
        sweep = {}
        for nt in sorted({1, 8, 32, n_all} & set(range(1, n_all + 1))):
            oracle.set_num_threads(nt)
            sweep[nt] = nd * n_samp / time_cpu(make_step(impl, tail), 3.0 if nt > 1 else 6.0)
        oracle.set_num_threads(n_all)
        best = max(sweep, key=lambda k: sweep[k])
        # `value` is the BEST point of the thread sweep with `cores` = the threads it used: the reference's host path lets
        # every thread scan every sample of build_noise_weighted (ops_mapmaker_utils.cpp:295-377), so all cores of a
        # 128-thread box are its pathological case, not its baseline (kept as all_cores / thread_sweep)
        out["cpu_baseline"] = {
            "value": sweep[best],
            "unit": "det-samples/s",
            "cores": int(best),
            "kind": kind,
            "sample": sample + ("; libtoast's own kernels compiled from the reference sources (use_accel=False)"
                                if ref is not None else "; our restatement with the reference's host parallelisation")
                      + "; best of a thread sweep",
            "one_thread": sweep.get(1),
            "all_cores": {"threads": int(n_all), "value": sweep[n_all]},
            "best_threads": {"threads": int(best), "value": sweep[best]},
            "thread_sweep": {str(k): v for k, v in sweep.items()},
        }
        if ref is not None:
            out["cpu_baseline"]["port_value"] = nd * n_samp / time_cpu(make_step(oracle, ()), 3.0)
        del cov_h

    # hand the manager's blocks back (the views above die with this frame; nothing is running any more)
    torch.cuda.synchronize()
    del d_pixels, d_weights, d_tod, d_tod2, d_dflags
    for ptr in managed:
        capi.device_free(ptr)
    # (the slabs stay with the arena: a second workload of this process -- the configs[3] shard -- is carved from them)
    return out


if __name__ == "__main__":
    main()
