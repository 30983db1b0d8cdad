"""GPU: the library's communicator (toast_amd/csrc/comm.cpp) with MORE THAN ONE RANK on the single GPU of the test
box.  RCCL refuses two ranks on one device, so these runs put a shared-memory stand-in for librccl
(tests/rccl_mock.cpp, built here with g++) behind ``TOAST_HIP_RCCL_LIB``: everything above the nccl* calls is the
shipped code -- pixel shards of sizes that do not divide by the number of ranks (padded scratch path), owner-computes
kernels on real shards, the PCG dot products summed over ranks on the stream, the unique-id bootstrap over the process
group -- with 2 and 3 ranks.  The python-level process group is gloo; ``TOAST_HIP_COMM=rccl`` sends device-resident
collectives through the library's communicator all the same."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MOCK = os.path.join(HERE, "librccl_mock.so")


@pytest.fixture(scope="module")
def mock_lib():
    src = os.path.join(HERE, "rccl_mock.cpp")
    if not os.path.exists(MOCK) or os.path.getmtime(MOCK) < os.path.getmtime(src):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        src, "-o", MOCK, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"],
                       check=True)
    return MOCK


def _run(n, script, port, mock, timeout=1500, **extra):
    env = dict(os.environ)
    env.update(OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TOAST_TEST_BACKEND="gloo", TOAST_HIP_COMM="rccl",
               TOAST_HIP_RCCL_LIB=mock)
    env.update(extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, script)]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("n", [2, 3, 5])
def test_map_reductions_and_owner_computes(mock_lib, n):
    # 37 submaps x 48 pixels: whole shards with 2 and 3 ranks (collectives in place on the map), ceil(1776 / 5) = 356
    # pixels per rank with 5 (zero-padded scratch copy, the last rank owns 352)
    out = _run(n, "rccl_worker.py", 29561 + n, mock_lib)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == n


@pytest.mark.parametrize("n", [2, 5])
def test_map_reductions_with_fine_grained_exchange_buffers(mock_lib, n):
    """The same worker with TOAST_HIP_COMM_PEER_MEM=fine: the peer modes' exchange buffers in fine-grained device memory
    (hipDeviceMallocFinegrained), both access widths, opened over hipIpc by the other ranks -- the A/B switch for the first
    run on several GPUs (VERDICT round 5, item 7 a); same results bit for bit, and the library reports the kind it took."""
    out = _run(n, "rccl_worker.py", 29591 + n, mock_lib, TOAST_HIP_COMM_PEER_MEM="fine")
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == n


@pytest.mark.parametrize("n", [2, 3])
def test_mapmaker_equals_single_process(mock_lib, n):
    out = _run(n, "dist_gpu_worker.py", 29571 + n, mock_lib)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == n


def test_mapmaker_equals_single_process_in_peer_mode(mock_lib):
    """The same map-maker run with every map reduction of the solve going through the hipIpc exchange buffers
    (TOAST_HIP_COMM_MODE=peer): three processes writing into and reading from each other's device memory."""
    out = _run(3, "dist_gpu_worker.py", 29579, mock_lib, TOAST_HIP_COMM_MODE="peer", TOAST_TEST_EXPECT_PEER="1")
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == 3


def test_peer_flags_give_up_loudly_when_a_rank_does_not_arrive(mock_lib):
    out = _run(2, "peer_timeout_worker.py", 29581, mock_lib, timeout=300, TOAST_HIP_COMM_PEER_TIMEOUT_MS="300")
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == 2


@pytest.mark.parametrize("n", [2, 3])
def test_bench_ranks_through_the_library_communicator(mock_lib, n):
    """`python3 bench.py --gpus N` (self-launched) with all ranks on the one GPU (N = 3: shards that do not divide): the N > 1 protocol of the benchmark
    -- every rank agrees that the library can be loaded, rank 0's id goes round, collective initialisation, the
    self-check of all-reduce and owner-computes pass against torch.distributed, the agreement to use it -- with two
    REAL ranks; the timed step then reduces the map through toast_hip_comm_allreduce_dev."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    env.update(TOAST_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1", TOAST_HIP_RCCL_LIB=mock_lib)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup",
                          "1", "--workload", "mini", "--no-fft", "--shard-workload", "cfg2"], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["allreduce"]["implementation"].startswith("toast_hip_comm"), d["allreduce"]
    assert d["allreduce"]["note"] is None and d["allreduce"]["owner_computes_reduce_apply_ms"] > 0
    # every implementation of the owner-computes pass was timed on this job (A/B material for the first 8-GPU lease),
    # and the packed left-hand side with the reduction inside is part of the same line
    modes = d["allreduce"]["reduce_apply_ms_by_mode"]
    assert set(modes) == {"owner", "allreduce", "peer", "peer:flags", "peer@16", "peer:flags@16", "allreduce_via_peer",
                          "allreduce_via_peer:flags", "allreduce_via_peer@16", "allreduce_via_peer:flags@16"}
    assert all(isinstance(v, float) and v > 0 for v in modes.values()), modes
    assert d["pcg_lhs_offset_templates"]["packed_ms"] > 0
    by_mode = d["pcg_lhs_offset_templates"]["packed_ms_by_mode"]
    assert set(by_mode) == {"owner", "allreduce", "peer", "peer:flags", "peer@16", "peer:flags@16"}
    assert all(isinstance(v, float) and v > 0 for v in by_mode.values()), by_mode
    assert set(d["configs3_shard"]["pcg_lhs_offset_templates"]["packed_ms_by_mode"]) == set(by_mode)
    assert d["configs3_shard"]["allreduce"]["implementation"].startswith("toast_hip_comm")


def test_workflow_two_ranks_equals_one_process(mock_lib):
    """`workflows/mapmaker_pcg.py` under torch.distributed.run with two ranks (README's multi-GPU command, here on one
    GPU through the stand-in): 2 x 32 detectors detector-sharded against one process with all 64 -- the same PCG
    trajectory (the workflow prints the last relative residual) and the same number of amplitudes."""
    import re

    args = ["--minutes", "5", "--rate", "100", "--nside", "256", "--iter", "8", "--no-filter", "--step-time", "10"]
    env = dict(os.environ)
    env.update(OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "workflows", "mapmaker_pcg.py"), "--ndet", "64", *args],
                         capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    env.update(TOAST_BENCH_SHARE_GPU="1", TOAST_HIP_COMM="rccl", TOAST_HIP_RCCL_LIB=mock_lib, TOAST_HIP_TRACE="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29581", os.path.join(ROOT, "workflows", "mapmaker_pcg.py"), "--ndet", "32",
           *args]
    two = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]

    def summary(text):
        m = re.search(r"ranks (\d+)\s+detectors (\d+).*amplitudes (\d+)\s+PCG iterations (\d+)\s+relative residual (\S+)",
                      text)
        assert m, text[-1500:]
        return int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5))

    r1, r2 = summary(one.stdout), summary(two.stdout)
    assert r1[0] == 1 and r2[0] == 2 and r1[1:4] == r2[1:4] == (64, r1[2], 8)
    assert abs(r2[4] - r1[4]) < 1e-6 * r1[4], (r1, r2)
    # the collectives of the two-rank run went through the library's communicator, on the device
    log = two.stderr + two.stdout
    assert "toast_hip_comm_map_reduce_apply_dev" in log and "toast_hip_comm_allreduce_dev" in log
