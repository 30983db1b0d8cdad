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


def _run(n, script, port, mock, timeout=1500):
    env = dict(os.environ)
    env.update(OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", TOAST_TEST_BACKEND="gloo", TOAST_HIP_COMM="rccl",
               TOAST_HIP_RCCL_LIB=mock)
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


@pytest.mark.parametrize("n", [2, 3])
def test_mapmaker_equals_single_process(mock_lib, n):
    out = _run(n, "dist_gpu_worker.py", 29571 + n, mock_lib)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == n
