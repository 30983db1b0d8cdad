"""GPU: the RCCL branch of the multi-GPU map reduction (PixelData.sync_allreduce /
sync_alltoallv on device tensors, backend "nccl").  A one-rank process group runs on any GPU box;
the two-rank run and ``bench.py --gpus 2`` need two GPUs and are skipped otherwise, so that the
first multi-GPU machine exercises the path under pytest, not under the timed bench."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _n_gpus():
    import torch

    return torch.cuda.device_count()


def _env():
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_rccl_single_rank_process_group():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py")], capture_output=True, text=True,
                         env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert "rank 0 of 1 OK" in out.stdout


def test_rccl_single_rank_mapmaker_through_the_device_communicator():
    """tests/dist_gpu_worker.py with ONE rank over `nccl`: the complete MapMaker with every multi-process branch taken
    on the device (owner-computes reductions in BinMap / CovarianceAndHits / the fused left-hand side, PCG dot products
    summed on the stream) equals the plain single-process run."""
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29551",
               TOAST_TEST_BACKEND="nccl")
    out = subprocess.run([sys.executable, os.path.join(HERE, "dist_gpu_worker.py")], capture_output=True, text=True,
                         env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert "rank 0 OK" in out.stdout


def _torchrun(n, script, *args, port=29543, timeout=1200):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), script, *args]
    return subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=timeout, cwd=ROOT)


def test_rccl_two_ranks_map_reductions():
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    out = _torchrun(2, os.path.join(HERE, "rccl_worker.py"))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == 2


def test_rccl_two_ranks_mapmaker_equals_single_process():
    """tests/dist_gpu_worker.py (complete MapMaker, detector-sharded) with one process per GPU and
    RCCL collectives instead of two processes on one GPU with gloo."""
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    env = _env()
    env["TOAST_TEST_BACKEND"] = "nccl"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29545", os.path.join(HERE, "dist_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == 2


def test_bench_two_gpus_over_rccl():
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    # the driver's own command form: bench.py launches its ranks itself
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
                          "1", "--workload", "mini", "--no-cpu-baseline", "--no-fft"], capture_output=True, text=True,
                         env=_env(), timeout=1200, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["allreduce"]["backend"] == "nccl"
    assert line["allreduce"]["bytes"] > 0 and line["kernel_ms"]["allreduce"] > 0
