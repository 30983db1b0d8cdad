"""GPU: the detector-sharded N > 1 path end to end with two processes sharing the one GPU of the
test box (collectives over gloo; on a multi-GPU node the same code runs one process per GPU
over RCCL) against the single-process run -- see tests/dist_gpu_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_process_mapmaker_equals_single_process():
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(HERE, "dist_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert out.stdout.count("OK") == 2
