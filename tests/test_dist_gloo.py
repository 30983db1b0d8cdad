"""CPU: the N>1 path with world_size 2 over the gloo backend (no GPU needed)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_process_collectives():
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(HERE, "dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("OK") == 2
