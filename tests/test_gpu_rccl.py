"""RCCL (torch.distributed backend "nccl") on real devices: runs only where at least two GPUs are visible (the one-GPU
test boxes skip it; the gloo tests in test_parallel_gloo.py cover the same code path on CPU)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_rccl_two_rank_gradient_equals_big_batch_gradient():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", "NCCL_DEBUG": os.environ.get("NCCL_DEBUG", "WARN")}
    env.pop("MULAN_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tests", "rccl_grad_check.py")],
                       capture_output=True, text=True, timeout=840, env=env, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_GRAD_CHECK ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
