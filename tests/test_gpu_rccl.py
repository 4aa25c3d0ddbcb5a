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
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0", "NCCL_DEBUG": os.environ.get("NCCL_DEBUG", "WARN")}
    env.pop("MULAN_DIST_BACKEND", None)
    # (--standalone: torchrun picks the rendezvous port itself, on the loop-back address -- as bench.py's self-launch does)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
                        "--nnodes=1", "--nproc-per-node", "2", os.path.join(ROOT, "tests", "rccl_grad_check.py")],
                       capture_output=True, text=True, timeout=840, env=env, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_GRAD_CHECK ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.timeout(600)
def test_rccl_single_rank_reducer_smoke():
    """What a one-GPU box can say about the RCCL path: backend "nccl" (= RCCL on ROCm) initialises in this image, and
    GradReducer's bucketed side-stream all-reduce + wait runs on it (world size 1: the sum is the identity), including
    the stream hand-over with the weight-gradient stream.  The arithmetic of N > 1 is covered by the gloo tests and by
    rccl_grad_check.py wherever two GPUs are visible."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from mulan_amd import parallel
flat = torch.arange(0, 3 << 20, device="cuda", dtype=torch.float32)
leaves = [(flat[i << 20:(i + 1) << 20].detach().requires_grad_(True), i << 20, 1 << 20) for i in range(3)]
red = parallel.GradReducer(flat, leaves, bucket_bytes=4 << 20)
red.world, red.enabled = 2, True            # exercise the launch path on the one-rank group
red.side = red.side or torch.cuda.Stream()
want = flat.clone()
for sync_ops in (True, False):              # the collective as a synchronous op on the side stream (what a multi-rank RCCL
    red.sync_ops = sync_ops                 # job uses) and as an asynchronous op + wait()
    red.prepare()
    red.finish()
    torch.cuda.synchronize()
    assert torch.equal(flat, want) and len(red.buckets) == 3 and sorted(red.ready_order) == [0, 1, 2]
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier()
print("RCCL_SMOKE ok", dist.get_backend(), torch.cuda.get_device_name(0))
dist.destroy_process_group()
'''
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MULAN_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "-c", code % (ROOT, port)], capture_output=True, text=True, timeout=540, env=env,
                       cwd=ROOT)
    assert r.returncode == 0 and "RCCL_SMOKE ok nccl" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.timeout(900)
def test_rccl_collectives_captured_into_the_replayed_step():
    """MULAN_GRAPH_COLLECTIVES (round 5, opt-in): the whole multi-rank train step as ONE HIP graph -- the bucket all-reduces
    are captured where the eager step's hooks launch them, the optimizer follows inside the graph; no signal words, no
    stream calibration.  ProcessGroupNCCL supports capture, gloo does not, so this runs on RCCL: with two ranks where two
    GPUs are visible, else with one rank whose reducer is told it is one of two (the collective path, streams and the
    1 / N of the optimizer are the real ones; the sum over one rank is the identity).  Four optimizer steps end
    bit-identical to the eager overlapped step (tests/captured_collectives_check.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MULAN_DIST_BACKEND",
                                                            "MULAN_HIP_GRAPH", "MULAN_GRAPH_OVERLAP")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    script = os.path.join(ROOT, "tests", "captured_collectives_check.py")
    if torch.cuda.device_count() >= 2:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", "2", script]
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
        cmd = [sys.executable, script]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=840, env=env, cwd=ROOT)
    assert r.returncode == 0 and "CAPTURED_COLLECTIVES_CHECK ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
