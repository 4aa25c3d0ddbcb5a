#!/usr/bin/env python3
"""Launched by tests/test_gpu_rccl.py (one process per rank, backend nccl = RCCL; on a one-GPU box: ONE rank whose reducer
is told it is one of two): the train step replayed as ONE HIP graph with the bucket all-reduces CAPTURED INTO it
(MULAN_GRAPH_COLLECTIVES / config.training.graph_collectives: the hooks launch the collectives during the capture, the
optimizer is part of the graph) must leave the state bit-identical to the eager overlapped step -- parameters, EMA, Adam
moments, reduced gradient, logged bits/dim, over four optimizer steps.  Prints `CAPTURED_COLLECTIVES_CHECK ok ...`."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = 4


def run(captured, fake_world):
    from mulan_amd import parallel
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = "synthetic"
    config.model.sm_n_layer = 2
    config.model.forward_n_layer = 1
    world = max(parallel.world_size(), fake_world)
    config.training.batch_size_train = 4 * world
    config.training.batch_size_eval = 4 * world
    config.training.substeps = 1
    config.training.num_steps_lr_warmup = 2
    config.training.hip_graph = bool(captured)
    config.training.graph_collectives = bool(captured)
    config.optimizer.ema_rate = 0.9
    exp = Experiment_VDM(config)
    if fake_world > exp.world:                     # one real rank: the step applies 1 / N, the optimizer leaves the eager path
        exp.world = fake_world
        exp.graph_collectives = bool(captured)
    g = torch.Generator().manual_seed(100 + exp.rank)
    state, snaps = exp.state, []
    for _ in range(STEPS):
        batch = {"images": torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8).to(exp.device),
                 "labels": torch.zeros(4, dtype=torch.int32, device=exp.device),
                 "conditioning": torch.zeros(4, dtype=torch.uint8, device=exp.device)}
        state, m = exp.train_step(exp._train_rng, state, batch)
        torch.cuda.synchronize()
        snaps.append(tuple(t.clone() for t in (state.flat, state.ema, state.mu, state.nu, state.grad)) +
                     (float(m["scalars"]["train_bpd"]),))
    info = dict(graphed=exp._graphed is not None, captured=bool(exp._graphed is not None and exp._graphed.captured_collectives),
                buckets=len(exp.reducer.buckets), error=exp.graph_capture_error)
    exp._graphed = None
    del exp
    torch.cuda.empty_cache()
    return snaps, info


def main():
    from mulan_amd import parallel
    os.environ.setdefault("MULAN_BUCKET_MB", "16")
    if "RANK" not in os.environ:                   # plain `python tests/captured_collectives_check.py`: one rank
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if int(os.environ["WORLD_SIZE"]) == 1:
        dist.init_process_group("nccl", rank=0, world_size=1)
        real = parallel.world_size
        parallel.world_size = lambda: 2            # GradReducer / the scalar mean believe in two ranks (the sum over one is the identity)
        fake = 2
    else:
        parallel.init_distributed(backend="nccl")
        fake = 0
    eager, _ = run(False, fake)
    cap, info = run(True, fake)
    bad = []
    if not (info["graphed"] and info["captured"] and info["buckets"] > 1):
        bad.append(f"not captured: {info}")
    for k, (a, b) in enumerate(zip(eager, cap)):
        for name, x, y in zip(("params", "ema", "mu", "nu", "grad"), a[:5], b[:5]):
            if not torch.equal(x, y):
                bad.append(f"step {k} {name}: {int((x != y).sum())} of {x.numel()} differ, max {float((x - y).abs().max()):.3e}")
        if a[5] != b[5]:
            bad.append(f"step {k} bpd {a[5]} vs {b[5]}")
    ok = torch.tensor([0.0 if bad else 1.0], device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if dist.get_rank() == 0:
        print(f"CAPTURED_COLLECTIVES_CHECK {'ok' if float(ok[0]) == 1.0 else 'FAILED'} ranks={dist.get_world_size()} "
              f"buckets={info['buckets']} " + "; ".join(bad[:6]), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if float(ok[0]) == 1.0 else 1)


if __name__ == "__main__":
    main()
