#!/usr/bin/env python3
"""Launched by tests/test_gpu_model.py::test_two_rank_replayed_step_overlaps_the_allreduce_and_equals_the_eager_step under
torch.distributed.run with two ranks (on one GPU: MULAN_DIST_BACKEND=gloo + MULAN_FORCE_DEVICE=0, the existing hook; on a
multi-GPU node: backend nccl = RCCL).  The multi-rank train step as a replayed HIP graph whose bucketed gradient all-reduce
is issued outside the graph but waits only for the event node planted behind each bucket (parallel.GradReducer
begin_capture / _mark / allreduce_captured; the reference's pmap(scan(train_step)) has both the single dispatch and the
overlapped pmean, ldm/experiment.py:89-95,341) must reproduce the eager overlapped step BIT FOR BIT: parameters, EMA,
Adam moments and the reduced gradient after every step.  Prints `OVERLAP_REPLAY_CHECK ok ...` on rank 0."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STEPS = 4


def run(graph, vdm_type="mulan_epsilon"):
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.vdm_type = vdm_type
    config.data.dataset = "synthetic"
    config.model.sm_n_layer = 2
    config.model.forward_n_layer = 1
    config.training.batch_size_train = 8
    config.training.batch_size_eval = 8
    config.training.substeps = 1
    config.training.num_steps_lr_warmup = 2
    config.training.hip_graph = graph
    config.training.graph_overlap = True      # (opt-in since round 5: the hand-off under test)
    config.optimizer.ema_rate = 0.9
    exp = Experiment_VDM(config)
    g = torch.Generator().manual_seed(100 + exp.rank)
    state = exp.state
    snaps, info = [], {}
    for i in range(STEPS):
        batch = {"images": torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8).to(exp.device),
                 "labels": torch.zeros(4, dtype=torch.int32, device=exp.device),
                 "conditioning": torch.zeros(4, dtype=torch.uint8, device=exp.device)}
        state, m = exp.train_step(exp._train_rng, state, batch)
        torch.cuda.synchronize()
        snaps.append(tuple(t.clone() for t in (state.flat, state.ema, state.mu, state.nu, state.grad)) +
                     (float(m["scalars"]["train_bpd"]),))
    red = exp.reducer
    info = {"buckets": len(red.buckets), "graphed": exp._graphed is not None,
            "marked": list(red.capture["order"]) if red.capture else [], "ready_order": list(red.ready_order)}
    exp._graphed = None
    del exp
    torch.cuda.empty_cache()
    return snaps, info


def main():
    from mulan_amd import parallel
    rank, world, local = parallel.init_distributed()
    assert world == 2, world
    bad, summary = [], []
    for vdm_type in os.environ.get("MULAN_CHECK_MODELS", "mulan_epsilon,mulan_velocity").split(","):
        check(vdm_type, bad, summary)
    ok = torch.tensor([0.0 if bad else 1.0], device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"OVERLAP_REPLAY_CHECK {'ok' if float(ok[0]) == 1.0 else 'FAILED'} backend={dist.get_backend()} " +
              " | ".join(summary), flush=True)
    for b in bad:
        print(f"[rank {rank}] {b}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if float(ok[0]) == 1.0 else 1)


def check(vdm_type, bad, summary):
    eager, ie = run(False, vdm_type)
    replay, ir = run(True, vdm_type)
    names = ("params", "ema", "mu", "nu", "reduced gradient")
    for s in range(STEPS):
        for n, a, b in zip(names, eager[s][:5], replay[s][:5]):
            if not torch.equal(a, b):
                bad.append(f"{vdm_type} step {s} {n}: max |diff| {float((a - b).abs().max()):.3e}")
        if eager[s][5] != replay[s][5]:
            bad.append(f"step {s} train_bpd {eager[s][5]} vs {replay[s][5]}")
    # the replayed run really took the overlapped path: the step was captured, more than one bucket was marked with an
    # event node during the capture, and the collectives of the last step were issued in that order behind the nodes
    if not ir["graphed"] or ie["graphed"]:
        bad.append(f"graph use: eager run {ie['graphed']}, replay run {ir['graphed']}")
    if len(ir["marked"]) < 2 or ir["ready_order"][:len(ir["marked"])] != ir["marked"]:
        bad.append(f"buckets marked in the capture {ir['marked']} vs issued {ir['ready_order']} of {ir['buckets']}")
    if sorted(ir["ready_order"]) != list(range(ir["buckets"])):
        bad.append(f"not every bucket was reduced: {ir['ready_order']}")
    summary.append(f"{vdm_type}: buckets={ir['buckets']} marked={ir['marked']} issued={ir['ready_order']} "
                   f"eager_order={ie['ready_order']} bpd={[round(s[5], 5) for s in replay]}")


if __name__ == "__main__":
    main()
