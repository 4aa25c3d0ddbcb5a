#!/usr/bin/env python3
"""dev: one train step (eager, then replayed) with the slab reductions folded into the next weight gradient and with the
separate reduction launches, from the same state: the gradient buffers must be bit-identical."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402
from mulan_amd.config import load_config_file  # noqa: E402
from mulan_amd.experiment import Experiment_VDM  # noqa: E402


def run(fold, graph, depth, B):
    ops.FOLD_SLAB_REDUCE = fold
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.model.sm_n_layer = depth
    config.model.forward_n_layer = 1
    config.data.dataset = "synthetic"
    config.training.batch_size_train = B
    config.training.batch_size_eval = B
    config.training.substeps = 1
    config.training.hip_graph = graph
    exp = Experiment_VDM(config)
    with torch.no_grad():       # un-zero the zero-initialised tensors: every gradient is live from step 0
        exp.state.flat.add_(0.01 * torch.randn(exp.state.flat.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(0)))
    g = torch.Generator().manual_seed(3)
    grads = []
    for i in range(4):
        batch = {"images": torch.randint(0, 256, (B, 32, 32, 3), generator=g, dtype=torch.uint8).cuda(),
                 "labels": torch.zeros(B, dtype=torch.int32).cuda(), "conditioning": torch.zeros(B, dtype=torch.uint8).cuda()}
        _, m = exp.train_step(exp._train_rng, exp.state, batch)
        torch.cuda.synchronize()
        grads.append(exp.state.grad.clone())
    return grads, exp.state.flat.clone(), float(m["scalars"]["train_bpd"]), exp


for graph in (False, True):
    for depth, B in ((2, 8),):
        ga, pa, ba, ea = run(False, graph, depth, B)
        gb, pb, bb, eb = run(True, graph, depth, B)
        for i, (a, b) in enumerate(zip(ga, gb)):
            d = (a - b).abs()
            bad = int((a != b).sum())
            print(f"graph={graph} depth={depth} B={B} step {i}: gradients differ in {bad} of {a.numel()} elements, max |d| {float(d.max()):.3e}")
            if bad and i == 0:
                for p, off, shape in ea.state.layout:
                    n = int(torch.tensor(shape).prod())
                    da = (a[off:off + n] != b[off:off + n])
                    if bool(da.any()):
                        print(f"     {'/'.join(p):50s} {tuple(shape)} differing {int(da.sum())}/{n} max|d| {float((a[off:off+n]-b[off:off+n]).abs().max()):.2e} scale {float(a[off:off+n].abs().max()):.2e}")
        print(f"   params equal: {torch.equal(pa, pb)}; bpd {ba} {bb}")
