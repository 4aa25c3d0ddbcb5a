#!/usr/bin/env python3
"""dev: which parameter leaves still reach the flat gradient buffer through a copy (TrainState.collect_grads / the
reducer hook: one __amd_rocclr_copyBuffer launch each inside the replayed step) instead of being written there by the
kernel that produces them (a gradient "sink")?  Runs one eager train step of the bench workload's model at batch 4."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mulan_amd import ops
from mulan_amd.config import load_config_file
from mulan_amd.experiment import Experiment_VDM

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
config.data.dataset = "synthetic"
config.training.batch_size_train = 4
config.training.batch_size_eval = 4
config.training.substeps = 1
config.training.hip_graph = False
exp = Experiment_VDM(config)
st = exp.state
batch = {"images": torch.randint(0, 256, (4, 32, 32, 3), dtype=torch.uint8).cuda(),
         "labels": torch.zeros(4, dtype=torch.int32).cuda(), "conditioning": torch.zeros(4, dtype=torch.uint8).cuda()}
exp.train_step(exp._train_rng, st, batch)
st.zero_grad()
packer = st.param_packer()
if packer is not None:
    packer.refresh()
rng = exp._train_rng.fold_in(0).fold_in(st.step)
bpd, _ = exp.loss_fn(st.params, batch, step=st.step, rng=rng, is_train=True)
with ops.weight_gradient_stream():
    bpd.backward()
torch.cuda.synchronize()
copied = collections.Counter()
names = []
for (path, off, shape), leaf in zip(st.layout, st._leaves):
    g = leaf.grad
    if g is not None and g.data_ptr() != leaf._gview.data_ptr():
        kind = path[-2] + "/" + path[-1] if len(path) >= 2 else "/".join(path)
        copied[(path[0], kind, tuple(shape))] += 1
        names.append("/".join(path))
print("leaves:", len(st._leaves), " copied by collect_grads:", sum(copied.values()))
for (top, kind, shape), n in copied.most_common():
    print(f"{n:4d}  {top:14s} {kind:28s} {shape}")
