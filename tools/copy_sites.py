#!/usr/bin/env python3
"""dev: where do the device-to-device copies and tiny torch kernels of one eager train step come from?
(torch.profiler with stacks; prints the Python call sites of Memcpy DtoD and of aten::add_/copy_/cat kernels)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

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
batch = {"images": torch.randint(0, 256, (4, 32, 32, 3), dtype=torch.uint8).cuda(),
         "labels": torch.zeros(4, dtype=torch.int32).cuda(), "conditioning": torch.zeros(4, dtype=torch.uint8).cuda()}
for _ in range(2):
    exp.train_step(exp._train_rng, exp.state, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    exp.train_step(exp._train_rng, exp.state, batch)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add_", "aten::add", "aten::cat", "aten::clone", "aten::contiguous", "aten::zero_",
                   "aten::fill_", "aten::mul", "aten::sum", "aten::mean", "aten::div", "aten::to", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "mulan_amd" in s or "torch/autograd" in s]
        sites[(ev.name, st[0] if st else "?")] += 1
for (name, site), n in sites.most_common(40):
    print(f"{n:5d}  {name:18s} {site}")
