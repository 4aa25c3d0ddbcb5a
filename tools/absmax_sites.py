"""dev: which tensors still need a separate maxima pass in one train step (call sites of ops.absmax_rows)"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulan_amd import ops
from mulan_amd.config import load_config_file
from mulan_amd.experiment import Experiment_VDM

config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
config.data.dataset = 'synthetic'
config.training.substeps = 1
exp = Experiment_VDM(config)
batch = next(exp.train_iter)
sub = {k: v[0] for k, v in batch.items()}
for _ in range(2):
    exp.state, _ = exp.train_step(exp._train_rng.fold_in(0), exp.state, sub)
sites = collections.Counter()
orig = ops.absmax_rows


def wrapped(x):
    fr = [f for f in traceback.extract_stack()[:-1] if "mulan_amd" in f.filename][-4:]
    sites[(tuple(x.shape), " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(fr)))] += 1
    return orig(x)


ops.absmax_rows = wrapped
exp.state, _ = exp.train_step(exp._train_rng.fold_in(0), exp.state, sub)
torch.cuda.synchronize()
for (shape, where), n in sites.most_common(20):
    print(n, shape, where)
