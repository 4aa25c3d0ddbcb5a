"""dev: where the device-to-device copies of one train step come from (aten::copy_ call sites)"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulan_amd.config import load_config_file
from mulan_amd.experiment import Experiment_VDM
from torch.utils._python_dispatch import TorchDispatchMode

config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
config.data.dataset = 'synthetic'
config.training.substeps = 1
exp = Experiment_VDM(config)
batch = next(exp.train_iter)
sub = {k: v[0] for k, v in batch.items()}
for _ in range(2):
    exp.state, _ = exp.train_step(exp._train_rng.fold_in(0), exp.state, sub)
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "contiguous", "cat", "add", "mul", "zero_", "fill_", "maximum")):
            fr = [f for f in traceback.extract_stack()[:-1] if "mulan_amd" in f.filename][-3:]
            shp = tuple(args[0].shape) if args and hasattr(args[0], "shape") else ()
            sites[(name, shp, " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    exp.state, _ = exp.train_step(exp._train_rng.fold_in(0), exp.state, sub)
torch.cuda.synchronize()
for (name, shp, where), n in sites.most_common(24):
    print(n, name, shp, where)
