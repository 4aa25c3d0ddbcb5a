#!/usr/bin/env python3
"""dev: every launch of one eager train step (flagship config, B = 4: launch counts do not depend on the batch) by
Python call site -- the library's entry points through ops.call (site = first frame outside lib.py), torch's own small
kernels and device-to-device copies through torch.profiler stacks.  `--only colsum,copy` filters by substring."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

from mulan_amd import ops, lib
from mulan_amd.config import load_config_file
from mulan_amd.experiment import Experiment_VDM

only = sys.argv[sys.argv.index("--only") + 1].split(",") if "--only" in sys.argv else None
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

calls = collections.Counter()
real = lib.call


def counting(name, *a):
    fr = [f for f in traceback.extract_stack()[:-1] if "mulan_amd" in f.filename and not f.filename.endswith("lib.py")]
    site = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:]))
    calls[(name, site)] += 1
    return real(name, *a)


for mod in (ops, lib):
    if getattr(mod, "call", None) is real:
        mod.call = counting
import mulan_amd.train_state as ts, mulan_amd.model as mm, mulan_amd.experiment as ee
for mod in (ts, mm, ee):
    if getattr(mod, "call", None) is real:
        mod.call = counting
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    exp.train_step(exp._train_rng, exp.state, batch)
    torch.cuda.synchronize()
print(f"library launches: {sum(calls.values())}")
by_name = collections.Counter()
for (n, s), c in calls.items():
    by_name[n] += c
for n, c in by_name.most_common():
    if only and not any(o in n for o in only):
        continue
    print(f"{c:5d}  {n}")
    for (n2, s), c2 in sorted(calls.items(), key=lambda kv: -kv[1]):
        if n2 == n and (only or c2 >= 8):
            print(f"        {c2:5d}  {s}")
sites = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided",
                                                         "aten::empty_strided", "aten::reshape", "aten::view_as", "aten::select",
                                                         "aten::slice", "aten::expand", "aten::alias", "aten::detach",
                                                         "aten::unsqueeze", "aten::squeeze", "aten::transpose", "aten::t",
                                                         "aten::permute", "aten::_unsafe_view", "aten::item", "aten::narrow",
                                                         "aten::_local_scalar_dense", "aten::unflatten", "aten::flatten",
                                                         "aten::resize_", "aten::is_nonzero", "aten::result_type", "aten::lift_fresh"):
        st = [s for s in (ev.stack or []) if "mulan_amd" in s]
        sites[(ev.name, st[0] if st else "?")] += 1
print("torch ops (may launch kernels):")
for (name, site), n in sites.most_common(60):
    if only and not any(o in name for o in only):
        continue
    print(f"{n:5d}  {name:18s} {site}")
