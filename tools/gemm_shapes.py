"""dev: every mulan_gemm call of one train step (CIFAR MuLAN config) with its shape and its time (event pair per call)"""
import collections
import os
import sys

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
log = []
orig = ops.gemm_raw


def wrapped(A, Bm, M, N, K, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig(A, Bm, M, N, K, **kw)
    e.record()
    log.append(((M, N, K, bool(kw.get("transA")), bool(kw.get("transB")), kw.get("batch", 1)), s, e))
    return out


ops.gemm_raw = wrapped
exp.state, _ = exp.train_step(exp._train_rng.fold_in(0), exp.state, sub)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, s, e in log:
    agg[key][0] += 1
    agg[key][1] += s.elapsed_time(e) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"{len(log)} GEMM calls, {tot / 1e3:.2f} ms (event pairs add ~5 us each)")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    M, N, K, ta, tb, b = key
    print(f"M={M:6d} N={N:5d} K={K:6d} tA={int(ta)} tB={int(tb)} batch={b:3d}: {n:3d} calls {us / n:8.1f} us each {us / 1e3:6.2f} ms  {2.0 * M * N * K * b * n / us * 1e-6:6.1f} TFLOP/s")
