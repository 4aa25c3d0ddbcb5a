#!/usr/bin/env python3
"""Launched by tests/test_gpu_rccl.py under torch.distributed.run with one rank per GPU (backend nccl = RCCL over xGMI):
the bucketed side-stream all-reduce of the flat gradient buffer (mulan_amd.parallel.GradReducer, 1/world applied as
the optimizer's grad_scale) must give the gradient one rank computes on the whole batch (lax.pmean of
ldm/experiment.py:341), and the scalar metrics their mean (:347).  Prints one line `RCCL_GRAD_CHECK ok ...` on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mulan_amd import model as M, parallel
    from mulan_amd.rng import PRNGKey
    from mulan_amd.train_state import TrainState
    from tests.test_gpu_model import make_cfg
    import dataclasses
    rank, world, local = parallel.init_distributed(backend="nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg, _ = make_cfg("mulan_velocity", "vdm", False)
    cfg = dataclasses.replace(cfg, antithetic_time_sampling=False)      # explicit per-sample t: the shards see their own
    vdm = M.make_vdm("mulan_velocity", cfg)
    tmpl = vdm.init(PRNGKey(3))
    gen = torch.Generator().manual_seed(5)
    for _, leaf in M.tree_leaves(tmpl):                                 # no zero-initialised layers
        leaf.copy_(torch.randn(leaf.shape, generator=gen) * (0.05 if leaf.dim() > 1 else 0.2))
    Bl = 2
    B = Bl * world
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8))
    noise = dict(t=torch.tensor(rng.random(B), dtype=torch.float32),
                 gamma_raw=torch.tensor(rng.gamma(1.0 / 15, size=(10, B, 50)), dtype=torch.float32),
                 eps_0=torch.tensor(rng.standard_normal((B, 3072)), dtype=torch.float32),
                 eps=torch.tensor(rng.standard_normal((B, 3072)), dtype=torch.float32))

    def grads(sel, reducer_on):
        st = TrainState.create(apply_fn=vdm.apply, variables={"params": tmpl}, device=dev)
        red = parallel.GradReducer(st.grad, st.reducer_leaves(), bucket_bytes=16 << 20)
        red.enabled = red.enabled and reducer_on
        st.zero_grad()
        red.prepare()
        n = {k: (v[:, sel] if k == "gamma_raw" else v[sel]).to(dev) for k, v in noise.items()}
        out = vdm.apply(st.params, x[sel].to(dev), None, None, step=0, rngs=None, deterministic=True, noise=n)
        bpd = (out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) / (3072 * np.log(2.0))
        bpd.backward()
        st.collect_grads()
        red.finish()
        torch.cuda.synchronize()
        return st.grad.clone(), float(bpd), red

    mine = slice(rank * Bl, (rank + 1) * Bl)
    g_red, bpd_local, red = grads(mine, True)
    g_red /= world
    g_all, bpd_all, _ = grads(slice(0, B), False)
    err = float((g_red - g_all).abs().max() / g_all.abs().max())
    m = parallel.allreduce_mean_scalars({"bpd": torch.tensor(bpd_local, device=dev)}, dev)
    ok = err < 2e-5 and abs(float(m["bpd"]) - bpd_all) < 1e-4 * abs(bpd_all) and len(red.buckets) > 1
    t = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"RCCL_GRAD_CHECK {'ok' if float(t[0]) == 1.0 else 'FAILED'} world={world} rel_err={err:.2e} "
              f"buckets={len(red.buckets)} ready_order={red.ready_order}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if float(t[0]) == 1.0 else 1)


if __name__ == "__main__":
    main()
