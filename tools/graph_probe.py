#!/usr/bin/env python3
"""Dev probe: which part of the train step can be captured into a HIP graph on this ROCm build?  Each stage runs in a
child process (a failed capture can take the process down).  python tools/graph_probe.py [stage]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGES = ["fwd_nograd", "fwd_bwd", "fwd_bwd_opt"]


def child(stage):
    import numpy as np
    import torch
    from mulan_amd import model as M, ops
    from mulan_amd.rng import PRNGKey, Key
    from mulan_amd.train_state import TrainState
    from tests.test_gpu_model import make_cfg
    ops.lib.load()
    cfg, _ = make_cfg("mulan_velocity", "vdm", False, n_layer=1)
    vdm = M.make_vdm("mulan_velocity", cfg)
    st = TrainState.create(apply_fn=vdm.apply, variables={"params": vdm.init(PRNGKey(0))}, device=torch.device("cuda"))
    B = 4
    x = torch.randint(0, 256, (B, 32, 32, 3), dtype=torch.uint8).cuda()
    noise = dict(t0=torch.tensor(0.3).cuda(), gamma_raw=torch.rand(10, B, 50).cuda(), eps_0=torch.randn(B, 3072).cuda(),
                 eps=torch.randn(B, 3072).cuda())
    seeds = torch.tensor([5, 7], dtype=torch.int64).cuda()
    rngs = {"dropout_pair": (Key(0, dev=seeds[0:1]), Key(0, dev=seeds[1:2]))}
    dyn = torch.tensor([1e-4, 0.1, 0.01, 0.0]).cuda()

    def body():
        if stage == "gn_only":
            h = torch.randn(B, 1024, 128).cuda()
            return ops.group_norm(h, None, torch.ones(128).cuda(), torch.zeros(128).cuda(), keep=0.9, seed=seeds[0:1], offset=0).sum()
        if stage == "conv_only":
            h = torch.randn(B, 1024, 128).cuda()
            return ops.conv3x3(h, torch.randn(3, 3, 128, 128).cuda() * 0.05).sum()
        if stage == "fwd_nograd":
            with torch.no_grad():
                out = vdm.apply(st.params, x, None, None, step=0, rngs=rngs, deterministic=False, noise=noise)
            return out.loss_diff.mean()
        st.zero_grad()
        if stage == "fwd_bwd_small":
            h = torch.randn(B, 1024, 128).cuda().requires_grad_(True)
            p = st.params["score_model"]["down.block_0"]
            y = ops.conv3x3(ops.group_norm(h, None, p["GroupNorm_0"]["scale"], p["GroupNorm_0"]["bias"]), p["conv1"]["kernel"], p["conv1"]["bias"])
            y.sum().backward()
            return y.sum()
        out = vdm.apply(st.params, x, None, None, step=0, rngs=rngs, deterministic=False, noise=noise)
        bpd = (out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) / (3072 * np.log(2.0))
        bpd.backward()
        st.collect_grads()
        if stage == "fwd_bwd_opt":
            st.apply_gradients(lr=0.0, ema_rate=0.9999, dyn=dyn, count_step=False)
        return bpd.detach()

    r = body()                      # eager warm-up
    st.zero_grad()
    st.drop_graph_refs()
    torch.cuda.synchronize()
    print(stage, "eager", float(r), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        r = body()
    print(stage, "captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(stage, "replayed", float(r), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for stg in (sys.argv[1:] or STAGES):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", stg], capture_output=True, text=True)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip()][-4:]
            print(f"== {stg}: rc={r.returncode}", *tail, sep="\n   ", flush=True)
