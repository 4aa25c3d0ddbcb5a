"""Exact-likelihood evaluator throughput at the flagship configuration (cifar10-conditioned: E = 128, 32-block U-Nets):
time per function evaluation of the probability-flow ODE (U-Net forward + input-gradient pass + drift / divergence
kernels) and per Dormand-Prince step incl. the controller's one scalar read-back.
    python tools/ode_bench.py [--batch 64] [--steps 3]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    from mulan_amd import ops
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd.ode import solve_fixed
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = 'synthetic'
    config.training.batch_size_eval = a.batch
    exp = Experiment_VDM(config)
    st, model = exp.state, exp.model
    B = a.batch
    packer = st.param_packer("ema")
    if packer is not None:
        packer.refresh()
    img = torch.randint(0, 256, (B, 32, 32, 3), dtype=torch.uint8, device=exp.device)
    u = ops.noise((B, 3072), 1, 0, exp.device, "truncated_normal")
    y, rq = ops.dequantize(img.view(B, 3072), u, False, 0.0013)
    ctx = model.ode_context(st.ema_params, rq)
    probe = ops.noise((B, 3072), 2, 0, exp.device, "rademacher")
    n_x = B * 3072

    from mulan_amd.model import ode_function
    fe = ode_function(model, st.ema_params, ctx, B, exp.device, True)      # replayed HIP graph (MULAN_ODE_GRAPH=0: eager)

    def f(t, y32, out):
        fe(t, y32[:n_x].view(B, 3072), probe, out[:n_x].view(B, 3072), out[n_x:])

    y0 = torch.cat([y.reshape(-1).double(), torch.zeros(B, device=exp.device, dtype=torch.float64)])
    solve_fixed(f, y0, [0.0, 0.01])                         # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    grid = [0.1 * i for i in range(a.steps + 1)]
    sol = solve_fixed(f, y0, grid)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(sol.y).all())
    per_nfe = dt / sol.nfev
    print(json.dumps({"metric": "ode_function_evaluations_per_sec", "batch": B, "nfev": sol.nfev,
                      "ms_per_nfe": per_nfe * 1e3, "image_nfe_per_sec": B / per_nfe,
                      "sec_per_image_at_nfe_300_is_20": 300 * 20 * per_nfe / B}))


if __name__ == "__main__":
    main()
