"""Ancestral sampler throughput at the flagship configuration (cifar10-conditioned: E = 128, sm_n_layer = 32):
reverse steps per second and images per second for a batch of B samples, T timed steps (the reference runs 1000).
    python tools/sampler_bench.py [--batch 64] [--steps 20]"""
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--T", type=int, default=1000)
    a = ap.parse_args()
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd.rng import PRNGKey
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = 'synthetic'
    config.training.batch_size_eval = a.batch
    exp = Experiment_VDM(config)
    st, model = exp.state, exp.model
    B = a.batch
    cond = torch.zeros(B, dtype=torch.uint8, device=exp.device)
    key = PRNGKey(0)
    packer = st.param_packer("ema")
    if packer is not None:
        packer.refresh()
    emb = model.deterministic_embedding(B, exp.device)
    coeffs = model.sample_coefficients(st.ema_params, emb)
    step = model.reverse_stepper(st.ema_params, B, exp.device, emb, cond, coeffs, a.T)   # MULAN_SAMPLER_GRAPH=0: eager
    z = key.normal((B, 3072), exp.device)
    with torch.no_grad():
        for i in range(3):
            z = step(i, z, key)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + a.steps):
            z = step(i, z, key)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    x = model.generate_x(st.ema_params, z, coeffs)
    torch.cuda.synchronize()
    assert x.shape == (B, 32, 32, 3) and bool(torch.isfinite(z).all())
    print(json.dumps({"metric": "sampler_reverse_steps_per_sec", "batch": B, "ms_per_step": dt * 1e3,
                      "steps_per_sec": 1.0 / dt, "images_per_sec_at_T1000": B / (dt * a.T),
                      "image_steps_per_sec": B / dt}))


if __name__ == "__main__":
    main()
