#!/usr/bin/env python3
"""Soak run: N optimiser steps of the full-depth CIFAR configuration (32 + 2 + 33 blocks, dropout on, lr warm-up, AdamW +
EMA, HIP-graph replay) on a batch of smooth synthetic images; prints the training BPD every 25 steps and fails if it is
not finite or does not come down.  python tools/soak_train.py [--steps 300] [--batch 32] [--config cifar10-conditioned.py]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--config", default="cifar10-conditioned.py")
    ap.add_argument("--vdm-type", default=None)
    a = ap.parse_args()
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", a.config))
    if a.vdm_type:
        config.vdm_type = a.vdm_type
    config.data.dataset = "synthetic"
    config.training.batch_size_train = a.batch
    config.training.batch_size_eval = a.batch
    config.training.substeps = 1
    config.training.num_steps_lr_warmup = 50
    exp = Experiment_VDM(config)
    yy, xx = torch.meshgrid(torch.arange(32.0), torch.arange(32.0), indexing="ij")
    imgs = []
    for k in range(a.batch):
        r = 128 + 100 * torch.sin((xx * (1 + k % 3) + yy * (k % 5)) / 10.0 + k)
        g = 128 + 100 * torch.cos((yy * (1 + k % 4)) / 9.0 - k)
        b = 255 * torch.exp(-((xx - 16 - k % 7) ** 2 + (yy - 12) ** 2) / 80.0)
        imgs.append(torch.stack([r, g, b], dim=-1))
    images = torch.stack(imgs).clamp(0, 255).round().to(torch.uint8).cuda()
    sub = {"images": images, "labels": torch.zeros(a.batch, dtype=torch.int32).cuda(),
           "conditioning": torch.zeros(a.batch, dtype=torch.uint8).cuda()}
    hist = []
    t0 = time.time()
    for i in range(a.steps):
        exp.state, m = exp.train_step(exp._train_rng, exp.state, sub)
        if i % 25 == 0 or i == a.steps - 1:
            v = float(m["scalars"]["train_bpd"])
            hist.append(v)
            print(f"step {i:4d}  train_bpd {v:8.4f}  ({(time.time() - t0):5.1f} s, graph={exp._graphed is not None})", flush=True)
            assert np.isfinite(v), "non-finite loss"
    ev = float(exp.eval_step(exp._eval_rng, exp.state.ema_params, sub, 0)["scalars"]["eval_bpd"])
    print(f"eval_bpd on the EMA parameters: {ev:.4f}; first / last train_bpd: {hist[0]:.3f} / {hist[-1]:.3f}")
    assert hist[-1] < 0.7 * hist[0], (hist[0], hist[-1])
    print("SOAK ok")


if __name__ == "__main__":
    main()
