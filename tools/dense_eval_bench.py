"""BASELINE config #5 per GPU: one dense-eval item = n_timesteps copies of one test image through loss_fn(is_train=False)
(ldm/notebook_utils.py:176-191), ImageNet-32 config (E = 256), timed over a few images.
    python tools/dense_eval_bench.py [--n-timesteps 1000] [--images 3]"""
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
    ap.add_argument("--n-timesteps", type=int, default=1000)
    ap.add_argument("--images", type=int, default=3)
    ap.add_argument("--config", default=os.path.join(ROOT, "ldm", "configs", "imagenet32.py"))
    a = ap.parse_args()
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd.rng import PRNGKey
    config = load_config_file(a.config)
    config.data.dataset = 'synthetic'
    config.vdm_type = 'mulan_velocity'
    config.model.velocity_from_epsilon = True
    config.training.batch_size_train = 8
    config.training.batch_size_eval = 8
    exp = Experiment_VDM(config)
    T = a.n_timesteps
    rng = PRNGKey(0)
    times, bpds = [], []
    for i in range(a.images + 1):
        img = torch.randint(0, 256, (1, 32, 32, 3), dtype=torch.uint8, device=exp.device)
        tiled = {'images': img.expand(T, 32, 32, 3).contiguous(), 'labels': torch.zeros(T, dtype=torch.int32, device=exp.device),
                 'conditioning': torch.zeros(T, dtype=torch.uint8, device=exp.device)}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            bpd, _ = exp.loss_fn(exp.state.ema_params, tiled, i, rng=rng, is_train=False, same_image=True)   # as the evaluator
        bpds.append(float(bpd))
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / a.images
    print(json.dumps({"metric": "dense_eval_images_per_sec", "n_timesteps": T, "sec_per_image": dt, "images_per_sec": 1.0 / dt,
                      "forward_images_per_sec": T / dt, "bpd_random_init": bpds[-1]}))


if __name__ == "__main__":
    main()
