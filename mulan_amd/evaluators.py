"""Variational-bound BPD evaluators (mirror of ldm/notebook_utils.py:28-39,157-191).

dense : per test image, a batch of `n_timesteps` copies -> antithetic t covers [0,1) with spacing
        1/n_timesteps, i.e. a Riemann estimate of the diffusion-loss integral; the SAME rng key for every
        image (PRNGKey(0), notebook_utils.py:178).  Images are independent, so under torchrun the test set
        is sharded by index across ranks and (sum bpd, count) is all-reduced once at the end.
sparse: batches of batch_size_eval distinct images, same fixed key (notebook_utils.py:157-173).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import checkpoint as ckpt_lib
from . import data as dataset
from . import parallel
from .experiment import Experiment_VDM
from .rng import PRNGKey


class Experiment_Colab(Experiment_VDM):
    """Experiment_VDM + EMA parameters restored from `<dir>/ckpt-<N>` (ldm/notebook_utils.py:28-39)."""

    def __init__(self, config, checkpoint_dir, checkpoint_num=None):
        super().__init__(config)
        if checkpoint_num is None:
            sd = ckpt_lib.restore_dict(checkpoint_dir)
        else:
            sd = ckpt_lib.restore_dict(os.path.join(checkpoint_dir, f'ckpt-{checkpoint_num}'))
        self.state.load_state_dict({"ema_params": sd["ema_params"]}, strict=True)
        self.orig_params = self.state.ema_params
        self.params = self.orig_params


def _reduce_mean(total, count, device):
    if parallel.world_size() > 1:
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total, count = float(t[0]), int(t[1])
    return total / max(count, 1), count


def eval_bpd_dense_sampling(experiment, config, n_timesteps=128, max_images=0):
    loader = dataset.create_one_time_eval_dataset(config, 1, experiment.device, experiment.rank, experiment.world)
    rng = PRNGKey(0)
    total, count = 0.0, 0
    for eval_step, batch in enumerate(loader):
        if max_images and eval_step * experiment.world >= max_images:
            break
        images = batch['images'].reshape(1, 32, 32, 3).expand(n_timesteps, 32, 32, 3).contiguous()
        tiled = {'images': images, 'labels': batch['labels'].expand(n_timesteps),
                 'conditioning': torch.zeros(n_timesteps, dtype=torch.uint8, device=experiment.device)}
        with torch.no_grad():
            bpd, _ = experiment.loss_fn(experiment.orig_params, tiled, eval_step, rng=rng, is_train=False)
        total += float(bpd)
        count += 1
        if count % 100 == 0 and experiment.rank == 0:
            print(f'eval_step {count} cum_avg_bpd {total / count} ')
    mean, n = _reduce_mean(total, count, experiment.device)
    if experiment.rank == 0:
        print('Num eval steps:', n)
    return mean


def eval_bpd_sparse_sampling(experiment, config, max_images=0):
    batch_size = config.training.batch_size_eval
    loader = dataset.create_one_time_eval_dataset(config, batch_size, experiment.device, experiment.rank,
                                                  experiment.world)
    rng = PRNGKey(0)
    total, count = 0.0, 0
    for eval_step, batch in enumerate(loader):
        if max_images and eval_step * batch_size * experiment.world >= max_images:
            break
        with torch.no_grad():
            bpd, _ = experiment.loss_fn(experiment.orig_params, batch, eval_step, rng=rng, is_train=False)
        total += float(bpd)
        count += 1
        if count % 100 == 0 and experiment.rank == 0:
            print(f'eval_step {count} cum_avg_bpd {total / count} ')
    mean, n = _reduce_mean(total, count, experiment.device)
    if experiment.rank == 0:
        print('Num eval steps:', n)
    return mean


def eval_bpd_ode(*args, **kwargs):
    raise NotImplementedError("exact-likelihood ODE evaluator (ldm/notebook_utils.py:264-373,446-531) is the next tier "
                              "(SURVEY 8f rank 2); use --bpd_eval_method=dense or sparse")
