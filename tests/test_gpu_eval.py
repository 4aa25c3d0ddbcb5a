"""Variational-bound evaluators (ldm/notebook_utils.py:157-191) against the float64 oracle, and BASELINE config #5
(eval_bpd --bpd_eval_method=dense, ImageNet-32 configuration, T = 1000 copies per image) at full size through the
properties the estimator has: the antithetic time grid, finiteness, invariance under sharding the images over ranks."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.oracle_dev import run_oracle
from oracle import torch_ref as tr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _imagenet32_experiment(tmp_path, images, n_layer, fwd_layers, batch_size_eval=2, vfe=True):
    """Experiment_VDM on configs/imagenet32.py as the README evaluates it (vdm_type=mulan_velocity,
    velocity_from_epsilon=True: /root/reference README.md:49), data = an npz of `images`"""
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    np.savez(tmp_path / "val.npz", images=images)
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "imagenet32.py"))
    config.vdm_type = "mulan_velocity"
    config.model.velocity_from_epsilon = vfe
    config.model.sm_n_layer = n_layer
    config.model.forward_n_layer = fwd_layers
    config.data.dataset = f"npz:{tmp_path / 'val.npz'}"
    config.training.batch_size_train = 2
    config.training.batch_size_eval = batch_size_eval
    config.training.substeps = 1
    exp = Experiment_VDM(config)
    exp.orig_params = exp.state.ema_params
    return exp, config


def _load_oracle_params(exp, ocfg, seed):
    """random (non-zero-init) parameters: the same tree in float64 for the oracle and on the device"""
    from mulan_amd import model as M
    ref_params = tr.init_params(ocfg, seed=seed, dtype=torch.float64)
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), exp.state.ema_params)
    return ref_params


def _eval_noise(exp, B):
    """the noise the evaluators draw: loss_fn splits PRNGKey(0) and hands the 'sample' key to the model
    (ldm/experiment_vdm.py:48-52, ldm/notebook_utils.py:160,178)"""
    from mulan_amd.rng import PRNGKey
    _, sample_rng = PRNGKey(0).split()
    return exp.model._noise({'sample': sample_rng}, None, B, exp.device, True)


def _oracle_bpd(ref_params, ocfg, x_u8, noise):
    B = x_u8.shape[0]
    with torch.no_grad():
        out = run_oracle(lambda P, *a: tr.mulan_forward(P, ocfg, *a), ref_params, torch.as_tensor(x_u8), float(noise["t0"]),
                         noise["gamma_raw"].double().cpu(), noise["eps_0"].double().cpu().view(B, 32, 32, 3),
                         noise["eps"].double().cpu().view(B, 32, 32, 3))
    return float(out["bpd"])


def test_dense_and_sparse_eval_match_oracle(tmp_path):
    """eval_bpd_dense_sampling / eval_bpd_sparse_sampling on the HIP path == the oracle's mean of per-image (per-batch)
    BPDs under the evaluators' own noise, E = 256 (imagenet32 configuration, velocity_from_epsilon).  Bar: the
    north-star's +-0.005 bits/dim, absolute."""
    from mulan_amd import evaluators as ev
    rng = np.random.default_rng(5)
    images = rng.integers(0, 256, (4, 32, 32, 3)).astype(np.uint8)
    exp, config = _imagenet32_experiment(tmp_path, images, n_layer=1, fwd_layers=1, batch_size_eval=2)
    ocfg = dict(vdm_type="mulan_velocity", n_embd=256, n_layer=1, forward_n_layer=1, latent_k=15, unet_type="vdm",
                velocity_from_epsilon=True)
    ref_params = _load_oracle_params(exp, ocfg, seed=11)

    T = 16
    got = ev.eval_bpd_dense_sampling(exp, config, n_timesteps=T, max_images=3)
    noise = _eval_noise(exp, T)
    want = np.mean([_oracle_bpd(ref_params, ocfg, np.repeat(images[i:i + 1], T, axis=0), noise) for i in range(3)])
    assert abs(got - want) < 0.005, (got, want)

    # the dense evaluator runs the encoder U-Net on one copy of the image and broadcasts its logits (same_image): the
    # same bits as running it on all T copies
    from mulan_amd.rng import PRNGKey
    tiled = {"images": torch.tensor(np.repeat(images[1:2], T, axis=0)).cuda(), "labels": torch.zeros(T, dtype=torch.int32).cuda(),
             "conditioning": torch.zeros(T, dtype=torch.uint8).cuda()}
    with torch.no_grad():
        b_all, _ = exp.loss_fn(exp.orig_params, tiled, 0, rng=PRNGKey(0), is_train=False)
        b_one, _ = exp.loss_fn(exp.orig_params, tiled, 0, rng=PRNGKey(0), is_train=False, same_image=True)
    assert float(b_all) == float(b_one)

    got_s = ev.eval_bpd_sparse_sampling(exp, config)              # two batches of two distinct images
    noise2 = _eval_noise(exp, 2)
    want_s = np.mean([_oracle_bpd(ref_params, ocfg, images[k:k + 2], noise2) for k in (0, 2)])
    assert abs(got_s - want_s) < 0.005, (got_s, want_s)
    assert abs(got - got_s) > 1e-6                                # two different estimators of the same bound


def test_dense_eval_full_size_config5(tmp_path):
    """BASELINE config #5 at its real size on one GPU: ImageNet-32 configuration (E = 256, 32 + 2 + 33 blocks),
    T = 1000 copies per image.  Properties: the 1000 times are the antithetic grid (t0 + i / 1000) mod 1 (spacing
    1/1000 over [0, 1)); the bound is finite and sane; splitting the test images over 2 ranks by index and combining
    (sum, count) gives the unsharded result; the same key for every image makes the estimate a function of the image
    only."""
    from mulan_amd import evaluators as ev
    rng = np.random.default_rng(6)
    images = rng.integers(0, 256, (3, 32, 32, 3)).astype(np.uint8)
    images[2] = images[0]                                          # the same image twice: the same bits per dim
    exp, config = _imagenet32_experiment(tmp_path, images, n_layer=32, fwd_layers=4, batch_size_eval=2)
    assert config.model.sm_n_embd == 256 and config.model.sm_n_layer == 32
    T = 1000
    noise = _eval_noise(exp, T)
    t = exp.model._times(noise, T, exp.device).double().cpu().numpy()
    ts = np.sort(t)
    assert ts.min() >= 0.0 and ts.max() < 1.0
    assert np.allclose(np.diff(ts), 1.0 / T, atol=2e-6) and abs(ts[0] - (float(noise["t0"]) % (1.0 / T))) < 2e-6

    full_total, full_count = ev._dense_partial(exp, config, T, 0, 1)
    assert full_count == 3 and np.isfinite(full_total)
    parts = [ev._dense_partial(exp, config, T, r, 2) for r in range(2)]
    assert [c for _, c in parts] == [2, 1]
    assert abs(sum(p for p, _ in parts) - full_total) < 1e-5 * abs(full_total)
    # image 0 and image 2 are identical and evaluated under the same key
    per_image = [ev._dense_partial(exp, config, T, r, 3)[0] for r in range(3)]
    assert abs(per_image[0] - per_image[2]) < 1e-6 * abs(per_image[0])
    mean = ev.eval_bpd_dense_sampling(exp, config, n_timesteps=T)
    assert abs(mean - full_total / 3) < 1e-6 * abs(mean)
    assert 3.0 < mean < 40.0                                       # random init on random images: far above 8 bits
