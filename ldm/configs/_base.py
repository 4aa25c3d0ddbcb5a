"""Shared builder for the shipped configurations: same keys and values as the reference's
ldm/configs/cifar10-conditioned.py and ldm/configs/imagenet32.py, expressed as one
table of defaults plus per-dataset deltas."""
import ml_collections

_MODEL = dict(
    unet_type='vdm', vocab_size=256, sample_softmax=False, antithetic_time_sampling=True,
    with_fourier_features=True, with_attention=False, condition='input', encoder='unet', forward_n_layer=4,
    latent_size=50, model_time=False, epsilon=0.0, monotone_layer='dense_monotone', gamma_type='poly_fixedend',
    latent_type='topk', z_conditioning=True, importance_sampling=False, topk_noise_type='gamma', sigma_prior=1.0,
    reparam_type='true', gamma_min=-13.3, gamma_max=5., velocity_from_epsilon=False, sm_n_timesteps=0,
    sm_n_layer=32, sm_pdrop=0.1)
_TRAINING = dict(seed=1, substeps=1000, num_steps_lr_warmup=100, num_steps_eval=100, steps_per_logging=1000,
                 steps_per_eval=10_000, steps_per_save=10_000, profile=False)
_OPTIMIZER = dict(name='adamw', args=dict(b1=0.9, b2=0.99, eps=1e-8, weight_decay=0.01), learning_rate=2e-4,
                  lr_decay=False, ema_rate=0.9999)


def build(*, dataset, vdm_type, model, training, extra=None):
    cfg = ml_collections.ConfigDict()
    cfg.exp_name = 'exp_vdm'
    cfg.model_type = 'model_vdm'
    cfg.ckpt_restore_dir = 'None'
    cfg.data = ml_collections.ConfigDict(dict(dataset=dataset, ignore_cache=False))
    cfg.vdm_type = vdm_type
    cfg.model = ml_collections.ConfigDict({**_MODEL, **model})
    cfg.training = ml_collections.ConfigDict({**_TRAINING, **training})
    cfg.optimizer = ml_collections.ConfigDict(_OPTIMIZER)
    for k, v in (extra or {}).items():
        cfg[k] = v
    return cfg
