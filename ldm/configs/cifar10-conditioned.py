"""CIFAR-10 MuLAN (velocity) -- values of the reference's ldm/configs/cifar10-conditioned.py."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("_mulan_cfg_base", os.path.join(os.path.dirname(__file__), "_base.py"))
_base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_base)


def get_config():
    return _base.build(
        dataset='cifar10', vdm_type='mulan_velocity',
        model=dict(latent_k=15, trace_matching=False, sigma_type='no_blur', sigma_min=0.0, sigma_max=20.0,
                   sm_n_embd=128),
        training=dict(num_steps_train=10_000_000, batch_size_train=128, batch_size_eval=128))
