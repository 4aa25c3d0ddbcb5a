"""ImageNet-32 MuLAN (epsilon) -- values of the reference's ldm/configs/imagenet32.py."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("_mulan_cfg_base", os.path.join(os.path.dirname(__file__), "_base.py"))
_base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_base)


def get_config():
    return _base.build(
        dataset='imagenet32', vdm_type='mulan_epsilon',
        model=dict(sm_n_embd=256),
        training=dict(num_steps_train=2_000_000, batch_size_train=512, batch_size_eval=512),
        extra=dict(lr_gamma_network_scale=1.0))
