"""ldm.ldm_unet surface: the per-pixel-FiLM U-Net is score_unet(...) with config.unet_type == 'ldm'."""
from mulan_amd.model import score_unet as UNet, resnet_block as ResnetBlock  # noqa: F401
