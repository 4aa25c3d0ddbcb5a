"""ldm.ldm_unet surface (ldm/ldm_unet.py of the reference): UNet(config).apply(params, z, g_t [B,32,32,3],
conditioning, deterministic=True) = the per-pixel-FiLM denoiser; ResnetBlock in its functional form."""
from mulan_amd.model import UNet, resnet_block as ResnetBlock  # noqa: F401
