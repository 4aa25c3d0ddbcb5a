"""ldm.model_vdm surface: VDMConfig, VDMOutput, VDM (scalar-schedule VDM), ScoreUNet application."""
from mulan_amd.model import VDMConfig, VDMOutput, PlainVDM as VDM, score_unet, attn_block, resnet_block  # noqa: F401
