"""ldm.model_vdm surface (ldm/model_vdm.py of the reference): VDMConfig, VDMOutput, VDM (scalar-schedule VDM), and the
module handles ScoreUNet / EncDec with the reference's call signatures (`Module(config).apply(params, ...)`), plus the
functional forms the build itself uses."""
from mulan_amd.model import (VDMConfig, VDMOutput, PlainVDM as VDM, ScoreUNet, EncDec,  # noqa: F401
                             score_unet, attn_block, resnet_block)
