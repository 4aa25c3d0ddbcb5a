"""ldm.model_mulan_epsilon surface: VDM(config) = MuLAN with the epsilon parameterisation; UnetEncoder and the
polynomial schedule are exposed as their functional forms."""
from mulan_amd.model import MulanVDM as _MulanVDM, unet_encoder, poly_coefficients  # noqa: F401


def VDM(config):
    return _MulanVDM(config, "epsilon")
