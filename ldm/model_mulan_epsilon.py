"""ldm.model_mulan_epsilon surface: VDM(config) = MuLAN with the epsilon parameterisation; UnetEncoder and
NoiseSchedule_polynomial_fixedend as module handles with the reference's call signatures
(ldm/model_mulan_epsilon.py:105,602), their functional forms, and the registries the reference keeps (:157,676-680)."""
from mulan_amd.model import (MulanVDM as _MulanVDM, UnetEncoder, NoiseSchedule_polynomial_fixedend,  # noqa: F401
                             unet_encoder, poly_coefficients)

ENCODER_MODELS = {"unet": UnetEncoder}
GAMMA_NETWORKS = {"poly_fixedend": NoiseSchedule_polynomial_fixedend}


def VDM(config):
    return _MulanVDM(config, "epsilon")
