"""ldm.model_mulan_velocity surface: VDM(config) = MuLAN with the velocity parameterisation."""
from mulan_amd.model import MulanVDM as _MulanVDM


def VDM(config):
    return _MulanVDM(config, "velocity")
