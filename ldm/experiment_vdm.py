"""ldm.experiment_vdm -> mulan_amd.experiment (Experiment_VDM)."""
from mulan_amd.experiment import Experiment_VDM  # noqa: F401
