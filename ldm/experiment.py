"""ldm.experiment -> mulan_amd.experiment (Experiment, restore_partial)."""
from mulan_amd.experiment import Experiment, restore_partial  # noqa: F401
