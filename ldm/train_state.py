"""ldm.train_state -> mulan_amd.train_state (TrainState)."""
from mulan_amd.train_state import TrainState  # noqa: F401
