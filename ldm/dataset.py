"""ldm.dataset -> mulan_amd.data (create_dataset, create_one_time_eval_dataset)."""
from mulan_amd.data import create_dataset, create_one_time_eval_dataset  # noqa: F401
