"""Evaluators of the reference's ldm/notebook_utils.py that sit on the hot path: Experiment_Colab
(:28-39, EMA parameters of a checkpoint) and the variational-bound BPD evaluators (:157-191)."""
from mulan_amd.evaluators import Experiment_Colab, eval_bpd_dense_sampling, eval_bpd_sparse_sampling, eval_bpd_ode  # noqa: F401
