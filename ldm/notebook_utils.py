"""Evaluators of the reference's ldm/notebook_utils.py: Experiment_Colab (:28-39, EMA parameters of a checkpoint),
the variational-bound BPD evaluators (:157-191) and the exact-likelihood ODE evaluator (:232-373, 446-531)."""
from mulan_amd.evaluators import (Experiment_Colab, Hutchinson, eval_bpd_dense_sampling,  # noqa: F401
                                  eval_bpd_ode, eval_bpd_sparse_sampling, get_ode_likelihood_fn, get_logits, get_sample_fn,
                                  logits_to_embeddings, _get_bpd_offset)
