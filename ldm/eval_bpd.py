"""python -m ldm.eval_bpd --config=... --checkpoint_directory=... [--checkpoint N] --bpd_eval_method=ode|dense|sparse

Flag surface of the reference (ldm/eval_bpd.py:17-31).  `dense` and `sparse` (variational bound) and `ode` (exact
likelihood: probability-flow ODE, Hutchinson divergence, device-resident RK45) run on the HIP path, sharded over
ranks by test-image index under torchrun."""
import logging
import os
import sys

from mulan_amd.config import Flags
from mulan_amd import checkpoint as ckpt_lib
from ldm.notebook_utils import Experiment_Colab, eval_bpd_dense_sampling, eval_bpd_sparse_sampling, eval_bpd_ode

FLAGS = Flags()
FLAGS.DEFINE_config_file('config', None, 'Training configuration.')
FLAGS.DEFINE_string('checkpoint_directory', None, 'Work unit directory.')
FLAGS.DEFINE_string('checkpoint', None, 'Checkpoint to evaluate.')
FLAGS.DEFINE_string('bpd_eval_method', 'ode', 'Dense / Sparse / ODE sampling to evaluate BPD.')
FLAGS.DEFINE_string('log_level', 'info', 'info/warning/error')
FLAGS.DEFINE_integer('n_timesteps', 128, 'discrete timesteps for dense sampling to evaluate BPD.')
FLAGS.DEFINE_integer('n_is', 20, 'Number of Importance Samples.')
FLAGS.DEFINE_integer('num_iters', 1, 'Number of iterations on test set.')
FLAGS.DEFINE_bool('deterministic_noise', False, 'Deterministic Hutchinson noise.')
FLAGS.DEFINE_string('hutchinson_type', 'Rademacher', 'Hutchinson noise type: (Rademacher/Gaussian)')
FLAGS.DEFINE_float('rtol', 1e-5, 'rtol for the ODE solver')
FLAGS.DEFINE_float('atol', 1e-5, 'atol for the ODE solver')
FLAGS.DEFINE_integer('max_images', 0, 'evaluate only the first N test images (0 = all); not in the reference')
FLAGS.mark_flags_as_required(['config', 'checkpoint_directory'])


def main(argv):
    FLAGS.parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    logging.basicConfig(level=getattr(logging, FLAGS.log_level.upper()) if rank == 0 else logging.ERROR)
    logging.warning(f'Num discrete timesteps: {FLAGS.n_timesteps}')
    ckpt_nums = ckpt_lib.checkpoint_numbers(FLAGS.checkpoint_directory)
    if not ckpt_nums:
        raise SystemExit(f'no ckpt-* files in {FLAGS.checkpoint_directory}')
    print(f'Found ckpts:{ckpt_nums[0]}: {ckpt_nums[-1]}')
    print('BPD eval method:', FLAGS.bpd_eval_method)
    ckpt_num = ckpt_nums[-1] if FLAGS.checkpoint is None else FLAGS.checkpoint
    experiment = Experiment_Colab(FLAGS.config, FLAGS.checkpoint_directory, ckpt_num)
    if FLAGS.bpd_eval_method == 'sparse':
        bpd = eval_bpd_sparse_sampling(experiment, FLAGS.config, max_images=FLAGS.max_images)
    elif FLAGS.bpd_eval_method == 'dense':
        bpd = eval_bpd_dense_sampling(experiment, FLAGS.config, n_timesteps=FLAGS.n_timesteps,
                                      max_images=FLAGS.max_images)
    elif FLAGS.bpd_eval_method == 'ode':
        bpd = eval_bpd_ode(experiment, FLAGS.config, hutchinson_type=FLAGS.hutchinson_type,
                           deterministic_noise=FLAGS.deterministic_noise, num_iters=FLAGS.num_iters,
                           num_is=FLAGS.n_is, rtol=FLAGS.rtol, atol=FLAGS.atol, max_images=FLAGS.max_images)
    else:
        raise SystemExit(f'unknown --bpd_eval_method {FLAGS.bpd_eval_method}')
    if rank == 0:
        print(f'Test BPD:{bpd} ckpt:{ckpt_num}')
    return bpd


if __name__ == '__main__':
    main(sys.argv[1:])
