"""python -m ldm.main --config=... --workdir=... [--mode train|eval] [--checkpoint DIR] [--config.a.b=v ...]

Same flags as the reference entry point (ldm/main.py:29-36).  Multi-GPU: launch under
`python -m torch.distributed.run --nproc-per-node N -m ldm.main ...` (one process per MI355X; the
reference used one process with jax.pmap)."""
import logging
import os
import sys

from mulan_amd.config import Flags
from ldm.utils import get_workdir
import ldm.experiment_vdm

FLAGS = Flags()
FLAGS.DEFINE_config_file('config', None, 'Training configuration.')
FLAGS.DEFINE_string('workdir', None, 'Work unit directory.')
FLAGS.DEFINE_string('checkpoint', '', 'Checkpoint to evaluate.')
FLAGS.DEFINE_string('mode', 'train', 'train / eval')
FLAGS.DEFINE_string('model', 'vdm', 'vdm')
FLAGS.DEFINE_string('log_level', 'info', 'info/warning/error')
FLAGS.mark_flags_as_required(['config', 'workdir'])


def main(argv):
    FLAGS.parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    logging.basicConfig(level=getattr(logging, FLAGS.log_level.upper()) if rank == 0 else logging.ERROR)
    logging.warning('=== Start of main() ===')
    if FLAGS.model == 'vdm':
        experiment = ldm.experiment_vdm.Experiment_VDM(FLAGS.config)
    else:
        raise RuntimeError(f"{FLAGS.model} is not implemented")
    if FLAGS.mode == 'train':
        workdir = os.path.join(FLAGS.workdir, get_workdir(sys.argv))
        logging.info('Training at workdir: ' + FLAGS.workdir)
        experiment.train_and_evaluate(workdir)
    elif FLAGS.mode == 'eval':
        print(experiment.evaluate(FLAGS.workdir, FLAGS.checkpoint))
    else:
        raise Exception('Unknown FLAGS.mode')


if __name__ == '__main__':
    main(sys.argv[1:])
