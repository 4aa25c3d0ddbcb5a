"""ldm.utils.get_workdir: run directory name derived from the command line (ldm/utils.py:38-65)."""
import os
import time


def get_workdir(argv=None):
    import sys
    argv = sys.argv if argv is None else argv
    tag = os.environ.get("SLURM_JOB_ID") or os.environ.get("COMPOSER_RUN_NAME") or time.strftime('%Y%m%d-%H%M%S')
    parts = [tag]
    config_file = "config"
    for a in argv[1:]:
        if a.startswith('--config='):
            config_file = a.split('/')[-1].split('.py')[0]
        elif a.startswith('--workdir=') or a.startswith('--config.ckpt_restore_dir='):
            continue
        elif a.startswith('--config'):
            pieces = a.split('.')
            leaf = pieces[-1]
            if leaf.isnumeric() or len(leaf) == 0:
                leaf = pieces[-2] + '.' + pieces[-1]
            parts.append(leaf)
    return os.path.join(config_file, "-".join(parts))
