"""SURVEY 8(f) rank 1 on the device: a Flax-msgpack checkpoint written the way the reference writes it (aliased gamma
network names, optax.chain(masked, masked) optimizer state, a chunked array) goes through `python -m ldm.eval_bpd
--bpd_eval_method=dense` (ldm/eval_bpd.py:50-62 -> notebook_utils.py:28-39 restore -> :176-191 dense evaluator) on
the HIP path, and the printed bits/dim must agree with the float64 oracle evaluated on the same parameters, images
and noise to +-0.005 (the north-star bar); tests/verify_checkpoint.py (the one-command pin for a released checkpoint)
must come to the same verdict with use_gpu=True."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.oracle_dev import run_oracle
from oracle import torch_ref as tr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config_file,vdm_type,vfe", [("cifar10-conditioned.py", "mulan_velocity", False),
                                                      ("imagenet32.py", "mulan_velocity", True)])
def test_flax_checkpoint_through_eval_bpd_matches_oracle(tmp_path, config_file, vdm_type, vfe):
    import importlib
    import ldm.eval_bpd
    from tests import verify_checkpoint as vc
    from mulan_amd import model as M
    from mulan_amd.config import load_config_file
    from mulan_amd.rng import PRNGKey
    cfgp = os.path.join(ROOT, "ldm", "configs", config_file)
    config = load_config_file(cfgp)
    config.vdm_type = vdm_type
    config.model.velocity_from_epsilon = vfe
    config.model.sm_n_layer = 1
    config.model.forward_n_layer = 1
    E = config.model.sm_n_embd
    os.makedirs(tmp_path / "ck")
    path = str(tmp_path / "ck" / "ckpt-9.flax")
    sd, ref, trees = vc.write_synthetic_flax_checkpoint(path, config, seed=4, step=9)
    imgs = np.random.default_rng(2).integers(0, 256, (3, 32, 32, 3)).astype(np.uint8)
    np.savez(tmp_path / "test.npz", images=imgs)

    T = 8
    importlib.reload(ldm.eval_bpd)
    got = ldm.eval_bpd.main(["--config=" + cfgp, f"--config.vdm_type={vdm_type}",
                             f"--config.model.velocity_from_epsilon={vfe}", "--config.model.sm_n_layer=1",
                             "--config.model.forward_n_layer=1", "--config.training.batch_size_train=2",
                             "--config.training.batch_size_eval=2", "--config.training.substeps=1",
                             "--config.data.dataset=npz:" + str(tmp_path / "test.npz"),
                             "--checkpoint_directory=" + str(tmp_path / "ck"), "--bpd_eval_method=dense",
                             f"--n_timesteps={T}"])
    # the oracle on what the checkpoint's ema_params hold (0.5 x the seeded tree, rounded to fp32), under the
    # evaluator's own noise: loss_fn splits PRNGKey(0) and hands the 'sample' key to the model (notebook_utils.py:178)
    ocfg = vc.oracle_cfg(config)
    half = tr.tree_map(lambda t: (t.detach() * 0.5).float().double(), ref)
    vdm = M.make_vdm(vdm_type, M.VDMConfig(**config.model.to_dict()))
    _, sample_rng = PRNGKey(0).split()
    noise = vdm._noise({"sample": sample_rng}, None, T, torch.device("cuda"), True)
    want = []
    with torch.no_grad():
        for i in range(len(imgs)):
            x = torch.tensor(np.repeat(imgs[i:i + 1], T, axis=0))
            o = run_oracle(lambda P, *a: tr.mulan_forward(P, ocfg, *a), half, x, float(noise["t0"]), noise["gamma_raw"].double().cpu(),
                           noise["eps_0"].double().cpu().view(T, 32, 32, 3), noise["eps"].double().cpu().view(T, 32, 32, 3))
            want.append(float(o["bpd"]))
    print(f"{config_file} E={E}: eval_bpd dense {got:.6f} vs oracle {np.mean(want):.6f}")
    assert abs(got - np.mean(want)) < 0.005, (got, want)

    lines = []
    res = vc.verify(path, config, imgs[:2], n_timesteps=4, use_gpu=True, seed=3, log=lines.append)
    assert res["tree_ok"] and len(res["hip"]) == 2
    assert max(abs(h - o) for h, o in zip(res["hip"], res["oracle"])) < 0.005, (res["hip"], res["oracle"])
