"""The HIP path against the committed golden fixtures (tests/golden/*.npz: oracle outputs on seeded inputs, written by
tests/golden/make_golden.py): the closed-form kernels and the whole small model in its four variants."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as tr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_closed_form_kernels_against_fixture():
    from mulan_amd import ops
    ops.lib.load()
    z = np.load(os.path.join(GOLD, "closed_forms.npz"))
    f32 = lambda k: torch.tensor(z[k], dtype=torch.float32).cuda()
    B = z["t"].shape[0]
    g0, g1, gt, gp = ops.poly_gamma(f32("a"), f32("b"), f32("c"), f32("t"), -13.3, 5.0)
    # the fixture's a, b, c are float64; rounding them to the fp32 inputs of the kernel costs a few 1e-6 by itself
    assert _rel(gt.cpu().numpy(), z["g_t"].reshape(B, -1)) < 2e-5
    assert _rel(gp.cpu().numpy(), z["g_prime"].reshape(B, -1)) < 5e-5
    x = torch.tensor(z["x"]).cuda().view(B, -1)
    zt, gbar, recon, klz, v0, v1 = ops.qsample(x, g0, g1, gt, f32("eps_0").view(B, -1), f32("eps").view(B, -1))
    assert _rel(zt.cpu().numpy(), z["z_t"].reshape(B, -1)) < 2e-5
    assert _rel(recon.cpu().numpy(), z["loss_recon"]) < 1e-4 and _rel(klz.cpu().numpy(), z["loss_klz"]) < 1e-5
    net = f32("net").view(B, -1)
    for mode, key in ((0, "loss_diff_velocity"), (1, "loss_diff_vfe"), (2, "loss_diff_epsilon")):
        got = ops.diffusion_loss(mode, x, gt, gp, f32("eps").view(B, -1), zt, net)
        assert _rel(got.cpu().numpy(), z[key]) < 2e-4, key
    emb, kl = ops.topk_embedding(f32("logits"), f32("gamma_raw"), 15)
    assert np.array_equal(np.round(emb.cpu().numpy()), z["embedding"]) and _rel(kl.cpu().numpy(), z["kl_z"]) < 1e-5
    four = ops.fourier_features(f32("fourier_z").view(1, 4, 3).expand(1, 4, 3).contiguous().repeat(1, 256, 1))
    got = four[0, :4, 3:15].cpu().numpy()
    assert np.abs(got - z["fourier"]).max() < 5e-4          # sin / cos of ~800 rad arguments in fp32


@pytest.mark.parametrize("name,vdm_type,unet_type,vfe", [("velocity", "mulan_velocity", "vdm", False),
                                                         ("epsilon", "mulan_epsilon", "vdm", False),
                                                         ("vfe", "mulan_velocity", "vdm", True),
                                                         ("ldm", "mulan_velocity", "ldm", False)])
def test_small_model_against_fixture(name, vdm_type, unet_type, vfe):
    """the parameter tree is rebuilt from the stored seed (the 38 M parameters are not in the file)"""
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    from tests.test_gpu_model import make_cfg
    z = np.load(os.path.join(GOLD, "tiny_model.npz"))
    cfg, ocfg = make_cfg(vdm_type, unet_type, vfe)
    ref_params = tr.init_params(ocfg, seed=int(z["param_seed"]), dtype=torch.float64)
    vdm = M.make_vdm(vdm_type, cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    B = z["x"].shape[0]
    f32 = lambda a: torch.tensor(a, dtype=torch.float32).cuda()
    noise = dict(t0=float(z["t0"]), gamma_raw=f32(z["gamma_raw"]), eps_0=f32(z["eps_0"]).view(B, -1),
                 eps=f32(z["eps"]).view(B, -1))
    with torch.no_grad():
        out = vdm.apply(params, torch.tensor(z["x"]).cuda(), None, None, step=0, rngs=None, deterministic=True, noise=noise)
    assert _rel(out.loss_recon.cpu().numpy(), z[f"{name}_recon"]) < 1e-4
    assert _rel(out.loss_klz.cpu().numpy(), z[f"{name}_klz"]) < 1e-4
    assert _rel(out.loss_diff.cpu().numpy(), z[f"{name}_diff"]) < 5e-4
    bpd = float((out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) / (3072 * np.log(2.0)))
    assert abs(bpd - float(z[f"{name}_bpd"])) < 0.005          # the north-star bar, absolute bits/dim
