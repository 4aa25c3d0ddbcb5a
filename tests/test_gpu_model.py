"""End-to-end parity of the HIP model (through mulan_amd.model / the C ABI) with the float64 oracle:
forward ELBO terms and every parameter gradient of one MuLAN step, with identical explicit noise and
the oracle reproducing the kernel's Philox dropout masks."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mulan_np as onp
from oracle import torch_ref as tr

from tests.oracle_dev import pin_oracle_to_host, run_oracle


def make_cfg(vdm_type="mulan_velocity", unet_type="vdm", vfe=False, n_layer=1, fwd_layers=1, E=128, with_attention=False):
    from mulan_amd.model import VDMConfig
    return VDMConfig(vocab_size=256, sample_softmax=False, antithetic_time_sampling=True, with_fourier_features=True,
                     with_attention=with_attention, gamma_type='poly_fixedend', gamma_min=-13.3, gamma_max=5.0,
                     sm_n_timesteps=0, sm_n_embd=E, sm_n_layer=n_layer, sm_pdrop=0.1, forward_n_layer=fwd_layers,
                     latent_size=50, latent_k=15, encoder='unet', latent_type='topk', z_conditioning=True,
                     reparam_type='true', unet_type=unet_type, velocity_from_epsilon=vfe, condition='input',
                     sigma_type='no_blur', sigma_prior=1.0), dict(
        vdm_type=vdm_type, n_embd=E, n_layer=n_layer, forward_n_layer=fwd_layers, latent_k=15, unet_type=unet_type,
        velocity_from_epsilon=vfe, with_attention=with_attention)


def oracle_masks(names, key, B, C, keep):
    masks = {}
    for site, name in enumerate(names, start=1):
        masks[name] = torch.tensor(onp.dropout_mask((B, 32, 32, C), keep, key.v, site << 34))
    return masks


def block_names(n_down, with_up):
    names = [f"down.block_{i}" for i in range(n_down)] + ["mid.block_1", "mid.block_2"]
    if with_up:
        names += [f"up.block_{i}" for i in range(n_down + 1)]
    return names


def to_device_tree(flax_tree, like):
    from mulan_amd import model as M
    return M.from_flax_layout(flax_tree, like)


def run_case(vdm_type, unet_type, vfe, train, E=128, tol=1.0, with_attention=False, n_layer=1, fwd_layers=1, B=4):
    """tol scales the fp32-vs-float64 bars (E = 256 doubles / quadruples every contraction length)"""
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg(vdm_type, unet_type, vfe, E=E, with_attention=with_attention, n_layer=n_layer,
                         fwd_layers=fwd_layers)
    rng = np.random.default_rng(17)
    ref_params = tr.init_params(ocfg, seed=3, dtype=torch.float64)
    for _, leaf in tr.tree_leaves(ref_params):
        leaf.requires_grad_(True)
    vdm = M.make_vdm(vdm_type, cfg)
    tmpl = vdm.init(PRNGKey(0))
    params = M.tree_map(lambda t: t.cuda(), tmpl)
    to_device_tree(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    for _, leaf in M.tree_leaves(params):
        leaf.requires_grad_(True)

    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    raw = rng.gamma(1.0 / 15, size=(10, B, 50))
    e0, e = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    t0 = 0.37
    noise = dict(t0=t0, gamma_raw=torch.tensor(raw, dtype=torch.float32).cuda(),
                 eps_0=torch.tensor(e0, dtype=torch.float32).cuda(), eps=torch.tensor(e, dtype=torch.float32).cuda())
    keep = 1.0
    enc_masks = score_masks = None
    rngs = None
    if train:
        keep = float(np.float32(0.9))
        dkey = PRNGKey(99)
        k_enc, k_score = dkey.split(2)
        enc_masks = oracle_masks(block_names(fwd_layers, False), k_enc, B, E, 0.9)
        score_masks = oracle_masks(block_names(n_layer, True), k_score, B, E, 0.9)
        rngs = {"dropout": dkey}
    # (the oracle's float64 pass -- and, in training mode, its backward pass -- through tests/oracle_dev.run_oracle)
    ref = run_oracle(lambda P, *a, **k: tr.mulan_forward(P, ocfg, *a, keep=keep, **k), ref_params, torch.tensor(x), t0,
                     torch.tensor(raw), torch.tensor(e0).view(B, 32, 32, 3), torch.tensor(e).view(B, 32, 32, 3),
                     enc_masks=enc_masks, score_masks=score_masks, backward="bpd" if train else None)
    out, aux = vdm.apply(params, torch.tensor(x).cuda(), None, None, step=0, rngs=rngs, deterministic=not train,
                         noise=noise, return_aux=True)
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
    # same hard top-k latent, same z_t
    assert np.array_equal(np.round(aux["emb"].detach().cpu().numpy()), np.round(ref["aux"]["emb"].detach().numpy()))
    assert rel(aux["zt"].detach().cpu().numpy(), ref["aux"]["z_t"].detach().numpy().reshape(B, -1)) < 1e-5
    assert rel(aux["net"].detach().cpu().numpy(), ref["aux"]["net"].detach().numpy().reshape(B, -1)) < 2e-4 * tol
    assert rel(out.loss_recon.detach().cpu().numpy(), ref["loss_recon"].detach().numpy()) < 1e-4 * tol
    assert rel(out.loss_klz.detach().cpu().numpy(), ref["loss_klz"].detach().numpy()) < 1e-4 * tol
    assert rel(out.loss_diff.detach().cpu().numpy(), ref["loss_diff"].detach().numpy()) < 5e-4 * tol
    r = 1.0 / (3072 * np.log(2.0))
    bpd = (out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) * r
    # BPD: the north-star bar, +-0.005 bits/dim absolute (at random init the BPD is ~12.5: 4e-4 relative)
    assert abs(float(bpd) - float(ref["bpd"])) < 0.005, (float(bpd), float(ref["bpd"]))
    if not train:
        return
    bpd.backward()
    flax_grads = M.to_flax_layout(M.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), params))
    worst = []
    for path, leaf in tr.tree_leaves(ref_params):
        g = flax_grads
        for k in path:
            g = g[k]
        rg = leaf.grad.numpy() if leaf.grad is not None else np.zeros(tuple(leaf.shape))
        scale = np.abs(rg).max()
        err = np.abs(g.cpu().double().numpy() - rg).max()
        worst.append((err / (scale + 1e-12) if scale > 1e-12 else err, "/".join(path)))
    worst.sort(reverse=True)
    # fp32 end-to-end through ~40 kernels vs float64: 2e-3 of each leaf's gradient scale
    assert worst[0][0] < 2e-3 * tol, worst[:8]


@pytest.mark.parametrize("vdm_type,unet_type,vfe", [("mulan_velocity", "vdm", False), ("mulan_velocity", "vdm", True),
                                                    ("mulan_epsilon", "vdm", False), ("mulan_velocity", "ldm", False)])
def test_mulan_forward_eval(vdm_type, unet_type, vfe):
    run_case(vdm_type, unet_type, vfe, train=False)


@pytest.mark.parametrize("vdm_type,unet_type,vfe", [("mulan_velocity", "vdm", False), ("mulan_epsilon", "vdm", False),
                                                    ("mulan_velocity", "vdm", True), ("mulan_velocity", "ldm", False)])
def test_mulan_train_gradients(vdm_type, unet_type, vfe):
    run_case(vdm_type, unet_type, vfe, train=True)


def test_zero_init_network_is_identity():
    """with the reference's zero-initialised conv2/cond_proj/proj_out/conv_out the score model
    returns z exactly (ldm/model_vdm.py:378-386)."""
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, _ = make_cfg()
    vdm = M.make_vdm("mulan_velocity", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(1)))
    z = torch.randn(2, 1024, 3).cuda()
    out = M.score_unet(params["score_model"], cfg, z, torch.tensor([0.5, -3.0]).cuda(), torch.randn(2, 50).cuda(),
                       M._Drop(None, 0.0))
    assert torch.equal(out, z)


def test_train_steps_reduce_loss_and_ema():
    """3 optimiser steps through Experiment_VDM.train_step on a fixed synthetic batch: finite loss that goes
    down, step counter, EMA lagging the parameters."""
    import os
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = 'synthetic'
    config.model.sm_n_layer = 1
    config.model.forward_n_layer = 1
    config.training.batch_size_train = 4
    config.training.batch_size_eval = 4
    config.training.substeps = 1
    config.training.num_steps_lr_warmup = 1
    exp = Experiment_VDM(config)
    batch = next(exp.train_iter)
    sub = {k: v[0] for k, v in batch.items()}
    losses = []
    for _ in range(3):
        exp.state, m = exp.train_step(exp._train_rng.fold_in(0), exp.state, sub)
        losses.append(float(m['scalars']['train_bpd']))
    assert all(np.isfinite(losses)), losses
    assert exp.state.step == 3
    assert not torch.equal(exp.state.flat, exp.state.ema)
    m = exp.eval_step(exp._eval_rng, exp.state.ema_params, sub, 0)
    assert np.isfinite(float(m['scalars']['eval_bpd']))


def test_mulan_epsilon_discrete_time_T1000():
    """sm_n_timesteps = 1000 branch of model_mulan_epsilon (ldm/model_mulan_epsilon.py:295-297,348-355): forward
    terms and a few gradients vs the float64 oracle."""
    import dataclasses
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg("mulan_epsilon")
    cfg = dataclasses.replace(cfg, sm_n_timesteps=1000)
    ocfg = dict(ocfg, n_timesteps=1000)
    B = 4
    rng = np.random.default_rng(5)
    ref_params = tr.init_params(ocfg, seed=9, dtype=torch.float64)
    for _, leaf in tr.tree_leaves(ref_params):
        leaf.requires_grad_(True)
    vdm = M.make_vdm("mulan_epsilon", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    for _, leaf in M.tree_leaves(params):
        leaf.requires_grad_(True)
    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    raw = rng.gamma(1.0 / 15, size=(10, B, 50))
    e0, e = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    noise = dict(t0=0.411, gamma_raw=torch.tensor(raw, dtype=torch.float32).cuda(),
                 eps_0=torch.tensor(e0, dtype=torch.float32).cuda(), eps=torch.tensor(e, dtype=torch.float32).cuda())
    ref = tr.mulan_forward(ref_params, ocfg, torch.tensor(x), 0.411, torch.tensor(raw),
                           torch.tensor(e0).view(B, 32, 32, 3), torch.tensor(e).view(B, 32, 32, 3))
    out = vdm.apply(params, torch.tensor(x).cuda(), None, None, step=0, rngs=None, deterministic=True, noise=noise)
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
    # fp32 expm1(gamma_t - gamma_s) of a ~1e-2 difference of O(10) numbers: 2e-3 relative on the loss
    assert rel(out.loss_diff.detach().cpu().numpy(), ref["loss_diff"].detach().numpy()) < 2e-3
    (out.loss_diff.mean()).backward()
    ref["loss_diff"].mean().backward()
    g = params["gamma"]["dense_out_b"]["bias"].grad.cpu().double().numpy()
    rg = ref_params["gamma"]["dense_out_b"]["bias"].grad.numpy()
    assert rel(g, rg) < 2e-2
    g = params["score_model"]["conv_out"]["kernel"].grad.cpu().double().numpy()
    rg = ref_params["score_model"]["conv_out"]["kernel"].grad.numpy()
    assert rel(g, rg) < 2e-3


@pytest.mark.parametrize("gamma_type,T", [("fixed", 0), ("learnable_scalar", 0), ("fixed", 1000), ("learnable_nnet", 0)])
def test_plain_vdm_matches_oracle(gamma_type, T):
    """BASELINE config #1 model (ldm/model_vdm.py:95-180): scalar schedule VDM, forward + schedule / U-Net grads."""
    import dataclasses
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg()
    cfg = dataclasses.replace(cfg, gamma_type=gamma_type, sm_n_timesteps=T, z_conditioning=False, reparam_type='noise')
    ocfg = dict(ocfg, n_timesteps=T)
    B = 4
    rng = np.random.default_rng(6)
    full = tr.init_params(ocfg, seed=4, dtype=torch.float64)
    ref_params = {"score_model": full["score_model"]}
    ref_params["score_model"]["dense0"]["kernel"] = ref_params["score_model"]["dense0"]["kernel"][:129].clone()
    if gamma_type == "learnable_scalar":
        ref_params["gamma"] = {"w": torch.tensor([-17.0], dtype=torch.float64), "b": torch.tensor([-12.5], dtype=torch.float64)}
    if gamma_type == "learnable_nnet":      # NoiseSchedule_NNet (ldm/model_vdm.py:471-509), strongly non-linear on purpose
        gg = torch.Generator().manual_seed(8)
        ref_params["gamma"] = {
            "l1": {"kernel": torch.tensor([[-16.0]], dtype=torch.float64), "bias": torch.tensor([-12.0], dtype=torch.float64)},
            "l2": {"kernel": torch.randn(1, 1024, generator=gg, dtype=torch.float64) * 3,
                   "bias": torch.randn(1024, generator=gg, dtype=torch.float64)},
            "l3": {"kernel": torch.randn(1024, 1, generator=gg, dtype=torch.float64) * 2}}
    for _, leaf in tr.tree_leaves(ref_params):
        leaf.requires_grad_(True)
    vdm = M.make_vdm("vdm", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    for _, leaf in M.tree_leaves(params):
        leaf.requires_grad_(True)
    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    e0, e = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    # t0 keeps t * T away from integers: ceil(t * T) at an exact grid point flips with fp32 vs fp64 rounding
    noise = dict(t0=0.2713, eps_0=torch.tensor(e0, dtype=torch.float32).cuda(), eps=torch.tensor(e, dtype=torch.float32).cuda())
    ref = tr.plain_vdm_forward(ref_params, ocfg, torch.tensor(x), 0.2713, torch.tensor(e0).view(B, 32, 32, 3),
                               torch.tensor(e).view(B, 32, 32, 3))
    out = vdm.apply(params, torch.tensor(x).cuda(), None, torch.zeros(B, dtype=torch.uint8).cuda(), step=0, rngs=None,
                    deterministic=True, noise=noise)
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
    assert rel(out.loss_recon.detach().cpu().numpy(), ref["loss_recon"].detach().numpy()) < 1e-4
    assert rel(out.loss_klz.detach().cpu().numpy(), ref["loss_klz"].detach().numpy()) < 1e-4
    assert rel(out.loss_diff.detach().cpu().numpy(), ref["loss_diff"].detach().numpy()) < (2e-3 if T else 5e-4)
    r = 1.0 / (3072 * np.log(2.0))
    ((out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) * r).backward()
    ref["bpd"].backward()
    g = params["score_model"]["conv_out"]["kernel"].grad.cpu().double().numpy()
    assert rel(g, ref_params["score_model"]["conv_out"]["kernel"].grad.numpy()) < 2e-3
    if gamma_type == "learnable_scalar":
        for k in ("w", "b"):
            assert rel(params["gamma"][k].grad.cpu().double().numpy(), ref_params["gamma"][k].grad.numpy()) < 5e-3
    if gamma_type == "learnable_nnet":
        for l, k in (("l1", "kernel"), ("l1", "bias"), ("l2", "kernel"), ("l2", "bias"), ("l3", "kernel")):
            assert rel(params["gamma"][l][k].grad.cpu().double().numpy(), ref_params["gamma"][l][k].grad.numpy()) < 5e-3, (l, k)


def test_full_depth_forward_bpd_parity(monkeypatch):
    """the shipped depth (sm_n_layer = 32: 67 ResnetBlocks in the score U-Net, forward_n_layer = 4 in the encoder) in
    evaluation mode against the float64 oracle: the north-star bar is +-0.005 bits/dim; the split-operand kernels
    must not accumulate error over 140 chained convolutions.  The oracle of this test runs on the HOST (checker
    independence: tests/oracle_dev.py)"""
    from mulan_amd import model as M
    pin_oracle_to_host(monkeypatch)
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg("mulan_velocity", "vdm", True, n_layer=32, fwd_layers=4)
    B = 2
    rng = np.random.default_rng(31)
    ref_params = tr.init_params(ocfg, seed=9, dtype=torch.float64)
    vdm = M.make_vdm("mulan_velocity", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    to_device_tree(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    raw = rng.gamma(1.0 / 15, size=(10, B, 50))
    e0, e = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    f32 = lambda a: torch.tensor(a, dtype=torch.float32).cuda()
    noise = dict(t0=0.41, gamma_raw=f32(raw), eps_0=f32(e0), eps=f32(e))
    with torch.no_grad():
        ref = run_oracle(lambda P, *a: tr.mulan_forward(P, ocfg, *a), ref_params, torch.tensor(x), 0.41, torch.tensor(raw),
                         torch.tensor(e0).view(B, 32, 32, 3), torch.tensor(e).view(B, 32, 32, 3))
        out, aux = vdm.apply(params, torch.tensor(x).cuda(), None, None, step=0, rngs=None, deterministic=True,
                             noise=noise, return_aux=True)
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
    assert np.array_equal(np.round(aux["emb"].cpu().numpy()), np.round(ref["aux"]["emb"].numpy()))
    net_err = rel(aux["net"].cpu().numpy(), ref["aux"]["net"].numpy().reshape(B, -1))
    r = 1.0 / (3072 * np.log(2.0))
    bpd = float((out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) * r)
    print(f"full depth: net rel err {net_err:.2e}, bpd hip {bpd:.6f} oracle {float(ref['bpd']):.6f}")
    assert net_err < 2e-3, net_err
    assert abs(bpd - float(ref["bpd"])) < 0.005, \
        (bpd, float(ref["bpd"]), net_err)


def test_deeper_stack_train_gradients(monkeypatch):
    """8 + 2 layers (19 + 4 ResnetBlocks), training mode with dropout: losses and every parameter gradient against
    float64 autograd through the whole stack (error growth with depth; the shipped depth is covered forward-only by
    test_full_depth_forward_bpd_parity).  The oracle of this test -- forward AND backward -- runs on the HOST (checker
    independence: tests/oracle_dev.py)."""
    pin_oracle_to_host(monkeypatch)
    run_case("mulan_velocity", "vdm", False, True, n_layer=8, fwd_layers=2)


def test_with_attention_after_every_block():
    """config.with_attention=True: an AttnBlock after each down / up ResnetBlock of both U-Nets
    (ldm/model_vdm.py:356-357, 371-372; ldm/model_mulan_epsilon.py:133-134): losses and every parameter gradient"""
    run_case("mulan_velocity", "vdm", False, True, with_attention=True)


def test_independent_times_and_gumbel_topk_noise():
    """antithetic_time_sampling=False (t ~ U[0,1)^B given explicitly) and topk_noise_type='gumbel' (epsilon model,
    ldm/model_mulan_epsilon.py:236-239, 299-300) against the oracle: forward terms and BPD"""
    import dataclasses
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg("mulan_epsilon")
    cfg = dataclasses.replace(cfg, antithetic_time_sampling=False, topk_noise_type='gumbel')
    B = 4
    rng = np.random.default_rng(23)
    ref_params = tr.init_params(ocfg, seed=3, dtype=torch.float64)
    vdm = M.make_vdm("mulan_epsilon", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    to_device_tree(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    t = rng.uniform(0.02, 0.98, B)
    gum = rng.gumbel(size=(B, 50))
    e0, e = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    f32 = lambda a: torch.tensor(a, dtype=torch.float32).cuda()
    noise = dict(t=f32(t), gumbel=f32(gum), eps_0=f32(e0), eps=f32(e))
    ref = tr.mulan_forward(ref_params, ocfg, torch.tensor(x), 0.0, None, torch.tensor(e0).view(B, 32, 32, 3),
                           torch.tensor(e).view(B, 32, 32, 3), t=torch.tensor(t.astype(np.float32).astype(np.float64)),
                           gumbel=torch.tensor(gum.astype(np.float32).astype(np.float64)))
    out, aux = vdm.apply(params, torch.tensor(x).cuda(), None, None, step=0, rngs=None, deterministic=True, noise=noise,
                         return_aux=True)
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
    assert np.array_equal(np.round(aux["emb"].cpu().numpy()), np.round(ref["aux"]["emb"].detach().numpy()))
    assert rel(out.loss_diff.cpu().numpy(), ref["loss_diff"].detach().numpy()) < 5e-4
    assert rel(out.loss_klz.cpu().numpy(), ref["loss_klz"].detach().numpy()) < 1e-4
    r = 1.0 / (3072 * np.log(2.0))
    bpd = float((out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) * r)
    assert abs(bpd - float(ref["bpd"])) < 0.005
    # drawn from the key when not given: shapes / ranges only
    out2 = vdm.apply(params, torch.tensor(x).cuda(), None, None, step=0, rngs={"sample": PRNGKey(5)}, deterministic=True)
    assert bool(torch.isfinite(out2.loss_diff).all())
    with pytest.raises(ValueError):
        M.make_vdm("mulan_velocity", cfg)          # the velocity model has no gumbel variant


def test_cli_train_checkpoint_and_dense_eval(tmp_path):
    """H2: python -m ldm.main (2 optimiser steps, synthetic data, checkpoint) then python -m ldm.eval_bpd dense and
    sparse on an npz test set through the same flag surface as the reference."""
    import os
    import ldm.main
    import ldm.eval_bpd
    from mulan_amd import checkpoint as ck
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgp = os.path.join(root, "ldm", "configs", "cifar10-conditioned.py")
    imgs = np.random.default_rng(0).integers(0, 256, (6, 32, 32, 3)).astype(np.uint8)
    np.savez(tmp_path / "test.npz", images=imgs)
    common = ["--config=" + cfgp, "--config.model.sm_n_layer=1", "--config.model.forward_n_layer=1",
              "--config.training.batch_size_train=4", "--config.training.batch_size_eval=2",
              "--config.training.substeps=1", "--config.training.num_steps_train=2", "--config.training.num_steps_eval=1",
              "--config.training.steps_per_logging=1", "--config.training.steps_per_eval=2",
              "--config.training.steps_per_save=2", "--config.training.sample_timesteps=3"]
    ldm.main.main(common + ["--config.data.dataset=synthetic", "--workdir=" + str(tmp_path / "run")])
    ckdirs = [os.path.join(dp, d) for dp, dn, _ in os.walk(tmp_path / "run") for d in dn if d == "checkpoints"]
    assert len(ckdirs) == 1 and ck.checkpoint_numbers(ckdirs[0]) == [1]
    sd = ck.restore_dict(ckdirs[0])
    assert sd["step"] == 2 and "ema_params" in sd
    ldm.eval_bpd.FLAGS.__init__()
    import importlib
    importlib.reload(ldm.eval_bpd)
    ldm.eval_bpd.main(common + ["--config.data.dataset=npz:" + str(tmp_path / "test.npz"),
                                "--checkpoint_directory=" + ckdirs[0], "--bpd_eval_method=dense", "--n_timesteps=8",
                                "--max_images=3"])
    importlib.reload(ldm.eval_bpd)
    ldm.eval_bpd.main(common + ["--config.data.dataset=npz:" + str(tmp_path / "test.npz"),
                                "--checkpoint_directory=" + ckdirs[0], "--checkpoint=1", "--bpd_eval_method=sparse"])
    # the reference's default method: exact likelihood by the probability-flow ODE (2 importance samples, loose
    # tolerances to keep the run short)
    importlib.reload(ldm.eval_bpd)
    ldm.eval_bpd.main(common + ["--config.data.dataset=npz:" + str(tmp_path / "test.npz"),
                                "--checkpoint_directory=" + ckdirs[0], "--n_is=2", "--rtol=1e-2", "--atol=1e-2",
                                "--max_images=2"])
    # sample grids written at the evaluation point of the training run
    assert any(f.startswith("samples_") and f.endswith(".ppm") for dp, _, fs in os.walk(tmp_path / "run") for f in fs)


def test_gradient_sink_equals_autograd_accumulation():
    """The backward kernels write parameter gradients straight into TrainState's flat buffer (`_gview` sink);
    the result must equal ordinary autograd accumulation bit for bit, and .grad must alias the flat buffer."""
    from mulan_amd import model as M
    from mulan_amd import ops as _ops
    from mulan_amd.rng import PRNGKey
    from mulan_amd.train_state import TrainState
    cfg, _ = make_cfg()
    vdm = M.make_vdm("mulan_velocity", cfg)
    st = TrainState.create(apply_fn=vdm.apply, variables={"params": vdm.init(PRNGKey(3))}, device=torch.device("cuda"))
    # this test is about the per-leaf sink; the grouped FiLM projections (one super-parameter per U-Net, different
    # summation order for d cond) have their own test below
    saved_group = _ops.GROUP_COND_PROJ
    _ops.GROUP_COND_PROJ = False
    with torch.no_grad():   # un-zero the zero-initialised tensors so every gradient is non-trivial
        st.flat.add_(0.01 * torch.randn(st.flat.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(0)))
        st.params["score_model"]["conv_in"]["kernel"][:, :, 15, :] = 0
    x = torch.randint(0, 256, (4, 32, 32, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
    rngs = {"sample": PRNGKey(5), "dropout": PRNGKey(6)}

    def grads(use_sink):
        for leaf in st._leaves:
            if not use_sink and hasattr(leaf, "_gview"):
                leaf._saved_gview = leaf._gview
                del leaf._gview
        st.zero_grad() if use_sink else [setattr(l, "grad", None) for l in st._leaves]
        out = vdm.apply(st.params, x, None, None, step=0, rngs=rngs, deterministic=False)
        (out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()).backward()
        if use_sink:
            aliased = sum(int(l.grad is not None and l.grad.data_ptr() == l._gview.data_ptr()) for l in st._leaves)
            st.collect_grads()
            return st.grad.clone(), aliased
        flat = torch.zeros_like(st.grad)
        for leaf, (path, off, shape) in zip(st._leaves, st.layout):
            if leaf.grad is not None:
                flat[off:off + leaf.numel()] = leaf.grad.reshape(-1)
            leaf._gview = leaf._saved_gview
        return flat, 0

    try:
        g_sink, aliased = grads(True)
        g_ref, _ = grads(False)
    finally:
        _ops.GROUP_COND_PROJ = saved_group
    # bit-identical, except the bias gradients: with a sink they are summed over the samples inside the GroupNorm
    # backward kernel, without one by a column-sum launch (another summation order, same fp32 sums)
    for path, off, shape in st.layout:
        n = int(np.prod(shape))
        a, r = g_sink[off:off + n], g_ref[off:off + n]
        if path[-1] == "bias":
            assert float((a - r).abs().max()) <= 2e-6 * float(r.abs().max()) + 1e-30, path
        else:
            assert torch.equal(a, r), path
    assert aliased >= 0.9 * len(st._leaves), (aliased, len(st._leaves))   # autograd adopted the flat views


@pytest.mark.timeout(900)
@pytest.mark.parametrize("graph,launcher,configs,world", [("", "self", True, 2), ("0", "torchrun", False, 2), ("", "self", True, 8)])
def test_bench_ranks_share_one_gpu(graph, launcher, configs, world):
    """bench.py's N > 1 control flow (sharded batches, bucketed side-stream all-reduce, barriers, max-over-ranks timing,
    rank-0 JSON) with two ranks on the one GPU of this box.  RCCL refuses two ranks per device, so the collective
    backend is gloo over the device tensors here; the calls are the same torch.distributed ones.  graph = "": the
    multi-rank default, which at 8 images per rank is the replayed backward with the collectives and the optimizer behind
    it (the host could not keep up with an eager step); "0": MULAN_HIP_GRAPH=0, the eager step whose all-reduce overlaps
    the backward pass (the default from 96 images per rank at E = 128).  launcher = "self": plain `python bench.py --gpus 2` -- the bench starts
    torch.distributed.run itself as a child process and relays rank 0's line (the reference needs no launcher either,
    ldm/experiment.py:89-95); "torchrun": the driver's command line.  configs: the default at every N -- the line also
    carries BASELINE configs[2] at a GLOBAL batch (strong scaling), configs[3] per rank, configs[4] through the sharded
    evaluator (eval_bpd_dense_sampling, one (sum, count) all-reduce) and the sampler / ODE entries -- here at the test
    sizes of --configs-small (the full sizes run in every `python bench.py`: profiles/r04_bench_n1.json.log); False:
    --no-also-configs."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {**os.environ, "MULAN_DIST_BACKEND": "gloo", "MULAN_FORCE_DEVICE": "0"}
    for k in ("MULAN_HIP_GRAPH", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MULAN_GRAPH_OVERLAP"):
        env.pop(k, None)
    if graph:
        env["MULAN_HIP_GRAPH"] = graph
    overlap = graph != "0" and world == 2
    if overlap:
        env["MULAN_GRAPH_OVERLAP"] = "1"       # opt-in (round 5): replay + signal hand-off; the 8-rank run keeps the default
    # all three use --small-depth (1 or 2 ResnetBlocks per stage: the control flow, not the sizes -- the full depth runs in
    # every other whole-model test and in every `python bench.py`); 8 ranks: 4 images per rank
    per = 8 if world == 2 else 4
    args = [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--per-gpu-batch", str(per),
            "--also-steps", "2"] + (["--configs-small", "--small-depth", "1"] if configs
                                    else ["--no-also-configs", "--small-depth", "2"])
    if launcher == "self":
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=840, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["global_batch"] == per * world and out["value"] > 0
    assert out["roofline"] is not None and out["cpu_baseline"] is None
    assert out.get("test_depth") == (1 if configs else 2)
    if configs:
        c = out["configs"]
        assert set(c) == {"3", "4", "5", "sampler", "ode"}       # ("1", the one-device configs[0], only at N = 1)
        for key in ("3", "4"):          # every training entry says what its replicas and its chip did
            assert c[key]["multi_gpu"]["replicas_in_sync"] is True and len(c[key]["multi_gpu"]["ms_per_step_by_rank"]) == world
            assert "sclk_mhz" in c[key]["chip"] and c[key]["chip"]["allocator_peak_gb"] > 0
        # configs[2]: the GLOBAL batch 32 split over the ranks (512 // world at full size); configs[3]: 8 images per rank
        assert c["3"]["global_batch"] == 32 and c["4"]["global_batch"] == 8 * world and c["3"]["value"] > 0 and c["4"]["value"] > 0
        assert f"sharded over {world} rank(s)" in c["5"]["workload"] and c["5"]["value"] > 0 and math.isfinite(c["5"]["bpd_random_init"])
        assert c["sampler"]["finite"] and c["sampler"]["value"] > 0 and c["ode"]["finite"] and c["ode"]["nfev"] >= 8
    else:
        assert out["configs"] is None
    assert out["hip_graph"] == (graph != "0")
    assert out["collective"]["backend"] == "gloo" and out["collective"]["rccl_ranks"] == 0    # (nccl on a multi-GPU node)
    # the run proves itself: replicas bit-identical after the timed steps (parameters and EMA), every
    # rank's own step time, the exposed part of the gradient exchange, the N = 1-equivalent throughput of this run
    mg = out["multi_gpu"]
    assert mg["replicas_in_sync"] is True and mg["max_abs_param_difference_over_ranks"] == 0.0 and mg["max_abs_ema_difference_over_ranks"] == 0.0
    assert len(mg["ms_per_step_by_rank"]) == world and mg["ms_per_step_min"] <= mg["ms_per_step_max"]
    assert abs(mg["ms_per_step_max"] - out["ms_per_step"]) < 0.25 * out["ms_per_step"] + 1.0
    assert mg["ms_per_step_without_collectives"] > 0 and math.isfinite(mg["allreduce_exposed_ms"])
    assert mg["n1_equivalent"]["images_per_sec_per_gpu"] > 0 and 0 < mg["n1_equivalent"]["scaling_efficiency"] < 1.5
    assert "replicas_out_of_sync" not in out and out["dtype"].startswith("f32 (f16x3 split") and out["step_roofline_frac"] > 0
    assert out["chip"]["allocator_peak_gb"] > 0 and "sclk_mhz" in out["chip"]
    ov = out["collective"]["replay_overlap"]
    if overlap:
        assert ov["handoff"] == "signal" and len(ov["marked"]) >= 2 and len(ov["released_ms_before_graph_end"]) == len(ov["marked"])
        assert max(ov["released_ms_before_graph_end"]) > 0.5          # some bucket was released before the graph ended
    else:      # eager overlapped step, or the default multi-rank replay: collectives behind the graph
        assert ov is None


@pytest.mark.timeout(900)
@pytest.mark.parametrize("handoff", ["signal", "event"])
def test_two_rank_replayed_step_overlaps_the_allreduce_and_equals_the_eager_step(handoff):
    """replay AND overlap.  Two ranks (gloo over the device tensors of the one GPU here; RCCL
    where two GPUs are visible): the train step replayed as a HIP graph, each gradient bucket all-reduced outside the
    graph behind the signal word a kernel node of the graph sets after that bucket (handoff = "signal": mulan_signal_set /
    hipStreamWaitValue32, the shipped form) or behind the event-record node planted there (handoff = "event":
    MULAN_OVERLAP_SIGNAL=0, the fallback), equals the eager overlapped step bit for bit over four optimizer steps --
    parameters, EMA, Adam moments, reduced gradient, logged bits/dim -- for the epsilon and the velocity model, and the
    buckets are issued in the order they were marked (tests/overlap_replay_check.py)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {**os.environ, "MULAN_BUCKET_MB": "16", "MULAN_OVERLAP_SIGNAL": "1" if handoff == "signal" else "0"}
    if handoff == "event":       # the fallback hand-off: one model is enough (the signal form runs both)
        env["MULAN_CHECK_MODELS"] = "mulan_epsilon"
    if torch.cuda.device_count() < 2:
        env.update({"MULAN_DIST_BACKEND": "gloo", "MULAN_FORCE_DEVICE": "0"})
    for k in ("MULAN_HIP_GRAPH", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MULAN_GRAPH_OVERLAP"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "tests", "overlap_replay_check.py")],
                       capture_output=True, text=True, timeout=840, env=env, cwd=root)
    assert r.returncode == 0 and "OVERLAP_REPLAY_CHECK ok" in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.timeout(900)
def test_imagenet32_width_train_parity():
    """E = 256 (ldm/configs/imagenet32.py): 256 / 512-channel convolutions, GroupNorm over 512 concatenated channels,
    two cout blocks per convolution tile, four weight-gradient tiles -- same parity bars as the CIFAR width.  The
    attention blocks of a training step at this width run on the fused kernels (round 4): no softmax / score-matrix
    launch, i.e. no [B, 1024, 1024] tensor, forward or backward (BASELINE configs[3])."""
    from mulan_amd import ops
    names = []
    real = ops.call
    ops.call = lambda n, *a: (names.append(n), real(n, *a))[1]
    try:
        run_case("mulan_velocity", "vdm", True, train=True, E=256, tol=3.0)
    finally:
        ops.call = real
    assert names.count("mulan_attention_fwd_f16x3") >= 2 and names.count("mulan_attention_bwd_f16x3") >= 2, names.count
    assert not any(n.startswith("mulan_softmax") for n in names)


def test_grouped_film_projections_match_per_block_gemms():
    """TrainState lays the cond_proj kernels of a U-Net back to back; ops.cond_proj then serves all blocks from one
    batched GEMM (forward and both gradients).  Values and every gradient in the flat buffer equal the per-block path."""
    from mulan_amd import ops
    from mulan_amd.train_state import TrainState
    ops.lib.load()
    g = torch.Generator().manual_seed(2)
    blk = lambda: {"cond_proj": {"kernel": torch.randn(512, 128, generator=g) * 0.05},
                   "conv1": {"kernel": torch.randn(3, 3, 16, 16, generator=g), "bias": torch.randn(16, generator=g)}}
    tree = {"score_model": {f"down.block_{i}": blk() for i in range(3)},
            "encoder_model": {f"down.block_{i}": blk() for i in range(2)},
            "gamma": {"dense": {"kernel": torch.randn(8, 8, generator=g)}}}
    cond = torch.randn(4, 512, generator=g).cuda()
    gout = [torch.randn(4, 128, generator=g).cuda() for _ in range(5)]
    results = {}
    for grouped in (True, False):
        ops.GROUP_COND_PROJ = grouped
        st = TrainState.create(apply_fn=None, variables={"params": tree}, device="cuda")
        assert len(st._supers) == 2 and sorted(w.shape[0] for w, _, _, _ in st._supers) == [2, 3]
        c = cond.clone().requires_grad_(True)
        st.zero_grad()
        outs, k = [], 0
        for mod, n in (("score_model", 3), ("encoder_model", 2)):
            for i in range(n):
                if (mod, i) == ("score_model", 1):
                    continue                              # an unused block: its gradient must come out as zero
                outs.append((ops.cond_proj(c, st.params[mod][f"down.block_{i}"]["cond_proj"]["kernel"]), gout[k]))
                k += 1
        loss = sum((o * go).sum() for o, go in outs)
        loss.backward()
        st.collect_grads()
        results[grouped] = ([o.detach().clone() for o, _ in outs], c.grad.clone(), st.grad.clone())
    ops.GROUP_COND_PROJ = True
    for a, b in zip(results[True][0], results[False][0]):        # (the per-block path sums K = 512 in 8 split-K pieces)
        assert float((a - b).abs().max()) < 1e-5 * float(b.abs().max())
    assert float((results[True][1] - results[False][1]).abs().max()) < 1e-5 * float(results[False][1].abs().max())
    assert float((results[True][2] - results[False][2]).abs().max()) < 1e-5 * float(results[False][2].abs().max())
    assert float(results[True][2].abs().max()) > 0


def test_arithmetic_modes_agree_over_training_steps():
    """Five optimiser steps from the same initialisation, data and noise in the three convolution modes (f16x3 with
    the plane hand-over, grouped FiLM projections and once-per-step weight preparation; bf16x6; exact-fp32 MFMA):
    the BPD trajectories agree to fp32 noise, i.e. far inside the +-0.005 BPD bar."""
    import os
    from mulan_amd import ops
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    saved = ops.CONV_MODE
    traj = {}
    try:
        for mode in ("f32", "bf16x6", "f16x3"):
            ops.CONV_MODE = mode
            config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
            config.data.dataset = 'synthetic'
            config.model.sm_n_layer = 2
            config.model.forward_n_layer = 1
            config.training.batch_size_train = 8
            config.training.batch_size_eval = 8
            config.training.substeps = 1
            config.training.num_steps_lr_warmup = 1
            exp = Experiment_VDM(config)
            sub = {k: v[0] for k, v in next(exp.train_iter).items()}
            losses = []
            for _ in range(5):
                exp.state, m = exp.train_step(exp._train_rng.fold_in(0), exp.state, sub)
                losses.append(float(m['scalars']['train_bpd']))
            traj[mode] = np.array(losses)
    finally:
        ops.CONV_MODE = saved
    assert np.all(np.isfinite(traj["f32"]))
    print("BPD trajectories:", {k: [round(float(x), 5) for x in v] for k, v in traj.items()})
    for mode in ("bf16x6", "f16x3"):
        assert np.abs(traj[mode] - traj["f32"]).max() < 5e-4, (mode, traj[mode], traj["f32"])   # measured: 1e-5


def test_training_learns_a_small_structured_set():
    """end-to-end learning check: 400 optimiser steps (dropout on, lr warm-up, AdamW + EMA) on a batch of smooth
    synthetic images must cut the training BPD substantially, and the EMA parameters must follow"""
    import os
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = 'synthetic'
    config.model.sm_n_layer = 2
    config.model.forward_n_layer = 1
    config.training.batch_size_train = 16
    config.training.batch_size_eval = 16
    config.training.substeps = 1
    config.training.num_steps_lr_warmup = 20
    config.optimizer.ema_rate = 0.9
    exp = Experiment_VDM(config)
    yy, xx = torch.meshgrid(torch.arange(32.0), torch.arange(32.0), indexing="ij")
    imgs = []
    for k in range(16):     # smooth colour ramps and blobs: a few bits per dimension at most
        r = 128 + 100 * torch.sin((xx * (1 + k % 3) + yy * (k % 5)) / 10.0 + k)
        g = 128 + 100 * torch.cos((yy * (1 + k % 4)) / 9.0 - k)
        b = 255 * torch.exp(-((xx - 16 - k % 7) ** 2 + (yy - 12) ** 2) / 80.0)
        imgs.append(torch.stack([r, g, b], dim=-1))
    images = torch.stack(imgs).clamp(0, 255).round().to(torch.uint8).cuda()
    sub = {"images": images, "labels": torch.zeros(16, dtype=torch.int32).cuda(),
           "conditioning": torch.zeros(16, dtype=torch.uint8).cuda()}
    hist = []
    for i in range(400):
        exp.state, m = exp.train_step(exp._train_rng, exp.state, sub)
        hist.append(float(m['scalars']['train_bpd']))
    first, last = float(np.mean(hist[:5])), float(np.mean(hist[-10:]))
    print(f"learning check: train bpd {first:.2f} -> {last:.2f} over {len(hist)} steps")
    assert all(np.isfinite(hist)) and last < 0.6 * first, (first, last)
    ev = float(exp.eval_step(exp._eval_rng, exp.state.ema_params, sub, 0)['scalars']['eval_bpd'])
    assert np.isfinite(ev) and ev < 0.8 * first, (first, last, ev)


@pytest.mark.parametrize("unet_type", ["vdm", "ldm"])
def test_module_surface_matches_oracle(unet_type):
    """The reference-shaped module handles of SURVEY 8(b) -- ScoreUNet / UNet (ldm/model_vdm.py:314, ldm/ldm_unet.py:69),
    UnetEncoder (ldm/model_mulan_epsilon.py:105), NoiseSchedule_polynomial_fixedend (:602, grad_t :540), EncDec.encode --
    called with the reference's argument lists on NHWC tensors, against the float64 oracle."""
    import ldm.ldm_unet
    import ldm.model_mulan_epsilon as me
    import ldm.model_vdm as mv
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg("mulan_velocity", unet_type)
    B, E = 3, 128
    ref = tr.init_params(ocfg, seed=21, dtype=torch.float64)
    vdm = M.make_vdm("mulan_velocity", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref), params)
    rng = np.random.default_rng(2)
    z = torch.tensor(rng.standard_normal((B, 32, 32, 3)))
    emb = torch.tensor((rng.random((B, 50)) < 0.3).astype(np.float64))
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    with torch.no_grad():
        if unet_type == "vdm":
            g = torch.tensor(rng.uniform(-10, 3, B))
            net = mv.ScoreUNet(cfg).apply(params["score_model"], z.float().cuda(), g.float().cuda(), emb.float().cuda(),
                                          deterministic=True)
            want = tr.score_unet(z, g, emb, ref["score_model"], E, 1)
        else:
            g = torch.tensor(rng.uniform(-10, 3, (B, 32, 32, 3)))
            net = ldm.ldm_unet.UNet(cfg).apply(params["score_model"], z.float().cuda(), g.float().cuda(),
                                               emb.float().cuda(), deterministic=True)
            want = tr.score_unet(z, g, emb, ref["score_model"], E, 1, per_pixel=True)
        assert net.shape == z.shape and rel(net, want) < 2e-4
        x = torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8))
        f = mv.EncDec(cfg).encode(x.cuda())
        assert torch.equal(f.cpu().double(), tr.encode(x.double()))
        # EncDec.decode / logprob / __call__ (ldm/model_vdm.py:269-303) with per-element, per-sample and scalar g_0
        ed = mv.EncDec(cfg)
        zz = torch.tensor(rng.uniform(-1.3, 1.3, (B, 32, 32, 3)))
        for g0 in (torch.tensor(rng.uniform(-13.5, -6.0, (B, 32, 32, 3))), torch.tensor(rng.uniform(-13.5, -6.0, B)), -9.7):
            g_np = g0.numpy() if torch.is_tensor(g0) else g0
            g_bc = g_np[:, None, None, None] * np.ones((1, 32, 32, 3)) if np.ndim(g_np) == 1 else g_np
            want_lp = onp.decode_logprobs(zz.float().double().numpy(), np.float32(g_bc).astype(np.float64))
            g_dev = g0.float().cuda() if torch.is_tensor(g0) else g0
            got_lp = ed.decode(zz.float().cuda(), g_dev)
            assert got_lp.shape == (B, 32, 32, 3, 256)
            err = np.abs(got_lp.cpu().double().numpy() - want_lp)
            # fp32 like the reference: each logit -0.5 u^2 (down to -2e6) carries ~3e-6 of relative error, and the
            # log-sum-exp of a row inherits the absolute error of its largest logits
            bad = err - (1e-4 + 1e-5 * np.abs(want_lp).max(axis=-1, keepdims=True))
            assert float(bad.max()) < 0, (float(err.max()), float(want_lp.reshape(-1)[bad.argmax()]),
                                          float(got_lp.cpu().reshape(-1)[bad.argmax()]))
            assert float(np.abs(np.exp(got_lp.cpu().double().numpy()).sum(-1) - 1).max()) < 1e-5
            want = onp.logprob(x.numpy(), zz.float().double().numpy(), np.float32(g_bc).astype(np.float64))
            got = ed.logprob(x.cuda(), zz.float().cuda(), g_dev)
            assert got.shape == (B,) and rel(got, torch.tensor(want)) < 2e-5, (got, want)
        g1 = torch.tensor(rng.uniform(-13.5, -6.0, B))
        assert torch.equal(ed(x.cuda(), g1.float().cuda()), ed.decode(ed.encode(x.cuda()), g1.float().cuda()))
        logits = me.UnetEncoder(cfg)(params["encoder_model"], f, deterministic=True)
        assert rel(logits, tr.unet_encoder(tr.encode(x.double()), ref["encoder_model"], E, 1)) < 2e-4
        sched = me.NoiseSchedule_polynomial_fixedend(cfg)
        t = torch.tensor(rng.random(B))
        a, b, c = tr.poly_coefficients(emb, ref["gamma"])
        assert rel(sched(params["gamma"], emb.float().cuda(), t.float().cuda()), tr.poly_gamma(a, b, c, t)) < 1e-5
        assert rel(sched.grad_t(params["gamma"], emb.float().cuda(), t.float().cuda()),
                   tr.poly_gamma_grad_t(a, b, c, t)) < 1e-5
        assert rel(sched(params["gamma"], emb.float().cuda(), 0.0), torch.full((B, 3072), -13.3, dtype=torch.float64)) < 1e-6
    with pytest.raises(ValueError):
        mv.ScoreUNet(cfg).apply(params["score_model"], z.float().cuda(), 0.0, emb.float().cuda(), deterministic=False)


def test_full_depth_train_mode_gradient_parity():
    """the shipped depth (32 + 2 + 33 ResnetBlocks, 4-layer encoder) in TRAINING mode (dropout on) at B = 2: loss terms
    and every parameter gradient against float64 autograd.  ~140 chained split-operand convolutions forward and
    backward: fp32-level noise grows with depth, so the per-leaf bar is 5x the single-layer one (1e-2 of the leaf's
    gradient scale); the BPD bar stays +-0.005 absolute.  (The float64 backward pass through this depth takes 40-100 s on
    the host cores: it runs on the device; the host-pinned comparisons are test_full_depth_forward_bpd_parity and
    test_deeper_stack_train_gradients, tests/oracle_dev.py.)"""
    run_case("mulan_velocity", "vdm", False, train=True, n_layer=32, fwd_layers=4, B=2, tol=5.0)


def test_graph_replayed_train_steps_equal_eager_steps():
    """GraphedStep (HIP-graph replay of the train step, the build's lax.scan, ldm/experiment.py:89-91) against the
    eager step: same batches, same keys -> the parameters, the EMA, the Adam moments and the logged scalars after 4 steps
    are bit-identical (the replay launches the very same kernels; everything step-dependent -- batch, noise, t0,
    dropout seeds, learning rate, bias corrections -- reaches them through device memory); the by-product hand-overs
    between autograd nodes fire under capture as they do eagerly (maxima passes per step stay <= 22)."""
    import os
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd import lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(graph):
        config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
        config.model.sm_n_layer = 2
        config.model.forward_n_layer = 1
        config.data.dataset = "synthetic"
        config.training.batch_size_train = 4
        config.training.batch_size_eval = 4
        config.training.substeps = 1
        config.training.hip_graph = graph
        exp = Experiment_VDM(config)
        g = torch.Generator().manual_seed(3)
        scal = []
        for i in range(4):
            batch = {"images": torch.randint(0, 256, (4, 32, 32, 3), generator=g, dtype=torch.uint8).cuda(),
                     "labels": torch.zeros(4, dtype=torch.int32).cuda(),
                     "conditioning": torch.zeros(4, dtype=torch.uint8).cuda()}
            _, m = exp.train_step(exp._train_rng, exp.state, batch)
            scal.append({k: float(v) for k, v in m["scalars"].items()})
        torch.cuda.synchronize()
        st = exp.state
        return (st.flat.clone(), st.ema.clone(), st.mu.clone(), st.nu.clone(), st.step, scal,
                exp._graphed is not None)

    calls = []
    orig = lib.call

    def counting(name, *a):
        calls.append(name)
        return orig(name, *a)
    e = run(False)
    g = run(True)
    assert not e[6] and g[6] and e[4] == g[4] == 4
    for a, b, name in zip(e[:4], g[:4], ("params", "ema", "mu", "nu")):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    assert e[5] == g[5]
    assert len({s["train_bpd"] for s in g[5]}) == 4               # four different steps, not one replayed result


def test_by_product_hand_overs_fire_in_a_full_depth_step(monkeypatch):
    """The f16x3 kernels hand their by-products on (output maxima from GroupNorm / convolution epilogues, split planes,
    channel sums): a separate mulan_absmax_rows pass is only needed where no producer kernel exists.  A silent miss
    (e.g. after a torch upgrade changed tensor versioning) would cost an extra pass per layer: count the launches of one
    full-depth (32 + 2 + 33 blocks) train step."""
    import os
    from collections import Counter
    from mulan_amd import ops as _ops
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = "synthetic"
    config.training.batch_size_train = 2
    config.training.batch_size_eval = 2
    config.training.substeps = 1
    config.training.hip_graph = False
    exp = Experiment_VDM(config)
    batch = {"images": torch.randint(0, 256, (2, 32, 32, 3), dtype=torch.uint8).cuda(),
             "labels": torch.zeros(2, dtype=torch.int32).cuda(), "conditioning": torch.zeros(2, dtype=torch.uint8).cuda()}
    exp.train_step(exp._train_rng, exp.state, batch)            # warm-up
    counts = Counter()
    orig = _ops.call

    def counting(name, *a):
        counts[name] += 1
        return orig(name, *a)
    monkeypatch.setattr(_ops, "call", counting)
    exp.train_step(exp._train_rng, exp.state, batch)
    torch.cuda.synchronize()
    print("launches per step:", sum(counts.values()), dict(counts.most_common(12)))
    # 2 attention blocks x (q, k, v, dO, delta) + conv_in / conv_out / loss-side tensors: no per-ResnetBlock passes
    assert counts["mulan_absmax_rows"] <= 24, counts["mulan_absmax_rows"]
    assert counts["mulan_conv3x3_pack_f16x3"] == 0 and counts["mulan_linear_pack_f16x3"] == 0   # ParamPacker did them all
    assert counts["mulan_param_pack_f16x3"] == 1 and counts["mulan_param_maxima"] == 1
    # plane-fed weight gradients (MULAN_FOLD_SLAB_REDUCE=1: the same launches through the _fold entry point)
    assert counts["mulan_conv3x3_wgrad_f16x3_planes"] + counts["mulan_conv3x3_wgrad_f16x3_planes_fold"] >= 2 * 67 + 2 * 6


@pytest.mark.timeout(900)
@pytest.mark.parametrize("config_file,batch,vdm_type,vfe", [
    ("cifar10-conditioned.py", 128, "mulan_epsilon", False),      # BASELINE config #2 = the bench workload, as worded
    ("cifar10-conditioned.py", 64, "mulan_velocity", False),      # config #3 at its per-GPU batch (512 / 8)
    ("cifar10-conditioned.py", 200, "mulan_velocity", False),     # 800 tiles per launch: a partial last round of blocks
    ("imagenet32.py", 128, "mulan_velocity", True)])              # config #4 per GPU: E = 256, velocity_from_epsilon
def test_full_size_step_is_repeatable_and_agrees_with_the_fp32_mfma_mode(config_file, batch, vdm_type, vfe):
    """BASELINE's training configurations exactly as it words them, at their full size (32 + 2 + 33 blocks, per-GPU
    batch -- far beyond what the float64 oracle can run): properties that do not
    need the oracle.  (a) Two fresh runs of
    two train steps end in bit-identical parameters, moments and gradients -- every kernel at its full launch size is
    free of races (this catches e.g. a block that reads its accumulators too early only when it shares a CU).  (b) The
    gradient of the same step with every convolution on the exact-fp32 MFMA kernels (an independent code path: other
    kernels, no operand split, no by-product hand-overs) agrees to fp32 noise; so does the loss."""
    import os
    from mulan_amd import ops
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B = int(os.environ.get("MULAN_TEST_FULL_B", batch))

    def run(mode, steps):
        saved = ops.CONV_MODE
        ops.CONV_MODE = mode
        try:
            config = load_config_file(os.path.join(root, "ldm", "configs", os.environ.get("MULAN_TEST_FULL_CONFIG", config_file)))
            config.data.dataset = "synthetic"
            config.vdm_type = vdm_type
            config.model.velocity_from_epsilon = vfe
            config.training.batch_size_train = B
            config.training.batch_size_eval = B
            config.training.substeps = 1
            config.training.hip_graph = False
            exp = Experiment_VDM(config)
            with torch.no_grad():   # un-zero the zero-initialised layers: every block contributes, every gradient is live
                exp.state.flat.add_(0.01 * torch.randn(exp.state.flat.shape, device="cuda",
                                                       generator=torch.Generator("cuda").manual_seed(0)))
                exp.state.ema.copy_(exp.state.flat)
            g = torch.Generator().manual_seed(17)
            bpd = []
            for _ in range(steps):
                batch = {"images": torch.randint(0, 256, (B, 32, 32, 3), generator=g, dtype=torch.uint8).cuda(),
                         "labels": torch.zeros(B, dtype=torch.int32).cuda(),
                         "conditioning": torch.zeros(B, dtype=torch.uint8).cuda()}
                _, m = exp.train_step(exp._train_rng, exp.state, batch)
                bpd.append(float(m["scalars"]["train_bpd"]))
            torch.cuda.synchronize()
            st = exp.state
            out = (st.flat.clone(), st.mu.clone(), st.nu.clone(), st.grad.clone(), bpd)
            del exp
            torch.cuda.empty_cache()
            return out
        finally:
            ops.CONV_MODE = saved

    a = run("f16x3", 2)
    b = run("f16x3", 2)
    for x, y, name in zip(a[:4], b[:4], ("params", "mu", "nu", "grad")):
        assert torch.equal(x, y), (name, int((x != y).sum()), float((x - y).abs().max()))
    assert a[4] == b[4]
    one = run("f16x3", 1)
    ref = run("f32", 1)
    g16, g32 = one[3].double(), ref[3].double()
    err = float((g16 - g32).norm() / g32.norm())
    print("full-size gradient, f16x3 vs exact-fp32 MFMA kernels: relative L2 error %.3e; bpd %r vs %r" % (err, one[4], ref[4]))
    assert err < 5e-4                                          # measured: 4e-5
    assert abs(one[4][0] - ref[4][0]) < 1e-4 * abs(ref[4][0])


def test_weight_gradient_stream_changes_nothing_but_the_schedule(monkeypatch):
    """ops.weight_gradient_stream (weight gradients on a second HIP stream beside the input-gradient chain): three
    train steps with and without it end in bit-identical parameters, moments and gradients -- the events that order the
    two streams and keep the operands alive are complete (SIDE_DEPTH = 1 makes the main stream wait early and often)."""
    import os
    from mulan_amd import ops
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(side, depth, share=False):
        monkeypatch.setattr(ops, "SIDE_STREAM", side)
        monkeypatch.setattr(ops, "SIDE_DEPTH", depth)
        monkeypatch.setattr(ops, "SIDE_WGRAD_SHARE", share)     # (True: other split counts, i.e. another summation order)
        config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
        config.model.sm_n_layer = 3
        config.model.forward_n_layer = 1
        config.data.dataset = "synthetic"
        config.training.batch_size_train = 16
        config.training.batch_size_eval = 16
        config.training.substeps = 1
        config.training.hip_graph = False
        exp = Experiment_VDM(config)
        with torch.no_grad():
            exp.state.flat.add_(0.01 * torch.randn(exp.state.flat.shape, device="cuda",
                                                   generator=torch.Generator("cuda").manual_seed(0)))
        g = torch.Generator().manual_seed(5)
        launches = []
        real = ops._on_side
        monkeypatch.setattr(ops, "_on_side", lambda fn, keep: (launches.append(1), real(fn, keep))[1])
        for _ in range(3):
            batch = {"images": torch.randint(0, 256, (16, 32, 32, 3), generator=g, dtype=torch.uint8).cuda(),
                     "labels": torch.zeros(16, dtype=torch.int32).cuda(),
                     "conditioning": torch.zeros(16, dtype=torch.uint8).cuda()}
            exp.train_step(exp._train_rng, exp.state, batch)
        torch.cuda.synchronize()
        monkeypatch.setattr(ops, "_on_side", real)
        st = exp.state
        return (st.flat.clone(), st.mu.clone(), st.nu.clone(), st.grad.clone()), len(launches)

    ref, n0 = run(False, 6)
    assert n0 == 0
    for depth in (6, 1):
        got, n = run(True, depth)
        assert n >= 3 * 2 * (3 + 2 + 4)                       # every ResnetBlock convolution of every step went there
        for a, b, name in zip(got, ref, ("params", "mu", "nu", "grad")):
            assert torch.equal(a, b), (depth, name, float((a - b).abs().max()))
    # with the shared-chip block count of the weight-gradient launches (half the pixel-range splits): the same sums in
    # another order
    got, _ = run(True, 6, share=True)
    assert float((got[3] - ref[3]).norm() / ref[3].norm()) < 1e-5
