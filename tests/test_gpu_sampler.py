"""Ancestral sampler: the HIP step / decode kernels and the whole T-step loop of
Experiment_VDM.sample_fn against the float64 oracle restatement (oracle/torch_ref.py) on the same noise."""
import dataclasses
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.oracle_dev import run_oracle
from oracle import torch_ref as tr
from tests.test_gpu_model import make_cfg


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("mode,kind", [(0, "velocity"), (1, "epsilon"), (2, "input")])
@pytest.mark.parametrize("per_sample", [False, True])
def test_ancestral_step_kernel(mode, kind, per_sample):
    from mulan_amd import ops
    rng = np.random.default_rng(3 + mode)
    B, D = 5, 3072
    zt, net, eps = (rng.standard_normal((B, D)).astype(np.float32) for _ in range(3))
    gshape = (B,) if per_sample else (B, D)
    gt = rng.uniform(-13.3, 5.0, gshape).astype(np.float32)
    gs = (gt - rng.uniform(1e-3, 0.5, gshape)).astype(np.float32)           # s < t: gamma_s < gamma_t
    dev = lambda a: torch.tensor(a).cuda()
    zs = ops.ancestral_step(dev(zt), dev(net), dev(gt), dev(gs), dev(eps), mode).cpu().numpy()
    bc = (lambda g: torch.tensor(g, dtype=torch.float64)[:, None]) if per_sample else \
        (lambda g: torch.tensor(g, dtype=torch.float64))
    ref = tr.ancestral_step(torch.tensor(zt, dtype=torch.float64), torch.tensor(net, dtype=torch.float64), bc(gt),
                            bc(gs), torch.tensor(eps, dtype=torch.float64), kind).numpy()
    # fp32 elementwise chain: a few ulp of the result scale ('input' divides by sigma_t ~ 1e-3 at gamma_min)
    assert _rel(zs, ref) < (2e-5 if mode == 2 else 2e-6)


def test_ancestral_step_last_step_is_deterministic_limit():
    """at s with gamma_s -> gamma_t the step is the identity on z_t and adds no noise (c = 0)"""
    from mulan_amd import ops
    z = torch.randn(2, 3072, device="cuda")
    g = torch.full((2, 3072), -3.0, device="cuda")
    out = ops.ancestral_step(z, torch.randn_like(z), g, g.clone(), torch.randn_like(z), 1)
    assert torch.equal(out, z)


@pytest.mark.parametrize("per_sample", [False, True])
def test_decode_argmax_kernel(per_sample):
    from mulan_amd import ops
    rng = np.random.default_rng(11)
    B, D = 4, 3072
    x = rng.integers(0, 256, (B, D))
    x[0, :256] = np.arange(256)                                             # every bin once
    g0 = rng.uniform(-13.3, -8.0, (B,) if per_sample else (B, D)).astype(np.float32)
    g0b = g0[:, None] if per_sample else g0
    v = 2 * ((x + 0.5) / 256) - 1
    # z_0 strictly inside bin x after the 1 / alpha_0 rescale: the argmax must be x exactly
    z0 = ((v + rng.uniform(-0.4, 0.4, (B, D)) * (2 / 256)) * np.sqrt(1 - 1 / (1 + np.exp(-g0b.astype(np.float64)))))
    z0 = z0.astype(np.float32)
    out = ops.decode_argmax(torch.tensor(z0).cuda(), torch.tensor(g0).cuda()).cpu().numpy()
    assert out.dtype == np.uint8 and np.array_equal(out, x)
    ref = tr.decode_argmax(torch.tensor(z0, dtype=torch.float64), torch.tensor(g0b, dtype=torch.float64) *
                           torch.ones(B, D, dtype=torch.float64)).numpy()
    assert np.array_equal(out, ref)
    # out-of-range latents clamp to the end bins
    far = torch.tensor([[-7.0, 7.0, -1.0001, 1.0001] * 768] * B).cuda()
    got = ops.decode_argmax(far, torch.tensor(g0).cuda()).cpu().numpy()
    assert np.array_equal(got[0, :4], [0, 255, 0, 255])


def test_decode_argmax_random_latents_tie_tolerant():
    from mulan_amd import ops
    rng = np.random.default_rng(12)
    B, D = 8, 3072
    z0 = rng.standard_normal((B, D)).astype(np.float32) * 0.6
    g0 = rng.uniform(-13.3, -6.0, (B, D)).astype(np.float32)
    out = ops.decode_argmax(torch.tensor(z0).cuda(), torch.tensor(g0).cuda()).cpu().numpy().astype(np.int64)
    ref = tr.decode_argmax(torch.tensor(z0, dtype=torch.float64), torch.tensor(g0, dtype=torch.float64)).numpy()
    bad = out != ref
    # a disagreement is only legal where the rescaled latent sits on a bin edge to fp32 precision
    zr = z0.astype(np.float64) / np.sqrt(1 - 1 / (1 + np.exp(-g0.astype(np.float64))))
    edge = np.abs(((zr + 1) * 128) - np.round((zr + 1) * 128))
    assert np.all(np.abs(out - ref)[bad] == 1) and np.all(edge[bad] < 1e-4) and bad.mean() < 1e-3


def test_rowmean_kernel():
    from mulan_amd import ops
    x = torch.randn(7, 3072, device="cuda") * 3 + 1
    assert _rel(ops.rowmean(x).cpu().numpy(), x.double().mean(dim=1).cpu().numpy()) < 1e-6


def _damp(ref_params, k):
    """the free-running comparison needs a contractive network: with random conv_out the map z_t -> z_s of a
    random-init U-Net amplifies an fp32-level difference by orders of magnitude per step (measured: 1e-3 -> 0.5 over
    3 steps), which says nothing about parity.  Scaling conv_out leaves eps_hat = z_t + k * (random U-Net)."""
    ref_params["score_model"]["conv_out"]["kernel"] = ref_params["score_model"]["conv_out"]["kernel"] * k
    return ref_params


def _mulan_setup(vdm_type, unet_type, damp=None):
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg(vdm_type, unet_type)
    ocfg = dict(ocfg, latent_size=50)
    ref_params = tr.init_params(ocfg, seed=5, dtype=torch.float64)
    if damp is not None:
        _damp(ref_params, damp)
    vdm = M.make_vdm(vdm_type, cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    return vdm, params, ref_params, ocfg


def _check_steps(vdm, params, key, z_init, T, oracle_loop):
    """per step: the product's reverse step from the oracle's z_t against the oracle's z_s, with the error budget
    (network-output bar of tests/test_gpu_model.py, 2e-4 of max|net|) x (d z_s / d net of that step, which the
    oracle reports: up to ~10 for the first of T = 3 steps, gamma 5 -> -1); generate_x from the oracle's z_0 exactly
    up to bin-edge ties"""
    B = z_init.shape[0]
    eps = [key.fold_in(i).normal((B, 3072), "cuda").cpu().double() for i in range(T)]
    cond = torch.zeros(B, dtype=torch.uint8, device="cuda")
    z_ref, x_ref, traj, budget = oracle_loop(z_init.cpu().double(), eps)
    for i in range(T):
        zs = vdm.sample(params, i, T, traj[i].reshape(B, -1).float().cuda(), cond, key)
        err = np.abs(zs.cpu().double().numpy() - traj[i + 1].reshape(B, -1).numpy()).max()
        assert err < 2e-4 * budget[i] + 2e-6 * float(traj[i + 1].abs().max()), (i, err, budget[i])
    x = vdm.generate_x(params, z_ref.reshape(B, -1).float().cuda())
    assert x.shape == (B, 32, 32, 3) and x.dtype == torch.uint8
    d = np.abs(x.cpu().numpy().astype(np.int64) - x_ref.numpy())
    assert d.max() <= 1 and (d != 0).mean() < 2e-3                      # fp32 rounding of z_0 at a bin edge only


def _check_free_run(vdm, params, key, z_init, T, oracle_loop):
    """the whole loop from z_T on a damped network (see _damp): trajectory end point and the decoded image"""
    B = z_init.shape[0]
    eps = [key.fold_in(i).normal((B, 3072), "cuda").cpu().double() for i in range(T)]
    cond = torch.zeros(B, dtype=torch.uint8, device="cuda")
    z_ref, x_ref, traj, budget = oracle_loop(z_init.cpu().double(), eps)
    z = z_init
    for i in range(T):
        z = vdm.sample(params, i, T, z, cond, key)
    free = np.abs(z.cpu().double().numpy() - z_ref.reshape(B, -1).numpy()).max()
    assert free < 2e-4 * sum(budget) * 4 + 1e-5, (free, budget)         # per-step errors carried through later steps
    d = np.abs(vdm.generate_x(params, z).cpu().numpy().astype(np.int64) - x_ref.numpy())
    assert d.max() <= 1, d.max()
    return z, cond


@pytest.mark.parametrize("vdm_type,unet_type", [("mulan_velocity", "vdm"), ("mulan_epsilon", "vdm"),
                                                ("mulan_velocity", "ldm")])
def test_sampler_loop_matches_oracle(vdm_type, unet_type):
    """T = 3 reverse steps + generate_x through model.sample (the per-step noise is the product's own Philox draw,
    handed to the oracle as data)"""
    from mulan_amd.rng import PRNGKey
    B, T = 2, 3
    key = PRNGKey(21)
    z_init = key.fold_in(1000).normal((B, 3072), "cuda")
    vdm, params, ref_params, ocfg = _mulan_setup(vdm_type, unet_type)
    _check_steps(vdm, params, key, z_init, T,
                 lambda zi, eps: run_oracle(lambda P, z_, e_: tr.mulan_sample_loop(P, ocfg, z_, e_, trajectory=True), ref_params, zi, eps))
    vdm, params, ref_params, ocfg = _mulan_setup(vdm_type, unet_type, damp=0.02)
    z, cond = _check_free_run(vdm, params, key, z_init, T,
                              lambda zi, eps: run_oracle(lambda P, z_, e_: tr.mulan_sample_loop(P, ocfg, z_, e_, trajectory=True), ref_params, zi, eps))
    # precomputed schedule coefficients (what sample_fn passes) give the identical trajectory
    coeffs = vdm.sample_coefficients(params, vdm.deterministic_embedding(B, "cuda"))
    z2 = z_init
    for i in range(T):
        z2 = vdm.sample(params, i, T, z2, cond, key, coeffs)
    assert torch.equal(z2, z)


@pytest.mark.parametrize("gamma_type,reparam", [("fixed", "noise"), ("learnable_scalar", "noise"), ("fixed", "input")])
def test_plain_vdm_sampler_loop_matches_oracle(gamma_type, reparam):
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg()
    cfg = dataclasses.replace(cfg, gamma_type=gamma_type, z_conditioning=False, reparam_type=reparam)
    ocfg = dict(ocfg, reparam_type=reparam)
    B, T = 2, 3
    key = PRNGKey(5)
    z_init = key.fold_in(77).normal((B, 3072), "cuda")
    vdm = M.make_vdm("vdm", cfg)
    for damp, check in ((None, _check_steps), (0.02, _check_free_run)):
        full = tr.init_params(ocfg, seed=4, dtype=torch.float64)
        ref_params = {"score_model": full["score_model"]}
        ref_params["score_model"]["dense0"]["kernel"] = ref_params["score_model"]["dense0"]["kernel"][:129].clone()
        if damp is not None:
            _damp(ref_params, damp)
        if gamma_type == "learnable_scalar":
            ref_params["gamma"] = {"w": torch.tensor([-17.0], dtype=torch.float64),
                                   "b": torch.tensor([-12.5], dtype=torch.float64)}
        params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
        M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
        check(vdm, params, key, z_init, T,
              lambda zi, eps: run_oracle(lambda P, z_, e_: tr.plain_sample_loop(P, ocfg, z_, e_, trajectory=True), ref_params, zi, eps))


def test_sample_softmax_draws_from_the_decoder_distribution():
    """sample_softmax=True: generate_x draws from softmax(logits) (jax.random.categorical): empirical bin frequencies
    of 200 000 draws at one latent value against the oracle's decoder probabilities; needs the 'sample' key"""
    from mulan_amd import model as M, ops
    from mulan_amd.rng import PRNGKey
    cfg, _ = make_cfg()
    vdm = M.make_vdm("mulan_velocity", dataclasses.replace(cfg, sample_softmax=True))
    with pytest.raises(ValueError):
        vdm.generate_x({}, torch.zeros(1, 3072, device="cuda"))
    n = 200_000
    g0 = -4.0                                  # sigma_0 = 0.13: about 40 bins carry mass
    z = 0.3137
    out = ops.decode_sample(torch.full((n,), z, device="cuda"), torch.full((n,), g0, device="cuda"), PRNGKey(4).v)
    freq = np.bincount(out.cpu().numpy(), minlength=256) / n
    zz = torch.tensor([z], dtype=torch.float64) / torch.sqrt(torch.sigmoid(torch.tensor(-g0, dtype=torch.float64)))
    vals = tr.encode(torch.arange(256, dtype=torch.float64))
    probs = torch.softmax(-0.5 * ((zz - vals) * math.exp(-0.5 * g0)) ** 2, dim=0).numpy()
    assert 0.5 * np.abs(freq - probs).sum() < 0.01                    # total variation
    assert abs((freq * np.arange(256)).sum() - (probs * np.arange(256)).sum()) < 0.1
    # different keys give different draws, the same key the same draws; sample=0 path unchanged
    a = ops.decode_sample(torch.full((64,), z, device="cuda"), torch.full((64,), g0, device="cuda"), 1)
    b = ops.decode_sample(torch.full((64,), z, device="cuda"), torch.full((64,), g0, device="cuda"), 1)
    c = ops.decode_sample(torch.full((64,), z, device="cuda"), torch.full((64,), g0, device="cuda"), 2)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_experiment_sample_fn():
    """Experiment_VDM.sample_fn on the EMA tree: uint8 [B,32,32,3], reproducible for a given key, a different key
    gives different samples, and the training parameters / packer state are left untouched"""
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd.rng import PRNGKey
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = 'synthetic'
    config.model.sm_n_layer = 1
    config.model.forward_n_layer = 1
    config.training.batch_size_train = 4
    config.training.batch_size_eval = 4
    exp = Experiment_VDM(config)
    dummy = torch.zeros(3, 32, 32, 3, dtype=torch.uint8, device="cuda")
    before = exp.state.flat.clone()
    a = exp.sample_fn(dummy_inputs=dummy, rng=PRNGKey(1), params=exp.state.ema_params, T=4)
    b = exp.sample_fn(dummy_inputs=dummy, rng=PRNGKey(1), params=exp.state.ema_params, T=4)
    c = exp.sample_fn(dummy_inputs=dummy, rng=PRNGKey(2), params=exp.state.ema_params, T=4)
    assert a.shape == (3, 32, 32, 3) and a.dtype == torch.uint8
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(before, exp.state.flat)


@pytest.mark.parametrize("vdm_type,unet_type", [("mulan_velocity", "vdm"), ("mulan_epsilon", "ldm")])
def test_replayed_reverse_step_equals_the_eager_step(vdm_type, unet_type, monkeypatch):
    """model.GraphedReverseStep (the sampler's reverse step as a replayed HIP graph: static buffers for z_t, the step's
    noise and the two times) against conditional_sample, the eager step (ldm/model_mulan_velocity.py:281-350): the latent
    after every one of six steps and the generated images are bit-identical; Experiment_VDM.sample_fn gives the same
    images with the replay on and off."""
    from mulan_amd import model as M
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd.rng import PRNGKey
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
    config.vdm_type = vdm_type
    config.model.unet_type = unet_type
    config.data.dataset = 'synthetic'
    config.model.sm_n_layer = 2
    config.model.forward_n_layer = 1
    config.training.batch_size_train = 4
    config.training.batch_size_eval = 4
    exp = Experiment_VDM(config)
    gen = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():          # (zero-initialised layers would make the network output trivial)
        exp.state.ema.copy_(torch.randn(exp.state.ema.shape, device="cuda", generator=gen) * 0.03)
    model, params, B, T = exp.model, exp.state.ema_params, 5, 6
    rng = PRNGKey(11)
    packer = exp.state.param_packer("ema")
    with torch.no_grad():
        if packer is not None:
            packer.refresh()
        emb = model.deterministic_embedding(B, exp.device)
        cond = torch.zeros(B, dtype=torch.uint8, device="cuda")
        coeffs = model.sample_coefficients(params, emb)
        z0 = rng.normal((B, 3072), exp.device)
        eager = model.reverse_stepper(params, B, exp.device, emb, cond, coeffs, T, graph=False)
        replay = model.reverse_stepper(params, B, exp.device, emb, cond, coeffs, T, graph=True)
        assert type(getattr(replay, "__self__", None)).__name__ == "GraphedReverseStep"
        za, zb = z0.clone(), z0.clone()
        for i in range(T):
            za = eager(i, za, rng)
            zb = replay(i, zb, rng).clone()
            assert torch.equal(za, zb), (i, float((za - zb).abs().max()))
        assert bool(torch.isfinite(za).all()) and float(za.std()) > 0
        xa, xb = model.generate_x(params, za, coeffs), model.generate_x(params, zb, coeffs)
        assert torch.equal(xa, xb)
        if packer is not None:
            packer.invalidate()
    dummy = torch.zeros(3, 32, 32, 3, dtype=torch.uint8, device="cuda")
    monkeypatch.setattr(M, "SAMPLER_GRAPH", True)
    a = exp.sample_fn(dummy_inputs=dummy, rng=PRNGKey(1), params=exp.state.ema_params, T=4)
    monkeypatch.setattr(M, "SAMPLER_GRAPH", False)
    b = exp.sample_fn(dummy_inputs=dummy, rng=PRNGKey(1), params=exp.state.ema_params, T=4)
    assert torch.equal(a, b) and a.dtype == torch.uint8


def test_kernels_against_golden_fixture():
    """the committed fixture tests/golden/sampler_ode.npz (oracle outputs on seeded inputs) as a file-based target"""
    import os
    from mulan_amd import ops
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_ode.npz"))
    dev = lambda k: torch.tensor(z[k], dtype=torch.float32).cuda().view(1, -1)
    for mode, kind in ((0, "velocity"), (1, "epsilon"), (2, "input")):
        got = ops.ancestral_step(dev("z"), dev("net"), dev("g_t"), dev("g_s"), dev("eps"), mode).cpu().numpy()[0]
        assert _rel(got, z[f"step_{kind}"]) < (5e-4 if mode == 2 else 2e-6), kind     # 'input' divides by sigma ~ 1e-3
    for mode, kind in ((0, "velocity"), (1, "vfe"), (2, "epsilon")):
        got, _ = ops.ode_drift(dev("net"), dev("z"), dev("g_t"), dev("g_p"), None, mode)
        assert _rel(got.cpu().numpy()[0], z[f"drift_{kind}"]) < 5e-6, kind
    dec = ops.decode_argmax(dev("z0"), dev("g0")).cpu().numpy()[0]
    assert np.abs(dec.astype(np.int64) - z["decoded"]).max() <= 1 and (dec != z["decoded"]).mean() < 0.01
    emb, kl = ops.topk_hard(torch.tensor(z["logits"], dtype=torch.float32).cuda(), 15)
    assert np.array_equal(emb.cpu().numpy(), z["hard_topk"]) and _rel(kl.cpu().numpy(), z["kl"]) < 1e-5


def test_colab_front_end_samplers(tmp_path):
    """Experiment_Colab.sample_conditionally / sample_randomly / test (ldm/notebook_utils.py:54-154) on a checkpoint of
    a two-step training run: image grids of the right shape, reproducible, different for different embeddings"""
    import os
    from mulan_amd import checkpoint as ck
    from mulan_amd.config import load_config_file
    from mulan_amd.evaluators import Experiment_Colab
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd import data as dataset
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def cfg():
        c = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
        c.data.dataset = 'synthetic'
        c.model.sm_n_layer = 1
        c.model.forward_n_layer = 1
        c.training.batch_size_train = 4
        c.training.batch_size_eval = 4
        c.training.substeps = 1
        return c
    exp = Experiment_VDM(cfg())
    ck.save(str(tmp_path), exp.state.state_dict())
    colab = Experiment_Colab(cfg(), str(tmp_path))
    e1 = np.zeros(50, dtype=np.float32); e1[:15] = 1
    e2 = np.zeros(50, dtype=np.float32); e2[20:35] = 1
    a = colab.sample_conditionally(e1, T=3)
    b = colab.sample_conditionally(e1, T=3)
    c = colab.sample_conditionally(e2, T=3)
    assert a.shape == (64, 64, 3) and a.dtype == np.uint8 and np.array_equal(a, b)
    r = colab.sample_randomly(T=3)
    assert r.shape == (64, 64, 3)
    m = colab.test([colab.eval_iter.next() for _ in range(2)])
    assert np.isfinite(m["eval_bpd"])
    del c
