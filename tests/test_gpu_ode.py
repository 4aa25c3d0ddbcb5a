"""Exact-likelihood ODE evaluator: device-resident RK45 against scipy.integrate.solve_ivp (the
solver the reference calls), the drift / divergence / noise / dequantisation kernels and the whole likelihood of a
small model against the float64 oracle (oracle/torch_ref.py: reverse_ode, value_div, ode_likelihood)."""
import dataclasses
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.oracle_dev import on_device, params_on_device
from oracle import torch_ref as tr
from tests.test_gpu_model import make_cfg


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ------------------------------------------------------------------------------------------------ integrator
@pytest.mark.parametrize("tol", [1e-3, 1e-5, 1e-8])
def test_rk45_takes_scipys_steps(tol):
    """y' = a y + b t with per-component a, b, evaluated in fp32 from the fp32-rounded state on both sides (mul and
    add are exactly rounded on CPU and GPU): the device integrator must take the very steps scipy's RK45 takes"""
    from scipy import integrate
    from mulan_amd.ode import solve_rk45
    rng = np.random.default_rng(0)
    n = 5000
    a = rng.uniform(-6.0, 1.5, n).astype(np.float32)
    b = rng.uniform(-3.0, 3.0, n).astype(np.float32)
    y0 = rng.standard_normal(n)
    calls = {"cpu": 0, "gpu": 0}

    def f_cpu(t, y):
        calls["cpu"] += 1
        y32 = torch.tensor(y, dtype=torch.float64).float()
        return (torch.tensor(a) * y32 + torch.tensor(b) * torch.tensor(np.float32(t))).double().numpy()

    ad, bd = torch.tensor(a).cuda(), torch.tensor(b).cuda()

    def f_gpu(t, y32, out):
        calls["gpu"] += 1
        out.copy_(ad * y32 + bd * torch.tensor(np.float32(t), device="cuda"))

    ref = integrate.solve_ivp(f_cpu, (0, 1), y0, rtol=tol, atol=tol, method="RK45")
    sol = solve_rk45(f_gpu, torch.tensor(y0).cuda(), (0.0, 1.0), rtol=tol, atol=tol)
    assert ref.status == 0 and sol.t == 1.0
    assert sol.nfev == ref.nfev == calls["gpu"] == calls["cpu"]
    assert sol.steps == len(ref.t) - 1
    # same steps; the states agree to fp32 resolution (f sees the fp32-rounded state, so a 1e-16 difference in the
    # float64 error-norm summation order can flip one fp32 rounding of y and move f by an fp32 ulp)
    assert _rel(sol.y.cpu().numpy(), ref.y[:, -1]) < 5e-7


def test_rk45_rejects_and_recovers():
    """a stiff-ish component forces rejected steps; the accepted sequence still matches scipy"""
    from scipy import integrate
    from mulan_amd.ode import solve_rk45
    a = np.array([-400.0, -1.0, 0.5, -50.0] * 64, dtype=np.float32)
    y0 = np.ones(len(a))
    ad = torch.tensor(a).cuda()
    ref = integrate.solve_ivp(lambda t, y: (torch.tensor(a) * torch.tensor(y).float()).double().numpy(), (0, 1), y0,
                              rtol=1e-4, atol=1e-6, method="RK45")
    sol = solve_rk45(lambda t, y32, out: out.copy_(ad * y32), torch.tensor(y0).cuda(), (0.0, 1.0), rtol=1e-4, atol=1e-6)
    assert sol.nfev == ref.nfev and sol.rejected > 0
    assert np.allclose(sol.y.cpu().numpy(), ref.y[:, -1], rtol=1e-4, atol=1e-9)


def test_rk45_backward_in_time():
    """(1, 0) integration, the direction of the ODE sampler"""
    from scipy import integrate
    from mulan_amd.ode import solve_rk45
    rng = np.random.default_rng(1)
    n = 3000
    a = rng.uniform(-2.0, 3.0, n).astype(np.float32)
    y0 = rng.standard_normal(n)
    ad = torch.tensor(a).cuda()
    ref = integrate.solve_ivp(lambda t, y: (torch.tensor(a) * torch.tensor(y).float() * torch.tensor(np.float32(t))).double().numpy(),
                              (1, 0), y0, rtol=1e-6, atol=1e-6, method="RK45")
    sol = solve_rk45(lambda t, y32, out: out.copy_(ad * y32 * torch.tensor(np.float32(t), device="cuda")),
                     torch.tensor(y0).cuda(), (1.0, 0.0), rtol=1e-6, atol=1e-6)
    assert sol.t == 0.0 and sol.nfev == ref.nfev and sol.steps == len(ref.t) - 1
    assert _rel(sol.y.cpu().numpy(), ref.y[:, -1]) < 5e-7


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("mode,kind", [(0, "velocity"), (1, "vfe"), (2, "epsilon")])
@pytest.mark.parametrize("per_sample", [False, True])
def test_ode_drift_and_div_kernels(mode, kind, per_sample):
    """closed form around the network, its cotangent and the divergence term, with a diagonal stand-in network
    net = w * x (so the input gradient of the 'U-Net' for a cotangent c is w * c)"""
    from mulan_amd import ops
    rng = np.random.default_rng(5 + mode)
    B, D = 3, 3072
    x, w = rng.standard_normal((B, D)).astype(np.float32), rng.standard_normal((B, D)).astype(np.float32)
    h = (rng.integers(0, 2, (B, D)) * 2.0 - 1.0).astype(np.float32)
    gshape = (B,) if per_sample else (B, D)
    gt = rng.uniform(-13.3, 5.0, gshape).astype(np.float32)
    gp = rng.uniform(1.0, 40.0, gshape).astype(np.float32)
    dev = lambda v: torch.tensor(v).cuda()
    net = w * x
    drift, cot = ops.ode_drift(dev(net), dev(x), dev(gt), dev(gp), dev(h), mode)
    div = ops.ode_div(dev(w) * cot, dev(gt), dev(gp), dev(h), mode)
    bc = (lambda g: torch.tensor(g, dtype=torch.float64)[:, None]) if per_sample else \
        (lambda g: torch.tensor(g, dtype=torch.float64))
    wd = torch.tensor(w, dtype=torch.float64)
    f_ref, div_ref = tr.value_div(lambda xx: tr.ode_drift(wd * xx, xx, bc(gt), bc(gp), kind),
                                  torch.tensor(x, dtype=torch.float64), torch.tensor(h, dtype=torch.float64))
    assert _rel(drift.cpu().numpy(), f_ref.numpy()) < 5e-6
    scale = float(np.abs(div_ref.numpy()).max())
    assert np.abs(div.cpu().double().numpy() - div_ref.numpy()).max() < 2e-5 * scale + 1e-3
    only, none = ops.ode_drift(dev(net), dev(x), dev(gt), dev(gp), None, mode)
    assert none is None and torch.equal(only, drift)


def test_noise_kinds_and_dequantisation():
    from mulan_amd import ops
    n = 1 << 20
    u = ops.noise((n,), 7, 0, "cuda", "uniform").cpu().numpy()
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    r = ops.noise((n,), 7, 0, "cuda", "rademacher").cpu().numpy()
    assert set(np.unique(r)) == {-1.0, 1.0} and abs(r.mean()) < 4e-3
    t = ops.noise((n,), 7, 0, "cuda", "truncated_normal", -3.0, 3.0).cpu().numpy()
    var_tn = 1 - 6 * math.exp(-4.5) / math.sqrt(2 * math.pi) / 0.9973002
    assert t.min() >= -3.0 and t.max() <= 3.0 and abs(t.mean()) < 4e-3 and abs(t.var() - var_tn) < 4e-3
    assert not np.array_equal(ops.noise((64,), 7, 1, "cuda", "uniform").cpu().numpy(), u[:64])
    # dequantisation + the integer image handed to the encoder
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (4, 3072)).astype(np.uint8)
    x[0, :4] = [0, 0, 255, 255]
    for uniform, noise, s in ((True, u[:4 * 3072].reshape(4, 3072), 1.0),
                              (False, (t[:4 * 3072] * 40).reshape(4, 3072).astype(np.float32), math.exp(-6.65))):
        data, rq = ops.dequantize(torch.tensor(x).cuda(), torch.tensor(noise).cuda(), uniform, s)
        f = tr.encode(torch.tensor(x, dtype=torch.float64))
        nz = 2 * (torch.tensor(noise, dtype=torch.float64) - 0.5) / 256 if uniform else torch.tensor(noise).double() * s
        ref = f + nz
        assert _rel(data.cpu().numpy(), ref.numpy()) < 2e-7
        ref_q = torch.round(torch.clamp(128 * (ref + 1) - 0.5, 0, 255)).numpy()
        got = rq.cpu().numpy().astype(np.float64)
        edge = np.abs((128 * (ref.numpy() + 1) - 0.5) % 1 - 0.5) < 1e-3          # fp32 vs fp64 exactly at .5
        assert np.array_equal(got[~edge], ref_q[~edge]) and np.abs(got - ref_q).max() <= 1
        if uniform:                                                             # |noise| <= half a bin
            assert (got == x).mean() > 0.999 and np.abs(got - x).max() <= 1


def test_normal_logp_and_hard_topk():
    from mulan_amd import ops
    rng = np.random.default_rng(1)
    z = rng.standard_normal((5, 3072)).astype(np.float32) * 1.7
    assert _rel(ops.normal_logp(torch.tensor(z).cuda()).cpu().numpy(), tr.prior_logp(torch.tensor(z).double()).numpy()) < 1e-6
    logits = rng.standard_normal((6, 50)).astype(np.float32)
    logits[0, 3] = logits[0, 9]                                                 # a tie
    emb, kl = ops.topk_hard(torch.tensor(logits).cuda(), 15)
    ref = tr.logits_to_embeddings(torch.tensor(logits).double())
    assert torch.equal(emb.cpu().double(), ref) and set(np.unique(emb.cpu().numpy())) <= {0.0, 1.0}
    assert _rel(kl.cpu().numpy(), tr.gumbel_kl_loss(torch.tensor(logits).double()).numpy()) < 1e-5
    emb0, kl0 = ops.topk_hard(torch.zeros(2, 50, device="cuda"), 15)            # the plain VDM's apply_encoder
    assert torch.equal(emb0, torch.ones_like(emb0)) and float(kl0.abs().max()) < 1e-6


@pytest.mark.parametrize("kind,mode", [("velocity", 0), ("vfe", 1), ("epsilon", 2)])
@pytest.mark.parametrize("per_sample", [False, True])
def test_ode_drift_and_divergence_high_precision(kind, mode, per_sample):
    """reverse_ode(high_precision=True): the selects of ldm/model_mulan_velocity.py:410-417 (alpha, sigma) and
    ldm/model_mulan_epsilon.py:472-475 (sigma) in mulan_ode_drift / mulan_ode_div (mode | 4) against the oracle's
    float64 closed form, with gamma on both sides of both thresholds (sigmoid(g) <= 1e-3 below -6.9, 1 - sigmoid(g)
    <= 1e-3 above +6.9); drift, the cotangent d drift / d net and the explicit diagonal term of the divergence."""
    from mulan_amd import ops
    rng = np.random.default_rng(3)
    B, d = 4, 3072
    g = rng.uniform(-15.0, 15.0, (B, 1) if per_sample else (B, d)).astype(np.float32)
    g[np.abs(np.abs(g) - 6.9068) < 2e-3] += 0.01          # fp32 and float64 may disagree exactly at a threshold
    gp = rng.uniform(0.5, 20.0, g.shape).astype(np.float32)
    net, x = rng.standard_normal((B, d)).astype(np.float32), rng.standard_normal((B, d)).astype(np.float32)
    h = (rng.integers(0, 2, (B, d)) * 2.0 - 1.0).astype(np.float32)
    gx = rng.standard_normal((B, d)).astype(np.float32)
    c = lambda a: torch.tensor(a).cuda()
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    for hp in (False, True):
        drift, cot = ops.ode_drift(c(net), c(x), c(g.reshape(-1)), c(gp.reshape(-1)), c(h), mode | (4 if hp else 0))
        div = ops.ode_div(c(gx), c(g.reshape(-1)), c(gp.reshape(-1)), c(h), mode | (4 if hp else 0))
        n64, x64 = t64(net).requires_grad_(True), t64(x).requires_grad_(True)
        f = tr.ode_drift(n64, x64, t64(g), t64(gp), kind, hp)
        dn, dx = torch.autograd.grad(f.sum(), (n64, x64), allow_unused=True)
        dx = torch.zeros_like(f) if dx is None else dx
        assert _rel(drift.cpu().numpy(), f.detach().numpy()) < 2e-6, hp
        assert _rel(cot.cpu().numpy(), (dn * t64(h)).numpy()) < 2e-6, hp
        div_ref = ((t64(gx) + dx * t64(h)) * t64(h)).sum(dim=1).numpy()
        assert np.abs(div.cpu().double().numpy() - div_ref).max() < 2e-5 * np.abs(div_ref).max() + 1e-3, hp
    # the selects change something: where sigmoid(g) <= 1e-3 the plain sigma = sqrt(sigmoid(g)) and exp(g / 2) differ
    # (sqrt(sigmoid(g)) = exp(g / 2) / sqrt(1 + exp(g)): relative difference exp(g) / 2)
    lo = torch.tensor(g < -7.0).expand(B, d)
    a = tr.ode_drift(t64(net), t64(x), t64(g), t64(gp), kind, False)
    b = tr.ode_drift(t64(net), t64(x), t64(g), t64(gp), kind, True)
    assert float(((a - b).abs() / (b.abs() + 1e-300))[lo].max()) > 0.2 * math.exp(float(g[g < -7.0].max()))


def test_reverse_ode_high_precision_reaches_the_kernels():
    """VDM.reverse_ode(high_precision=True) and the likelihood function built with it evaluate the selected forms (not
    silently the plain ones): at t = 0 (gamma = -13.3: sigmoid(g) = 1.7e-6 <= 1e-3) the drift agrees with the
    oracle's high_precision drift, eager and replayed; the plain model_vdm.VDM has no such argument (ldm/model_vdm.py:243)
    and raises like the reference's apply() would."""
    from mulan_amd import model as M
    vdm, params, ref_params, ocfg = _setup("mulan_velocity", "vdm", True)
    rng = np.random.default_rng(11)
    B = 2
    img = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    ctx = vdm.ode_context(params, torch.tensor(img).cuda())
    x = torch.tensor(rng.standard_normal((B, 3072)).astype(np.float32)).cuda()
    h = torch.tensor((rng.integers(0, 2, (B, 3072)) * 2.0 - 1.0).astype(np.float32)).cuda()
    dev_params = params_on_device(ref_params)
    emb = ctx["emb"].cpu().double()
    for t in (0.0, 0.4):
        drift, div = vdm.reverse_ode(params, x, ctx, t, h, high_precision=True)
        ref = on_device(lambda xx, ee: tr.reverse_ode(dev_params, ocfg, xx, ee, t, True))(
            x.cpu().double().reshape(B, 32, 32, 3), emb)
        assert _rel(drift.cpu().numpy(), ref.reshape(B, -1).numpy()) < 3e-4, t
        f = M.ode_function(vdm, params, ctx, B, "cuda", True, high_precision=True)
        assert isinstance(f, M.GraphedOdeFunction)
        d2, v2 = torch.empty_like(drift), torch.empty_like(div)
        f(t, x, h, d2, v2)
        assert torch.equal(d2, drift) and torch.equal(v2, div)
    plain, _ = vdm.reverse_ode(params, x, ctx, 0.0, h)
    hp, _ = vdm.reverse_ode(params, x, ctx, 0.0, h, high_precision=True)
    assert not torch.equal(plain, hp)
    from mulan_amd.rng import PRNGKey
    cfg, _ = make_cfg()
    pvdm = M.make_vdm("vdm", dataclasses.replace(cfg, gamma_type="learnable_scalar", z_conditioning=False, reparam_type="noise"))
    pparams = M.tree_map(lambda t: t.cuda(), pvdm.init(PRNGKey(0)))
    with pytest.raises(TypeError):
        pvdm.reverse_ode(pparams, x, pvdm.ode_context(pparams, torch.tensor(img).cuda()), 0.5, h, high_precision=True)


# ------------------------------------------------------------------------------------------------ model level
def _setup(vdm_type, unet_type, vfe, seed=5, E=128):
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg(vdm_type, unet_type, vfe, E=E)
    ref_params = tr.init_params(ocfg, seed=seed, dtype=torch.float64)
    vdm = M.make_vdm(vdm_type, cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    return vdm, params, ref_params, ocfg


@pytest.mark.parametrize("vdm_type,unet_type,vfe", [("mulan_velocity", "vdm", False), ("mulan_velocity", "vdm", True),
                                                    ("mulan_epsilon", "vdm", False), ("mulan_velocity", "ldm", False)])
def test_reverse_ode_value_and_divergence(vdm_type, unet_type, vfe):
    """drift and Hutchinson term of one function evaluation (U-Net forward + input-gradient pass) vs float64 autograd"""
    vdm, params, ref_params, ocfg = _setup(vdm_type, unet_type, vfe)
    rng = np.random.default_rng(9)
    B = 2
    img = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    ctx = vdm.ode_context(params, torch.tensor(img).cuda())
    E, FL = ocfg["n_embd"], ocfg["forward_n_layer"]
    logits = tr.unet_encoder(tr.encode(torch.tensor(img).double()), ref_params["encoder_model"], E, FL)
    emb = tr.logits_to_embeddings(logits)
    assert torch.equal(ctx["emb"].cpu().double(), emb)
    assert _rel(ctx["kl"].cpu().numpy(), tr.gumbel_kl_loss(logits).numpy()) < 1e-4
    x = rng.standard_normal((B, 3072)).astype(np.float32)
    h = (rng.integers(0, 2, (B, 3072)) * 2.0 - 1.0).astype(np.float32)
    dev_params = params_on_device(ref_params)
    for t in (0.0, 0.37, 1.0):
        drift, div = vdm.reverse_ode(params, torch.tensor(x).cuda(), ctx, t, torch.tensor(h).cuda())
        xr = torch.tensor(x, dtype=torch.float64).reshape(B, 32, 32, 3).requires_grad_(True)
        f = on_device(lambda xx, ee: tr.reverse_ode(dev_params, ocfg, xx, ee, t))(xr, emb)
        hr = torch.tensor(h, dtype=torch.float64).reshape(B, 32, 32, 3)
        (g,) = torch.autograd.grad((f * hr).sum(), xr)
        div_ref = (g * hr).reshape(B, -1).sum(dim=1).numpy()
        assert _rel(drift.cpu().numpy(), f.detach().reshape(B, -1).numpy()) < 3e-4, t
        # h^T J h sums 3072 products g_i h_i whose fp32 errors (2e-3 of max|g| each, the input-gradient bar of
        # tests/test_gpu_model.py) add up like a random walk: sqrt(3072) * 2e-3 = 0.11 of max|g|
        bound = 0.11 * float(g.abs().max()) + 1e-3 * np.abs(div_ref).max()
        assert np.abs(div.cpu().double().numpy() - div_ref).max() < bound, (t, div.cpu().numpy(), div_ref, bound)
        only, none = vdm.reverse_ode(params, torch.tensor(x).cuda(), ctx, t)
        assert none is None and torch.equal(only, drift)


def test_plain_vdm_reverse_ode():
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg, ocfg = make_cfg()
    cfg = dataclasses.replace(cfg, gamma_type="learnable_scalar", z_conditioning=False, reparam_type="noise")
    full = tr.init_params(ocfg, seed=4, dtype=torch.float64)
    ref_params = {"score_model": full["score_model"],
                  "gamma": {"w": torch.tensor([-17.0], dtype=torch.float64), "b": torch.tensor([-12.5], dtype=torch.float64)}}
    ref_params["score_model"]["dense0"]["kernel"] = ref_params["score_model"]["dense0"]["kernel"][:129].clone()
    vdm = M.make_vdm("vdm", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    rng = np.random.default_rng(2)
    B = 2
    ctx = vdm.ode_context(params, torch.zeros(B, 32, 32, 3, dtype=torch.uint8, device="cuda"))
    assert torch.equal(ctx["emb"], torch.ones(B, 50, device="cuda"))
    x = rng.standard_normal((B, 3072)).astype(np.float32)
    h = rng.standard_normal((B, 3072)).astype(np.float32)
    drift, div = vdm.reverse_ode(params, torch.tensor(x).cuda(), ctx, 0.6, torch.tensor(h).cuda())
    xr = torch.tensor(x, dtype=torch.float64).reshape(B, 32, 32, 3).requires_grad_(True)
    f = tr.plain_reverse_ode(ref_params, ocfg, xr, torch.ones(B, 50, dtype=torch.float64), 0.6)
    hr = torch.tensor(h, dtype=torch.float64).reshape(B, 32, 32, 3)
    (g,) = torch.autograd.grad((f * hr).sum(), xr)
    div_ref = (g * hr).reshape(B, -1).sum(dim=1).numpy()
    assert _rel(drift.cpu().numpy(), f.detach().reshape(B, -1).numpy()) < 3e-4
    assert np.abs(div.cpu().double().numpy() - div_ref).max() < 0.11 * float((g * hr).abs().max()) + 1e-3 * np.abs(div_ref).max()


class _FakeExperiment:
    def __init__(self, model, params):
        from mulan_amd.train_state import TrainState  # noqa: F401
        self.model, self.orig_params, self.device = model, params, torch.device("cuda")
        self.state = type("S", (), {"ema_params": None, "param_packer": lambda self, which: None})()


@pytest.mark.parametrize("vdm_type,vfe,deq", [("mulan_velocity", True, "tn"), ("mulan_epsilon", False, "uniform")])
def test_ode_likelihood_matches_oracle(vdm_type, vfe, deq):
    """the whole likelihood_fn (dequantise -> encoder -> embedding -> Dormand-Prince over [x, delta logp] with a
    fixed Hutchinson probe -> prior) against the oracle integrating its float64 model.
    (a) on a prescribed time grid, so both sides take identical steps: log p to fp32 noise;
    (b) adaptive at rtol = atol = 1e-3 against scipy's RK45 on the oracle: same number of function evaluations.
    With only ~5 adaptive steps the quadrature of delta logp (the divergence runs 40 -> 3500 -> -350 over [0, 1], and
    the RMS error norm weighs the B logp components B / (B * 3073)) is itself only good to a few per cent, so (b)
    compares log p loosely; the controller's own parity with scipy is test_rk45_takes_scipys_steps."""
    from mulan_amd.evaluators import get_ode_likelihood_fn, _get_bpd_offset
    from mulan_amd import model as M
    from mulan_amd import ops
    from mulan_amd.rng import PRNGKey
    vdm, params, ref_params, ocfg = _setup(vdm_type, "vdm", vfe)
    # a smooth stand-in for a trained network: output = z + 0.002 * (random U-Net).  At full scale the random-init
    # flow is violently expanding and two correct integrations end hundreds of nats apart (measured) although every
    # single function evaluation agrees (test_reverse_ode_value_and_divergence).
    ref_params["score_model"]["conv_out"]["kernel"] = ref_params["score_model"]["conv_out"]["kernel"] * 0.002
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    B = 1                                   # the float64 oracle integrates on the CPU: one image keeps it affordable
    rng = np.random.default_rng(4)
    img = torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8))
    kind = "truncated_normal" if deq == "tn" else "uniform"
    u = ops.noise((B, 3072), 11, 0, "cuda", kind)
    probe = ops.noise((B, 3072), 12, 0, "cuda", "rademacher")
    fn = get_ode_likelihood_fn(_FakeExperiment(vdm, params), rtol=1e-3, atol=1e-3, dequantization=deq)
    E, FL = ocfg["n_embd"], ocfg["forward_n_layer"]
    dev_params = params_on_device(ref_params)       # the oracle's network evaluations on its device (tests/oracle_dev.py);
    oracle = lambda **kw: tr.ode_likelihood(        # the integrator (scipy / the fixed grid) stays on the host
        on_device(lambda x, emb, t: tr.reverse_ode(dev_params, ocfg, x, emb, t)),
        on_device(lambda im: tr.unet_encoder(tr.encode(im), dev_params["encoder_model"], E, FL)),
        img, u.cpu().double(), lambda: probe.cpu().double(), dequantization=deq, rtol=1e-3, atol=1e-3, **kw)
    grid = [0.0, 0.04, 0.14, 0.32, 0.55, 0.8, 1.0]
    log_p, log_q, aux, info = fn(PRNGKey(0), img.cuda(), deterministic_noise=True, u=u, probes=lambda: probe, t_grid=grid)
    lp_ref, lq_ref, aux_ref, nfev = oracle(t_grid=grid)
    # log p ~ -3e3 nats; 1 nat = 4.7e-4 bits/dim, a tenth of the +-0.005 BPD bar (measured: 0.09 ... 0.5 nat)
    print("fixed-grid log_p", log_p.cpu().numpy(), lp_ref.numpy())
    assert np.abs(log_p.cpu().numpy() - lp_ref.numpy()).max() < 1.0, (log_p, lp_ref)
    assert info["nfev"] == nfev + 1                       # the device integrator evaluates f(t0) once up front (FSAL)
    assert _rel(aux.cpu().numpy(), aux_ref.numpy()) < 1e-4
    if deq == "tn":
        assert _rel(log_q.cpu().numpy(), lq_ref.numpy()) < 1e-6
    else:
        assert log_q is None and lq_ref is None
    assert abs(_get_bpd_offset(deq, 1) - tr.bpd_offset(deq, 1)) < 1e-12
    assert abs(_get_bpd_offset("tn", 20) - tr.bpd_offset("tn", 20)) < 1e-12
    if deq != "tn":          # the adaptive comparison once is enough (each oracle solve costs ~1 min of CPU)
        return
    log_p, _, _, info = fn(PRNGKey(0), img.cuda(), deterministic_noise=True, u=u, probes=lambda: probe)
    lp_ref, _, _, nfev = oracle()
    assert abs(info["nfev"] - nfev) <= 6, (info["nfev"], nfev)
    assert _rel(log_p.cpu().numpy(), lp_ref.numpy()) < 0.05, (log_p, lp_ref)


def test_ode_likelihood_high_precision_matches_oracle():
    """get_ode_likelihood_fn(high_precision=True) (ldm/notebook_utils.py:290: the flag reaches VDM.reverse_ode at every
    function evaluation) on a prescribed time grid against the oracle integrating its float64 model with
    reverse_ode(high_precision=True): log p to fp32 noise -- and not the plain path's numbers (the grid starts at t = 0,
    where gamma = -13.3 takes the exp(g / 2) branch of sigma)."""
    from mulan_amd.evaluators import get_ode_likelihood_fn
    from mulan_amd import model as M
    from mulan_amd import ops
    from mulan_amd.rng import PRNGKey
    vdm, params, ref_params, ocfg = _setup("mulan_velocity", "vdm", True)
    ref_params["score_model"]["conv_out"]["kernel"] = ref_params["score_model"]["conv_out"]["kernel"] * 0.002
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    B = 1
    rng = np.random.default_rng(4)
    img = torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8))
    u = ops.noise((B, 3072), 11, 0, "cuda", "uniform")
    probe = ops.noise((B, 3072), 12, 0, "cuda", "rademacher")
    E, FL = ocfg["n_embd"], ocfg["forward_n_layer"]
    dev_params = params_on_device(ref_params)
    grid = [0.0, 0.04, 0.14, 0.32, 0.55, 0.8, 1.0]
    got = {}
    for hp in (True, False):
        fn = get_ode_likelihood_fn(_FakeExperiment(vdm, params), rtol=1e-3, atol=1e-3, dequantization="uniform", high_precision=hp)
        log_p, _, _, info = fn(PRNGKey(0), img.cuda(), deterministic_noise=True, u=u, probes=lambda: probe, t_grid=grid)
        got[hp] = log_p.cpu().numpy()
    lp_ref, _, _, nfev = tr.ode_likelihood(
        on_device(lambda x, emb, t: tr.reverse_ode(dev_params, ocfg, x, emb, t, True)),
        on_device(lambda im: tr.unet_encoder(tr.encode(im), dev_params["encoder_model"], E, FL)),
        img, u.cpu().double(), lambda: probe.cpu().double(), dequantization="uniform", rtol=1e-3, atol=1e-3, t_grid=grid)
    lp_plain, _, _, _ = tr.ode_likelihood(
        on_device(lambda x, emb, t: tr.reverse_ode(dev_params, ocfg, x, emb, t, False)),
        on_device(lambda im: tr.unet_encoder(tr.encode(im), dev_params["encoder_model"], E, FL)),
        img, u.cpu().double(), lambda: probe.cpu().double(), dequantization="uniform", rtol=1e-3, atol=1e-3, t_grid=grid)
    print("high_precision fixed-grid log_p", got[True], lp_ref.numpy(), "plain", got[False], lp_plain.numpy())
    assert np.abs(got[True] - lp_ref.numpy()).max() < 1.0, (got[True], lp_ref)
    assert np.abs(got[False] - lp_plain.numpy()).max() < 1.0, (got[False], lp_plain)
    # the two forms are different integrands near t = 0 (the expanding random-init flow amplifies the 5e-4 relative
    # difference of sigma below gamma = -6.9): each implementation must follow its own oracle, not the other's
    assert np.abs(got[True] - got[False]).max() > 2.0


def test_ode_sampler_matches_oracle():
    """get_sample_fn: prior at t = 1 integrated down to t = 0 along the drift, same embedding and prior draw on both
    sides (recomputed from the product's Philox stream), smooth stand-in network as above"""
    from mulan_amd.evaluators import get_sample_fn
    from mulan_amd import model as M
    from mulan_amd import ops
    from mulan_amd.rng import PRNGKey
    vdm, params, ref_params, ocfg = _setup("mulan_velocity", "vdm", False)
    ref_params["score_model"]["conv_out"]["kernel"] = ref_params["score_model"]["conv_out"]["kernel"] * 0.002
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), ref_params), params)
    fn = get_sample_fn(_FakeExperiment(vdm, params), rtol=1e-4, atol=1e-4)
    n = 1
    z, nfev = fn(PRNGKey(3), sample_size=n)
    assert z.shape == (n, 32, 32, 3) and bool(torch.isfinite(z).all())
    # the draws sample_fn made
    rng, logits_rng = PRNGKey(3).split()
    emb, _ = ops.topk_hard(logits_rng.normal((n, 50), "cuda"), 15)
    rng, _ = rng.split()
    rng, prior_rng = rng.split()
    prior = prior_rng.normal((n, 3072), "cuda")
    dev_params = params_on_device(ref_params)
    z_ref, nfev_ref = tr.ode_sample(on_device(lambda x, e, t: tr.reverse_ode(dev_params, ocfg, x, e, t)), emb.cpu().double(),
                                    prior.cpu().double(), rtol=1e-4, atol=1e-4)
    assert abs(nfev - nfev_ref) <= 6, (nfev, nfev_ref)
    assert _rel(z.cpu().numpy(), z_ref.numpy()) < 2e-3


@pytest.mark.parametrize("vdm_type,unet_type,vfe", [("mulan_velocity", "vdm", True), ("mulan_epsilon", "ldm", False)])
def test_replayed_ode_function_equals_the_eager_one(vdm_type, unet_type, vfe):
    """model.GraphedOdeFunction: one function evaluation (U-Net forward + Hutchinson term through its input gradient) as a
    replayed HIP graph -- state, probe and time reach the kernels through static buffers.  Drift and divergence are bit
    for bit those of the eager VDM.reverse_ode at every time asked for, with the divergence (likelihood) and without
    (sampler); the replayed solver run equals the eager one."""
    from mulan_amd import model as M
    from mulan_amd import ode
    vdm, params, _, _ = _setup(vdm_type, unet_type, vfe)
    rng = np.random.default_rng(3)
    B = 3
    img = torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)).cuda()
    ctx = vdm.ode_context(params, img)
    f_div = M.GraphedOdeFunction(vdm, params, ctx, B, torch.device("cuda"), True)
    f_plain = M.GraphedOdeFunction(vdm, params, ctx, B, torch.device("cuda"), False)
    for t in (0.0, 0.123, 0.77, 1.0, 0.123):
        x = torch.tensor(rng.standard_normal((B, 3072)).astype(np.float32)).cuda()
        h = torch.tensor((rng.integers(0, 2, (B, 3072)) * 2.0 - 1.0).astype(np.float32)).cuda()
        drift, div = vdm.reverse_ode(params, x, ctx, t, h)
        d2, v2 = torch.empty_like(drift), torch.empty_like(div)
        f_div(t, x, h, d2, v2)
        assert torch.equal(drift, d2) and torch.equal(div, v2), t
        only, _ = vdm.reverse_ode(params, x, ctx, t, None)
        d3 = torch.empty_like(only)
        f_plain(t, x, None, d3)
        assert torch.equal(only, d3), t
    # a whole (fixed-grid) solve through ode_function: replayed == eager
    y0 = torch.tensor(rng.standard_normal(B * 3072)).cuda()
    runs = []
    for graph in (True, False):
        f = M.ode_function(vdm, params, ctx, B, torch.device("cuda"), False, graph=graph)
        sol = ode.solve_fixed(lambda t, y32, out: f(t, y32.view(B, 3072), None, out.view(B, 3072)), y0, [1.0, 0.6, 0.3, 0.0])
        runs.append(sol.y.clone())
    assert torch.equal(runs[0], runs[1])
