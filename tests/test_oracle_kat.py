"""CPU tests of the oracle: analytic known-answer tests, agreement of the NumPy and torch
restatements, and the committed golden fixtures.  The reference has no tests or vectors of its own."""
import os

import numpy as np
import torch

from oracle import mulan_np as onp
from oracle import torch_ref as tr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _abc(rng, B=4):
    a = rng.standard_normal((B, 3072))
    b = rng.standard_normal((B, 3072))
    c = 1e-3 + np.logaddexp(rng.standard_normal((B, 3072)), 0)
    return a, b, c


def test_gamma_fixed_end_points_and_monotone():
    rng = np.random.default_rng(0)
    a, b, c = _abc(rng)
    assert np.allclose(onp.poly_gamma(a, b, c, np.zeros(4)), -13.3, atol=1e-12)      # A.9 #1
    assert np.allclose(onp.poly_gamma(a, b, c, np.ones(4)), 5.0, atol=1e-9)
    ts = np.linspace(0, 1, 33)
    g = np.stack([onp.poly_gamma(a, b, c, np.full(4, t)) for t in ts])
    assert np.all(np.diff(g, axis=0) >= -1e-9)                                        # A.9 #3
    assert np.all(onp.poly_gamma_grad_t(a, b, c, np.full(4, 0.3)) >= 0)


def test_gamma_grad_matches_finite_difference():
    rng = np.random.default_rng(1)
    a, b, c = _abc(rng)
    t = rng.uniform(0.05, 0.95, 4)
    h = 1e-6
    fd = (onp.poly_gamma(a, b, c, t + h) - onp.poly_gamma(a, b, c, t - h)) / (2 * h)
    assert np.allclose(fd, onp.poly_gamma_grad_t(a, b, c, t), rtol=1e-6, atol=1e-6)   # A.9 #2


def test_fresh_init_gamma_is_cubic():
    rng = np.random.default_rng(2)
    _, b, c = _abc(rng)
    a = np.zeros_like(b)                                                              # dense_out_a zero-init
    ts = np.linspace(0, 1, 9)
    g = np.stack([onp.poly_gamma(a, b, c, np.full(4, t))[0, :5] for t in ts])
    coef = np.polyfit(ts, g, 3)
    assert np.allclose(np.polyval(coef, ts[:, None] * np.ones((1, 5))), g, atol=1e-9)  # A.9 #4


def test_velocity_loss_equals_epsilon_loss_under_reparameterisation():
    rng = np.random.default_rng(3)
    B = 2
    f = rng.uniform(-1, 1, (B, 32, 32, 3))
    g = rng.uniform(-10, 4, (B, 32, 32, 3))
    gp = rng.uniform(1, 30, (B, 32, 32, 3))
    eps, eps_hat = rng.standard_normal(f.shape), rng.standard_normal(f.shape)
    s2 = onp.sigmoid(g)
    al, sg = np.sqrt(1 - s2), np.sqrt(s2)
    zt = al * f + sg * eps
    v_hat = (eps_hat - sg * zt) / al
    lv = onp.diffusion_loss_velocity(f, g, gp, eps, zt, v_hat)
    le = onp.diffusion_loss_epsilon(gp, eps, eps_hat)
    assert np.allclose(lv, le, rtol=1e-9)                                             # A.9 #5
    lvfe = onp.diffusion_loss_velocity(f, g, gp, eps, zt, eps_hat, velocity_from_epsilon=True)
    assert np.allclose(lvfe, le, rtol=1e-9)                                           # A.9 #6


def test_decoder_is_a_distribution_and_picks_the_bin():
    rng = np.random.default_rng(4)
    x = rng.integers(0, 256, (2, 4, 4, 3))
    f = onp.encode(x)
    lp = onp.decode_logprobs(f, np.full(f.shape, -13.3))
    assert np.allclose(np.exp(lp).sum(-1), 1.0)
    assert np.array_equal(lp.argmax(-1), x)
    top2 = np.sort(lp, axis=-1)[..., -2:]
    assert np.allclose(top2[..., 1] - top2[..., 0], 0.5 * (2 / 256 * np.exp(6.65)) ** 2, rtol=1e-3)  # gap ~18.2
    assert np.all(onp.logprob(x, f, np.full(f.shape, -13.3)) <= 0)                   # A.9 #8


def test_klz_closed_form():
    f = np.full((1, 32, 32, 3), 0.5)
    z = np.zeros_like(f)
    _, _, _, klz, _, v1 = onp.elbo_pre(np.full((1, 32, 32, 3), 191), np.full_like(f, -13.3), np.full_like(f, 5.0),
                                       np.full_like(f, 0.0), z, z)
    s2 = 1 / (1 + np.exp(-5.0))
    fx = onp.encode(np.array(191.0))
    assert np.isclose(v1, s2) and abs(s2 - 0.993307) < 1e-6
    assert np.isclose(klz[0], 3072 * 0.5 * ((1 - s2) * fx ** 2 + s2 - np.log(s2) - 1))  # A.9 #9


def test_antithetic_times_cover_unit_interval():
    t = onp.antithetic_t(0.73, 8)
    assert np.allclose(np.sort(t), (0.73 % 0.125) + np.arange(8) / 8)                # A.9 #10


def test_topk_is_k_hot_with_straight_through_value():
    rng = np.random.default_rng(5)
    logits = rng.standard_normal((6, 50))
    raw = rng.gamma(1 / 15, size=(10, 6, 50))
    emb, kl, soft = onp.topk_embedding_and_loss(logits, raw, 15)
    assert np.all(np.round(emb).sum(1) == 15) and np.allclose(emb, np.round(emb), atol=1e-12)
    assert np.all(kl >= 0) and np.allclose(np.linalg.norm(soft, axis=1), 1)


def test_adamw_matches_torch_optimizer():
    rng = np.random.default_rng(6)                                                    # A.9 #11
    p0, g = rng.standard_normal(50), rng.standard_normal(50)
    tp = torch.tensor(p0, requires_grad=True)
    opt = torch.optim.AdamW([tp], lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    p, m, v, ema = p0.copy(), np.zeros(50), np.zeros(50), p0.copy()
    for step in range(1, 4):
        tp.grad = torch.tensor(g * step)
        opt.step()
        p, m, v, ema = onp.adamw_ema_step(p, g * step, m, v, ema, 2e-4, step, np.ones(50))
    # torch decays with p*(1 - lr*wd) before the Adam update; optax adds wd*p to the update: equal to O(lr^2)
    assert np.allclose(p, tp.detach().numpy(), atol=1e-8)
    assert not np.allclose(ema, p)


def test_lr_schedule_warmup():
    assert onp.lr_schedule(0) == 0 and np.isclose(onp.lr_schedule(50), 1e-4) and onp.lr_schedule(5000) == 2e-4


def test_philox_known_answer():
    # Random123 known-answer vector for philox4x32-10: counter = key = 0
    r = onp.philox4x32_10(0, np.array([0], dtype=np.uint64))[0]
    assert [hex(int(v)) for v in r] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']


def test_numpy_and_torch_restatements_agree():
    cfg = dict(vdm_type='mulan_velocity', n_embd=32, n_layer=1, forward_n_layer=1, latent_k=15, unet_type='vdm')
    P = tr.init_params(cfg, seed=0)
    rng = np.random.default_rng(0)
    B = 2
    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    raw = rng.gamma(1 / 15, size=(10, B, 50))
    e0, e = rng.standard_normal((B, 32, 32, 3)), rng.standard_normal((B, 32, 32, 3))
    for extra in ({}, {"velocity_from_epsilon": True}, {"vdm_type": "mulan_epsilon"}, {"unet_type": "ldm"}):
        c = dict(cfg, **extra)
        Pc = tr.init_params(c, seed=1) if extra.get("unet_type") == "ldm" else P
        a = onp.mulan_forward(tr.to_np_tuples(Pc), c, x, 0.3, raw, e0, e)
        b = tr.mulan_forward(Pc, c, torch.from_numpy(x), 0.3, torch.from_numpy(raw), torch.from_numpy(e0),
                             torch.from_numpy(e))
        assert abs(a["bpd"] - float(b["bpd"])) < 1e-10 * abs(a["bpd"])
        assert np.allclose(a["loss_diff"], b["loss_diff"].detach().numpy(), rtol=1e-10)


def test_golden_closed_forms_fixture():
    z = np.load(os.path.join(GOLD, "closed_forms.npz"))
    a, b, c, t = z["a"], z["b"], z["c"], z["t"]
    shp = z["eps"].shape
    gt = onp.poly_gamma(a, b, c, t).reshape(shp)
    gp = onp.poly_gamma_grad_t(a, b, c, t).reshape(shp)
    assert np.allclose(gt, z["g_t"], rtol=1e-12) and np.allclose(gp, z["g_prime"], rtol=1e-12)
    g0 = onp.poly_gamma(a, b, c, np.zeros(len(t))).reshape(shp)
    g1 = onp.poly_gamma(a, b, c, np.ones(len(t))).reshape(shp)
    f, zt, recon, klz, v0, v1 = onp.elbo_pre(z["x"], g0, g1, gt, z["eps_0"], z["eps"])
    assert np.allclose(zt, z["z_t"]) and np.allclose(recon, z["loss_recon"]) and np.allclose(klz, z["loss_klz"])
    assert np.allclose(onp.diffusion_loss_velocity(f, gt, gp, z["eps"], zt, z["net"]), z["loss_diff_velocity"])
    assert np.allclose(onp.diffusion_loss_velocity(f, gt, gp, z["eps"], zt, z["net"], True), z["loss_diff_vfe"])
    assert np.allclose(onp.diffusion_loss_epsilon(gp, z["eps"], z["net"]), z["loss_diff_epsilon"])
    emb, kl, _ = onp.topk_embedding_and_loss(z["logits"], z["gamma_raw"], 15)
    assert np.array_equal(emb, z["embedding"]) and np.allclose(kl, z["kl_z"])
    assert np.allclose(onp.timestep_embedding(z["temb_t"], 128), z["temb"])
    assert np.allclose(onp.fourier_features(z["fourier_z"]), z["fourier"])
    # torch restatement against the same file
    tt = lambda v: torch.tensor(v)
    assert np.allclose(tr.poly_gamma(tt(a), tt(b), tt(c), tt(t)).numpy().reshape(shp), z["g_t"], rtol=1e-12)
    e2, k2 = tr.topk_embedding_and_loss(tt(z["logits"]), tt(z["gamma_raw"]), 15)
    assert np.allclose(e2.numpy(), z["embedding"]) and np.allclose(k2.numpy(), z["kl_z"])


def test_golden_tiny_model_fixture():
    z = np.load(os.path.join(GOLD, "tiny_model.npz"))
    cfg = dict(vdm_type="mulan_epsilon", n_embd=128, n_layer=1, forward_n_layer=1, latent_k=15, unet_type="vdm")
    P = tr.init_params(cfg, seed=int(z["param_seed"]), dtype=torch.float64)
    r = tr.mulan_forward(P, cfg, torch.tensor(z["x"]), float(z["t0"]), torch.tensor(z["gamma_raw"]),
                         torch.tensor(z["eps_0"]), torch.tensor(z["eps"]))
    assert abs(float(r["bpd"]) - float(z["epsilon_bpd"])) < 1e-9 * abs(float(z["epsilon_bpd"]))
    assert np.allclose(r["loss_diff"].detach().numpy(), z["epsilon_diff"], rtol=1e-9)


def test_ancestral_step_marginals_are_consistent():
    """KAT for the sampler restatement: with the exact eps of z_t = alpha_t x + sigma_t eps the reverse step is
    q(z_s | z_t, x): mean alpha_s x + (sigma_s^2 alpha_t / (alpha_s sigma_t)) eps, variance sigma_s^2 - that^2, so the
    total noise variance of z_s is sigma_s^2 (variance-preserving marginals, VDM eq. 32-34); all three
    parameterisations agree when each is handed its own exact target"""
    rng = np.random.default_rng(0)
    n = 4096
    x, eps = torch.tensor(rng.uniform(-1, 1, n)), torch.tensor(rng.standard_normal(n))
    g_t = torch.tensor(rng.uniform(-13.3, 5.0, n))
    g_s = g_t - torch.tensor(rng.uniform(0.01, 2.0, n))
    a_t, s_t = torch.sqrt(torch.sigmoid(-g_t)), torch.sqrt(torch.sigmoid(g_t))
    a_s, s_s = torch.sqrt(torch.sigmoid(-g_s)), torch.sqrt(torch.sigmoid(g_s))
    z_t = a_t * x + s_t * eps
    zero, one = torch.zeros(n, dtype=torch.float64), torch.ones(n, dtype=torch.float64)
    mean = tr.ancestral_step(z_t, eps, g_t, g_s, zero, "epsilon")
    k = s_s ** 2 * a_t / (a_s * s_t)
    assert torch.allclose(mean, a_s * x + k * eps, rtol=1e-9, atol=1e-12)
    std = tr.ancestral_step(z_t, eps, g_t, g_s, one, "epsilon") - mean
    assert torch.allclose(k ** 2 + std ** 2, s_s ** 2, rtol=1e-9, atol=1e-12)
    v = a_t * eps - s_t * x                                           # velocity target (model_mulan_velocity.py:247)
    assert torch.allclose(tr.ancestral_step(z_t, v, g_t, g_s, zero, "velocity"), mean, rtol=1e-8, atol=1e-10)
    assert torch.allclose(tr.ancestral_step(z_t, x, g_t, g_s, zero, "input"), mean, rtol=1e-6, atol=1e-8)


def test_decode_argmax_inverts_encode():
    x = torch.arange(256).repeat(3, 1)
    for g in (-13.3, -9.0, -5.0):
        g0 = torch.full(x.shape, g, dtype=torch.float64)
        z0 = tr.encode(x.double()) * torch.sqrt(torch.sigmoid(-g0))   # noise-free z_0 = alpha_0 f(x)
        assert torch.equal(tr.decode_argmax(z0, g0), x)
    assert torch.equal(tr.deterministic_embedding(2, 50, 15).sum(dim=1), torch.full((2,), 15.0, dtype=torch.float64))
    assert torch.equal(tr.deterministic_embedding(2, 50, 15)[:, :15], torch.ones(2, 15, dtype=torch.float64))


def test_sample_loop_runs_and_is_deterministic_given_noise():
    cfg = dict(vdm_type='mulan_velocity', n_embd=32, n_layer=1, forward_n_layer=1, latent_k=15, unet_type='vdm')
    P = tr.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(0)
    z = torch.randn(1, 3072, generator=g, dtype=torch.float64)
    eps = [torch.randn(1, 3072, generator=g, dtype=torch.float64) for _ in range(2)]
    z0, x0 = tr.mulan_sample_loop(P, cfg, z, eps)
    z1, x1 = tr.mulan_sample_loop(P, cfg, z, eps)
    assert torch.equal(z0, z1) and torch.equal(x0, x1) and x0.shape == (1, 32, 32, 3)
    assert int(x0.min()) >= 0 and int(x0.max()) <= 255 and torch.isfinite(z0).all()


def test_ode_drift_forms_agree_on_exact_targets():
    """the velocity, velocity-from-epsilon and epsilon forms of reverse_ode are the same ODE when each network
    output is its own exact target (v = alpha eps - sigma x0 = (eps - sigma z) / alpha; the vfe network predicts eps)"""
    rng = np.random.default_rng(1)
    n = 2048
    z, eps = torch.tensor(rng.standard_normal(n)), torch.tensor(rng.standard_normal(n))
    g_t, g_p = torch.tensor(rng.uniform(-13.3, 5.0, n)), torch.tensor(rng.uniform(1.0, 30.0, n))
    alpha = torch.sqrt(torch.sigmoid(-g_t))
    sigma = torch.sqrt(torch.sigmoid(g_t))
    d_eps = tr.ode_drift(eps, z, g_t, g_p, "epsilon")
    d_vel = tr.ode_drift((eps - sigma * z) / alpha, z, g_t, g_p, "velocity")
    d_vfe = tr.ode_drift(eps, z, g_t, g_p, "vfe")
    assert torch.allclose(d_vel, d_eps, rtol=1e-9, atol=1e-12)
    # the reference's vfe transform -e^{g/2} x + sqrt(1 + e^g) eps equals (eps - sigma z) / alpha
    assert torch.allclose(d_vfe, d_eps, rtol=1e-6, atol=1e-9)


def test_hutchinson_is_exact_for_a_diagonal_jacobian():
    rng = np.random.default_rng(2)
    a = torch.tensor(rng.standard_normal((3, 4, 4, 3)))
    x = torch.tensor(rng.standard_normal((3, 4, 4, 3)))
    h = torch.tensor(rng.integers(0, 2, (3, 4, 4, 3)) * 2.0 - 1.0)
    f, div = tr.value_div(lambda xx: a * xx + torch.sin(xx), x, h)
    assert torch.allclose(f, a * x + torch.sin(x))
    assert torch.allclose(div, (a + torch.cos(x)).reshape(3, -1).sum(dim=1), rtol=1e-12)


def test_ode_likelihood_of_a_gaussian_is_exact():
    """KAT for the whole likelihood machinery (dequantisation, flattening, solve_ivp, Hutchinson, prior): for data
    ~ N(0, s^2 I) under a variance-preserving schedule the exact eps-prediction is sigma z / var_t with
    var_t = s^2 + (1 - s^2) sigma_t^2; the flow is z_1 = z_0 sqrt(var_1 / var_0) and
    log p = log N(z_1; 0, I) + (D / 2) log(var_1 / var_0)."""
    import math
    s2 = 0.25
    gmin, gmax = -13.3, 5.0

    def drift(x, emb, t):
        g = torch.tensor(gmin + (gmax - gmin) * t, dtype=x.dtype)
        var = s2 + (1 - s2) * torch.sigmoid(g)
        eps_hat = torch.sqrt(torch.sigmoid(g)) * x / var
        return tr.ode_drift(eps_hat, x, g, torch.tensor(gmax - gmin, dtype=x.dtype), "epsilon")

    rng = np.random.default_rng(3)
    B = 2
    x_u8 = torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8))
    u = torch.tensor(rng.uniform(0, 1, (B, 3072)))
    probe = torch.tensor(rng.integers(0, 2, (B, 3072)) * 2.0 - 1.0)
    log_p, log_q, aux, nfev = tr.ode_likelihood(drift, lambda img: torch.zeros(B, 50, dtype=torch.float64), x_u8, u,
                                                lambda: probe, dequantization="uniform", rtol=1e-8, atol=1e-8)
    data = tr.encode(x_u8.double()).reshape(B, -1) + 2 * (u - 0.5) / 256
    var0 = s2 + (1 - s2) / (1 + math.exp(-gmin))
    var1 = s2 + (1 - s2) / (1 + math.exp(-gmax))
    expect = tr.prior_logp(data * math.sqrt(var1 / var0)) + 0.5 * 3072 * math.log(var1 / var0)
    assert log_q is None and torch.allclose(aux, torch.zeros(B, dtype=torch.float64), atol=1e-12) and nfev > 10
    assert torch.allclose(log_p, expect, rtol=1e-7)
    # offsets of the reference: uniform -> log2(128) = 7 bits; tn, num_is = 1 includes the entropy correction
    assert tr.bpd_offset("uniform", 1) == 7.0
    ls = 0.5 * (-13.3 - math.log1p(math.exp(-13.3)))
    assert abs(tr.bpd_offset("tn", 20) + ls / math.log(2)) < 1e-12
    assert abs(tr.bpd_offset("tn", 1) + (0.5 * (1 + math.log(2 * math.pi)) - 0.01522 + ls) / math.log(2)) < 1e-12
    emb = tr.logits_to_embeddings(torch.tensor(rng.standard_normal((4, 50))))
    assert torch.equal(emb.sum(dim=1), torch.full((4,), 15.0, dtype=torch.float64))


def test_golden_sampler_and_ode_fixture():
    z = np.load(os.path.join(GOLD, "sampler_ode.npz"))
    tt = lambda k: torch.tensor(z[k])
    for kind in ("velocity", "epsilon", "input"):
        assert np.allclose(tr.ancestral_step(tt("z"), tt("net"), tt("g_t"), tt("g_s"), tt("eps"), kind).numpy(),
                           z[f"step_{kind}"], rtol=1e-12)
    for kind in ("velocity", "vfe", "epsilon"):
        assert np.allclose(tr.ode_drift(tt("net"), tt("z"), tt("g_t"), tt("g_p"), kind).numpy(), z[f"drift_{kind}"], rtol=1e-12)
    assert np.array_equal(tr.decode_argmax(tt("z0"), tt("g0")).numpy(), z["decoded"])
    assert np.array_equal(tr.logits_to_embeddings(tt("logits")).numpy(), z["hard_topk"])
    assert np.allclose(tr.gumbel_kl_loss(tt("logits")).numpy(), z["kl"], rtol=1e-12)
    assert np.allclose(tr.prior_logp(tt("z").reshape(2, 16, 16, 1)).numpy(), z["prior_logp"], rtol=1e-12)
    assert np.allclose([tr.bpd_offset("uniform", 1), tr.bpd_offset("tn", 1), tr.bpd_offset("tn", 20)], z["bpd_offsets"])
    y1 = tr.dopri5_fixed(lambda t, y: -y * (1 + t) + np.sin(3 * t), np.array([1.0, -0.5, 2.0]), list(z["dopri_grid"]))
    assert np.allclose(y1, z["dopri_y1"], rtol=1e-12)
    # the fixed-grid Dormand-Prince solution agrees with scipy's adaptive RK45 at a tight tolerance
    from scipy import integrate
    ref = integrate.solve_ivp(lambda t, y: -y * (1 + t) + np.sin(3 * t), (0, 1), np.array([1.0, -0.5, 2.0]), rtol=1e-10,
                              atol=1e-12, method="RK45").y[:, -1]
    assert np.allclose(y1, ref, rtol=2e-5)
