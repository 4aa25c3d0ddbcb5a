"""CPU ORACLE (test infrastructure only) -- torch restatement of the same path as oracle/mulan_np.py.

PARITY UNPINNED (see oracle/mulan_np.py header).  Purpose: (1) float64 autograd gradients to check the
hand-written HIP backward kernels, (2) the fp32 multi-threaded CPU "port" timed as `cpu_baseline` by
bench.py.  Only tests/, __graft_entry__.smoke() and bench.py may import it; the product never does.

Parameters are a Flax-layout tree: {'score_model': {...}, 'encoder_model': {...}, 'gamma': {...}} with
leaves {'kernel': [..in, out], 'bias': [out]} / {'scale','bias'} exactly as the reference checkpoint stores
them (ldm/experiment.py:165-168).
"""
import math

import torch
import torch.nn.functional as F

GAMMA_MIN, GAMMA_MAX = -13.3, 5.0


def swish(x):
    return x * torch.sigmoid(x)


def dense(x, p):
    y = x @ p["kernel"]
    return y + p["bias"] if "bias" in p else y


def conv3x3(x, p):
    """flax nn.Conv((3,3)) SAME NHWC/HWIO (ldm/model_vdm.py:633-634)"""
    w = p["kernel"].permute(3, 2, 0, 1).contiguous()   # HWIO -> OIHW
    y = F.conv2d(x.permute(0, 3, 1, 2), w, p.get("bias"), padding=1)
    return y.permute(0, 2, 3, 1)


def group_norm(x, p, groups=32, eps=1e-6):
    """flax nn.GroupNorm() (ldm/model_vdm.py:622): fast variance, eps 1e-6"""
    B, H, W, C = x.shape
    xg = x.reshape(B, H * W, groups, C // groups)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = torch.clamp((xg * xg).mean(dim=(1, 3), keepdim=True) - mean * mean, min=0.)
    y = (xg - mean) * torch.rsqrt(var + eps)
    return y.reshape(B, H, W, C) * p["scale"] + p["bias"]


def timestep_embedding(t, dim):
    """ldm/model_vdm.py:391-413"""
    half = dim // 2
    w = torch.exp(torch.arange(half, dtype=t.dtype) * (-(math.log(10000) / (half - 1))))
    e = (t * 1000.)[:, None] * w[None, :]
    return torch.cat([torch.sin(e), torch.cos(e)], dim=1)


def fourier_features(z):
    """ldm/model_vdm.py:812-829 (start=6, stop=8)"""
    w = (2. ** torch.tensor([6., 7.], dtype=z.dtype)) * 2 * math.pi
    w = w.repeat(z.shape[-1])
    h = torch.repeat_interleave(z, 2, dim=-1) * w
    return torch.cat([torch.sin(h), torch.cos(h)], dim=-1)


def attn_block(x, p):
    """ldm/model_vdm.py:660-701"""
    B, H, W, C = x.shape
    h = group_norm(x, p["GroupNorm_0"])
    q, k, v = (dense(h, p[n]).reshape(B, H * W, C) for n in ("q", "k", "v"))
    wgt = torch.softmax(torch.einsum("bqc,bkc->bqk", q / math.sqrt(C), k), dim=-1)
    o = torch.einsum("bqk,bkc->bqc", wgt, v).reshape(B, H, W, C)
    return x + dense(o, p["proj_out"])


def resnet_block(x, cond, p, keep_mask=None, keep=1.0):
    """ldm/model_vdm.py:610-657; ldm/ldm_unet.py:10-61"""
    h = conv3x3(swish(group_norm(x, p["GroupNorm_0"])), p["conv1"])
    cb = cond @ p["cond_proj"]["kernel"]
    h = h + (cb[:, None, None, :] if cb.dim() == 2 else cb)
    h = swish(group_norm(h, p["GroupNorm_1"]))
    if keep_mask is not None:
        h = torch.where(keep_mask, h / keep, torch.zeros_like(h))
    h = conv3x3(h, p["conv2"])
    if "nin_shortcut" in p:
        x = dense(x, p["nin_shortcut"])
    return x + h


def unet_stem(z, t, conditioning, p, n_embd, n_layers, per_pixel=False, masks=None, keep=1.0):
    B = z.shape[0]
    masks = masks or {}
    if per_pixel:
        temb = timestep_embedding(t.reshape(-1), n_embd).reshape(B, 32, 32, 3 * n_embd)
        cnd = conditioning[:, None, None, :].expand(B, 32, 32, conditioning.shape[1])
        cond = torch.cat([temb, cnd], dim=-1)
    else:
        cond = torch.cat([timestep_embedding(t, n_embd), conditioning], dim=1)
    cond = swish(dense(cond, p["dense0"]))
    cond = swish(dense(cond, p["dense1"]))
    h = conv3x3(torch.cat([z, fourier_features(z)], dim=-1), p["conv_in"])
    hs = [h]
    for i in range(n_layers):
        n = f"down.block_{i}"
        h = resnet_block(hs[-1], cond, p[n], masks.get(n), keep)
        if f"down.attn_{i}" in p:               # config.with_attention (ldm/model_vdm.py:356-357)
            h = attn_block(h, p[f"down.attn_{i}"])
        hs.append(h)
    h = resnet_block(hs[-1], cond, p["mid.block_1"], masks.get("mid.block_1"), keep)
    h = attn_block(h, p["mid.attn_1"])
    h = resnet_block(h, cond, p["mid.block_2"], masks.get("mid.block_2"), keep)
    return h, hs, cond


def score_unet(z, g_t, conditioning, p, n_embd, n_layers, per_pixel=False, gmin=GAMMA_MIN, gmax=GAMMA_MAX,
               masks=None, keep=1.0):
    """ldm/model_vdm.py:314-388; ldm/ldm_unet.py:69-142"""
    masks = masks or {}
    t = (g_t - gmin) / (gmax - gmin)
    h, hs, cond = unet_stem(z, t, conditioning, p, n_embd, n_layers, per_pixel, masks, keep)
    for i in range(n_layers + 1):
        n = f"up.block_{i}"
        h = resnet_block(torch.cat([h, hs.pop()], dim=-1), cond, p[n], masks.get(n), keep)
        if f"up.attn_{i}" in p:                 # config.with_attention (ldm/model_vdm.py:371-372)
            h = attn_block(h, p[f"up.attn_{i}"])
    h = swish(group_norm(h, p["GroupNorm_0"]))
    return conv3x3(h, p["conv_out"]) + z


def unet_encoder(f, p, n_embd, n_layers, masks=None, keep=1.0):
    """ldm/model_mulan_epsilon.py:101-154"""
    B = f.shape[0]
    h, _, _ = unet_stem(f, torch.zeros(B, dtype=f.dtype), torch.zeros(B, 1, dtype=f.dtype), p, n_embd, n_layers,
                        False, masks, keep)
    h = conv3x3(swish(group_norm(h, p["GroupNorm_0"])), p["conv_out"])
    return dense(swish(h.reshape(B, -1)), p["dense_layer_final"])


def encode(x):
    """ldm/model_vdm.py:274-280"""
    return 2 * ((torch.round(x) + .5) / 256) - 1


def logprob(x_int, z, g_0):
    """ldm/model_vdm.py:282-303"""
    vals = encode(torch.arange(256, dtype=z.dtype))
    logits = -0.5 * torch.square((z[..., None] - vals) * torch.exp(-0.5 * g_0)[..., None])
    lp = torch.log_softmax(logits, dim=-1)
    sel = torch.gather(lp, -1, x_int[..., None].long())[..., 0]
    return sel.reshape(sel.shape[0], -1).sum(dim=1)


def poly_coefficients(emb, p):
    """ldm/model_mulan_epsilon.py:531-538"""
    h = swish(dense(emb, p["dense_1"]))
    h = swish(dense(h, p["dense_2"]))
    return dense(h, p["dense_out_a"]), dense(h, p["dense_out_b"]), 1e-3 + F.softplus(dense(h, p["dense_out_c"]))


def poly_gamma(a, b, c, t, gmin=GAMMA_MIN, gmax=GAMMA_MAX):
    """ldm/model_mulan_epsilon.py:514-529"""
    t = t.reshape(-1, 1)
    poly = (a ** 2) * t ** 5 / 5.0 + (b ** 2 + 2 * a * c) * t ** 3 / 3.0 + a * b * t ** 4 / 2.0 + b * c * t ** 2 + c ** 2 * t
    scale = (a ** 2) / 5.0 + (b ** 2 + 2 * a * c) / 3.0 + a * b / 2.0 + b * c + c ** 2
    return gmin + (gmax - gmin) * poly / scale


def poly_gamma_grad_t(a, b, c, t, gmin=GAMMA_MIN, gmax=GAMMA_MAX):
    """ldm/model_mulan_epsilon.py:540-555"""
    t = t.reshape(-1, 1)
    poly = (a ** 2) * t ** 4 + (b ** 2 + 2 * a * c) * t ** 2 + a * b * t ** 3 * 2.0 + b * c * t * 2 + c ** 2
    scale = (a ** 2) / 5.0 + (b ** 2 + 2 * a * c) / 3.0 + a * b / 2.0 + b * c + c ** 2
    return (gmax - gmin) * poly / scale


def topk_embedding_and_loss(logits, raw_gamma, k, tau=10.0, gumbel=None):
    """ldm/model_mulan_velocity.py:78-120; gumbel given: topk_noise_type 'gumbel' (ldm/model_mulan_epsilon.py:236-239),
    the noise is added as it is"""
    L = logits.shape[1]
    q = torch.softmax(logits, dim=1)
    kl = torch.sum(q * (torch.log_softmax(logits, dim=1) - math.log(1.0 / L)), dim=1)
    if gumbel is not None:
        l = logits + gumbel
    else:
        beta = k / torch.arange(1., 11., dtype=logits.dtype)
        s = (raw_gamma / beta[:, None, None]).sum(dim=0) - math.log(10.0)
        l = logits + tau * (s / k)
    l = l - l.mean(dim=1, keepdim=True)
    soft = l / torch.linalg.norm(l, dim=1, keepdim=True)
    thr = torch.topk(l, k, dim=1).values[:, -1]
    hard = (l >= thr[:, None]).to(l.dtype)
    return (hard - soft).detach() + soft, kl


def mulan_forward(params, cfg, x_u8, t0, raw_gamma, eps_0, eps, enc_masks=None, score_masks=None, keep=1.0,
                  dtype=torch.float64, t=None, gumbel=None):
    """VDM.__call__ (ldm/model_mulan_velocity.py:188-268, ldm/model_mulan_epsilon.py:280-363, T = 0) +
    Experiment_VDM.loss_fn BPD (ldm/experiment_vdm.py:62-66)."""
    B = x_u8.shape[0]
    x = x_u8.reshape(B, 32, 32, 3)
    if t is None:                       # antithetic_time_sampling (ldm/model_mulan_velocity.py:196-198); else t given
        t = torch.remainder(t0 + torch.arange(B, dtype=dtype) / B, 1.)
    if cfg.get("n_timesteps", 0) > 0:   # ldm/model_mulan_epsilon.py:295-297
        t = torch.ceil(t * cfg["n_timesteps"]) / cfg["n_timesteps"]
    f = encode(x.to(dtype))
    logits = unet_encoder(f, params["encoder_model"], cfg["n_embd"], cfg["forward_n_layer"], enc_masks, keep)
    emb, kl_z = topk_embedding_and_loss(logits, raw_gamma, cfg["latent_k"], gumbel=gumbel)
    a, b, c = poly_coefficients(emb, params["gamma"])
    shp = f.shape
    g_0 = poly_gamma(a, b, c, torch.zeros(B, dtype=dtype)).reshape(shp)
    g_1 = poly_gamma(a, b, c, torch.ones(B, dtype=dtype)).reshape(shp)
    g_t = poly_gamma(a, b, c, t).reshape(shp)
    g_p = poly_gamma_grad_t(a, b, c, t).reshape(shp)
    var_t, var_0, var_1 = torch.sigmoid(g_t), torch.sigmoid(g_0), torch.sigmoid(g_1)
    z_0 = f + torch.exp(0.5 * g_0) * eps_0
    loss_recon = -logprob(x, z_0, g_0)
    loss_klz = 0.5 * ((1. - var_1) * f * f + var_1 - torch.log(var_1) - 1.).reshape(B, -1).sum(dim=1)
    z_t = torch.sqrt(1. - var_t) * f + torch.sqrt(var_t) * eps
    per_pixel = cfg.get("unet_type", "vdm") == "ldm"
    g_in = g_t if per_pixel else g_t.reshape(B, -1).mean(dim=1)
    net = score_unet(z_t, g_in, emb, params["score_model"], cfg["n_embd"], cfg["n_layer"], per_pixel,
                     masks=score_masks, keep=keep)
    if cfg["vdm_type"] == "mulan_velocity":
        v_hat = net
        if cfg.get("velocity_from_epsilon", False):
            v_hat = -torch.exp(0.5 * g_t) * z_t + torch.sqrt(1 + torch.exp(g_t)) * net
        v_target = torch.sqrt(1. - var_t) * eps - torch.sqrt(var_t) * f
        loss_diff = .5 * ((1 - var_t) * g_p * (v_target - v_hat) ** 2).reshape(B, -1).sum(dim=1)
    elif cfg.get("n_timesteps", 0) == 0:
        loss_diff = .5 * (g_p * (eps - net) ** 2).reshape(B, -1).sum(dim=1)
    else:   # ldm/model_mulan_epsilon.py:348-355
        T = cfg["n_timesteps"]
        g_s = poly_gamma(a, b, c, t - 1. / T).reshape(shp)
        loss_diff = .5 * T * (torch.expm1(g_t - g_s) * (eps - net) ** 2).reshape(B, -1).sum(dim=1)
    klz = kl_z + loss_klz
    r = 1. / (3072 * math.log(2.))
    return dict(loss_recon=loss_recon, loss_klz=klz, loss_diff=loss_diff, var_0=var_0.mean(), var_1=var_1.mean(),
                bpd=(loss_recon.mean() + klz.mean() + loss_diff.mean()) * r,
                aux=dict(logits=logits, emb=emb, z_t=z_t, net=net, g_t=g_t, g_p=g_p))


# ------------------------------------------------------------------------------ ancestral sampler
def deterministic_embedding(B, latent_size, latent_k, dtype=torch.float64):
    """VDM._get_deterministic_embedding, latent_type 'topk' (ldm/model_mulan_velocity.py:270-279)"""
    return torch.cat([torch.ones(B, latent_k, dtype=dtype), torch.zeros(B, latent_size - latent_k, dtype=dtype)], dim=1)


def ancestral_step(z_t, net, g_t, g_s, eps, kind):
    """the closed-form part of VDM.sample (ldm/model_mulan_velocity.py:335-350 `velocity`, ldm/model_mulan_epsilon.py:
    400-406 `epsilon`, ldm/model_vdm.py:197-210 `epsilon` / `input`); g broadcastable against z_t"""
    a, b, c = torch.sigmoid(-g_s), torch.sigmoid(-g_t), -torch.expm1(g_s - g_t)
    sigma_t, alpha_t = torch.sqrt(torch.sigmoid(g_t)), torch.sqrt(torch.sigmoid(-g_t))
    if kind == "velocity":
        eps_hat = net * alpha_t + sigma_t * z_t
    elif kind == "input":
        eps_hat = (z_t - alpha_t * net) / sigma_t
    else:
        eps_hat = net
    return torch.sqrt(a / b) * (z_t - sigma_t * c * eps_hat) + torch.sqrt((1. - a) * c) * eps


def decode_argmax(z_0, g_0):
    """VDM.generate_x with sample_softmax False (ldm/model_mulan_velocity.py:352-368, ldm/model_vdm.py:212-227):
    argmax of EncDec.decode (ldm/model_vdm.py:282-296) at z_0 / sqrt(1 - sigmoid(g_0))"""
    z = z_0 / torch.sqrt(1. - torch.sigmoid(g_0))
    vals = encode(torch.arange(256, dtype=z.dtype))
    logits = -0.5 * torch.square((z[..., None] - vals) * torch.exp(-0.5 * g_0)[..., None])
    return torch.argmax(torch.log_softmax(logits, dim=-1), dim=-1)


def step_gain(g_t, g_s, kind):
    """|d z_s / d net| of ancestral_step: how far an error of the network output is carried into z_s (test budgets)"""
    a, b, c = torch.sigmoid(-g_s), torch.sigmoid(-g_t), -torch.expm1(g_s - g_t)
    k = torch.sqrt(a / b) * torch.sqrt(torch.sigmoid(g_t)) * c
    if kind == "velocity":
        k = k * torch.sqrt(b)
    elif kind == "input":
        k = k * torch.sqrt(b) / torch.sqrt(torch.sigmoid(g_t))
    return float(k.max())


def mulan_sample_loop(params, cfg, z_init, eps_list, dtype=torch.float64, trajectory=False):
    """Experiment_VDM.sample_fn (ldm/experiment_vdm.py:80-110) with T = len(eps_list) and the per-step noise given
    (the reference draws it from fold_in(rng, i)): returns (z_0, uint8 samples) [+ per-step z_t, network output
    scale x step_gain when trajectory=True]"""
    B, T = z_init.shape[0], len(eps_list)
    shp = (B, 32, 32, 3)
    emb = deterministic_embedding(B, cfg.get("latent_size", 50), cfg["latent_k"], dtype)
    a, b, c = poly_coefficients(emb, params["gamma"])
    per_pixel = cfg.get("unet_type", "vdm") == "ldm"
    kind = "velocity" if cfg["vdm_type"] == "mulan_velocity" else "epsilon"
    z = z_init.reshape(shp).to(dtype)
    traj, budget = [z], []
    for i in range(T):
        g_t = poly_gamma(a, b, c, torch.full((B,), (T - i) / T, dtype=dtype)).reshape(shp)
        g_s = poly_gamma(a, b, c, torch.full((B,), (T - i - 1) / T, dtype=dtype)).reshape(shp)
        g_in = g_t if per_pixel else g_t.reshape(B, -1).mean(dim=1)
        net = score_unet(z, g_in, emb, params["score_model"], cfg["n_embd"], cfg["n_layer"], per_pixel)
        z = ancestral_step(z, net, g_t, g_s, eps_list[i].reshape(shp).to(dtype), kind)
        traj.append(z)
        budget.append(step_gain(g_t, g_s, kind) * float(net.abs().max()))
    g_0 = poly_gamma(a, b, c, torch.zeros(B, dtype=dtype)).reshape(shp)
    if trajectory:
        return z, decode_argmax(z, g_0), traj, budget
    return z, decode_argmax(z, g_0)


def plain_sample_loop(params, cfg, z_init, eps_list, gmin=GAMMA_MIN, gmax=GAMMA_MAX, dtype=torch.float64,
                      trajectory=False):
    """the same loop for model_vdm.VDM (ldm/model_vdm.py:182-227)"""
    B, T = z_init.shape[0], len(eps_list)
    shp = (B, 32, 32, 3)
    if "gamma" in params:
        w, b0 = torch.abs(params["gamma"]["w"]), params["gamma"]["b"]
        gamma = lambda tt: (b0 + w * tt).reshape(())
    else:
        gamma = lambda tt: torch.tensor(gmin + (gmax - gmin) * tt, dtype=dtype)
    z = z_init.reshape(shp).to(dtype)
    kind = "input" if cfg.get("reparam_type") == "input" else "epsilon"
    traj, budget = [z], []
    for i in range(T):
        g_t, g_s = gamma((T - i) / T), gamma((T - i - 1) / T)
        net = score_unet(z, g_t * torch.ones(B, dtype=dtype), torch.zeros(B, 1, dtype=dtype), params["score_model"],
                         cfg["n_embd"], cfg["n_layer"], gmin=gmin, gmax=gmax)
        z = ancestral_step(z, net, g_t, g_s, eps_list[i].reshape(shp).to(dtype), kind)
        traj.append(z)
        budget.append(step_gain(g_t, g_s, kind) * float(net.abs().max()))
    g_0 = gamma(0.) * torch.ones(shp, dtype=dtype)
    if trajectory:
        return z, decode_argmax(z, g_0), traj, budget
    return z, decode_argmax(z, g_0)


# ------------------------------------------------------------------------------ exact likelihood (ODE)
def logits_to_embeddings(logits, k=15):
    """ldm/notebook_utils.py:548-551"""
    kth = torch.topk(logits, k, dim=-1).values[:, -1:]
    return (logits >= kth).to(logits.dtype)


def gumbel_kl_loss(logits):
    """ldm/notebook_utils.py:222-229"""
    q, logq = torch.softmax(logits, dim=-1), torch.log_softmax(logits, dim=-1)
    return (q * (logq - math.log(1.0 / logits.shape[-1]))).sum(dim=-1)


def ode_drift(net, x, g_t, g_p, kind, high_precision=False):
    """the closed form around the network in VDM.reverse_ode: kind 'velocity' / 'vfe' (velocity_from_epsilon)
    (ldm/model_mulan_velocity.py:403-421), 'epsilon' (ldm/model_mulan_epsilon.py:471-478); high_precision: the
    jnp.where selects of model_mulan_velocity.py:410-417 / model_mulan_epsilon.py:472-475"""
    sigma = torch.sqrt(torch.sigmoid(g_t))
    if high_precision:
        sigma = torch.where(torch.sigmoid(g_t) <= 1e-3, torch.exp(g_t / 2), sigma)
    if kind == "epsilon":
        return 0.5 * (-sigma * x + net) * sigma * g_p
    v_hat = net
    if kind == "vfe":
        v_hat = -torch.exp(0.5 * g_t) * x + torch.sqrt(1 + torch.exp(g_t)) * net
    alpha = torch.sqrt(1 - torch.sigmoid(g_t))
    if high_precision:
        alpha = torch.where(1 - torch.sigmoid(g_t) <= 1e-3, torch.exp(-g_t / 2), alpha)
    return v_hat * (0.5 * alpha * sigma * g_p)


def reverse_ode(params, cfg, x, emb, t, high_precision=False):
    """VDM.reverse_ode (ldm/model_mulan_velocity.py:393-421 incl. velocity_from_epsilon; ldm/model_mulan_epsilon.py:
    459-478); x [B,32,32,3], t float"""
    B = x.shape[0]
    a, b, c = poly_coefficients(emb, params["gamma"])
    tt = torch.full((B,), float(t), dtype=x.dtype)
    g_t = poly_gamma(a, b, c, tt).reshape(x.shape)
    g_p = poly_gamma_grad_t(a, b, c, tt).reshape(x.shape)
    per_pixel = cfg.get("unet_type", "vdm") == "ldm"
    g_in = g_t if per_pixel else g_t.reshape(B, -1).mean(dim=1)
    net = score_unet(x, g_in, emb, params["score_model"], cfg["n_embd"], cfg["n_layer"], per_pixel)
    if cfg["vdm_type"] == "mulan_velocity":
        return ode_drift(net, x, g_t, g_p, "vfe" if cfg.get("velocity_from_epsilon", False) else "velocity", high_precision)
    return ode_drift(net, x, g_t, g_p, "epsilon", high_precision)


def plain_reverse_ode(params, cfg, x, emb, t, gmin=GAMMA_MIN, gmax=GAMMA_MAX):
    """model_vdm.VDM.reverse_ode + sde (ldm/model_vdm.py:229-260)"""
    B = x.shape[0]
    if "gamma" in params:
        w, b0 = torch.abs(params["gamma"]["w"]), params["gamma"]["b"]
        g_t, g_p = (b0 + w * t).reshape(()), w.reshape(())
    else:
        g_t, g_p = torch.tensor(gmin + (gmax - gmin) * t, dtype=x.dtype), torch.tensor(gmax - gmin, dtype=x.dtype)
    drift = -0.5 * torch.sigmoid(g_t) * g_p * x
    diffusion_sqr = torch.sigmoid(g_t) * g_p
    eps_hat = score_unet(x, g_t * torch.ones(B, dtype=x.dtype), emb[:, :1], params["score_model"], cfg["n_embd"],
                         cfg["n_layer"], gmin=gmin, gmax=gmax)
    score_hat = -eps_hat / torch.sqrt(torch.sigmoid(g_t))
    return drift - 0.5 * diffusion_sqr * score_hat


def value_div(fn, x, hutch):
    """notebook_utils._get_value_div_fn (:203-215): (f(x), sum_i (d sum(f * hutch) / d x)_i hutch_i per sample)"""
    x = x.detach().requires_grad_(True)
    f = fn(x)
    (g,) = torch.autograd.grad((f * hutch).sum(), x)
    return f.detach(), (g * hutch).reshape(x.shape[0], -1).sum(dim=1)


def prior_logp(z):
    """notebook_utils._prior_logp (:218-221)"""
    n = z[0].numel()
    return -0.5 * n * math.log(2 * math.pi) - 0.5 * (z.reshape(z.shape[0], -1) ** 2).sum(dim=1)


def bpd_offset(dequantization, num_is):
    """notebook_utils._get_bpd_offset (:446-458)"""
    if dequantization == "uniform":
        return math.log2(128)
    gt = -13.3
    log_sigma = 0.5 * (gt - math.log1p(math.exp(gt)))
    extra = 0.5 * (1 + math.log(2 * math.pi)) - 0.01522 if num_is == 1 else 0.0
    return -(extra + log_sigma) / math.log(2)


def dopri5_fixed(fun, y0, t_grid):
    """Dormand-Prince 5(4) steps (the tableau of scipy's RK45 = Dormand & Prince 1980) on a prescribed grid, 5th
    order solution, no error control; fun(t, y) -> dy/dt (numpy float64)"""
    import numpy as np
    c = [0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1]
    a = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
         [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
         [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656]]
    b = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]
    y = np.asarray(y0, dtype=np.float64)
    for t, t_new in zip(t_grid[:-1], t_grid[1:]):
        h = t_new - t
        k = [fun(t, y)]
        for s in range(1, 6):
            k.append(fun(t + c[s] * h, y + h * sum(a[s][j] * k[j] for j in range(s))))
        y = y + h * sum(b[j] * k[j] for j in range(6))
    return y


def ode_likelihood(drift_fn, encoder_fn, x_u8, u, probes, dequantization="tn", rtol=1e-5, atol=1e-5,
                   dtype=torch.float64, t_grid=None):
    """likelihood_fn of get_ode_likelihood_fn (ldm/notebook_utils.py:303-371) with the dequantisation noise `u`
    (U[0,1) for 'uniform', truncated normal for 'tn') and the Hutchinson probes (callable, one per function
    evaluation) given.  drift_fn(x, emb, t) = reverse_ode, encoder_fn(images_int) = apply_encoder.  Integrated by
    scipy.integrate.solve_ivp(method='RK45') exactly like the reference.
    Returns (log_p [B], log_q_eps [B] | None, aux_loss [B], nfev)"""
    import numpy as np
    from scipy import integrate
    B = x_u8.shape[0]
    shp = (B, 32, 32, 3)
    data = encode(x_u8.reshape(shp).to(dtype))
    u = u.reshape(shp).to(dtype)
    if dequantization == "uniform":
        noise, log_q_eps = 2 * (u - 0.5) / 256, None
    else:
        log_q_eps = prior_logp(u) - 3072 * math.log(0.9974613)
        noise = u * math.exp(0.5 * -13.3)
    data = data + noise
    logits = encoder_fn(torch.round(torch.clamp(128 * (data + 1) - 0.5, 0, 255)))
    aux = gumbel_kl_loss(logits)
    emb = logits_to_embeddings(logits)

    def ode_func(t, y):
        # the reference hands the network the fp32-rounded state (_from_flattened_numpy, notebook_utils.py:198-200)
        xt = torch.tensor(y[:-B], dtype=torch.float32).to(dtype).reshape(shp)
        f, div = value_div(lambda xx: drift_fn(xx, emb, t), xt, probes().reshape(shp).to(dtype))
        return np.concatenate([f.reshape(-1).numpy(), div.numpy()])

    init = np.concatenate([data.reshape(-1).numpy(), np.zeros(B)])
    if t_grid is not None:          # test hook: fixed steps instead of the adaptive controller
        zp, nfev = dopri5_fixed(ode_func, init, t_grid), 6 * (len(t_grid) - 1)
    else:
        sol = integrate.solve_ivp(ode_func, (0, 1), init, rtol=rtol, atol=atol, method="RK45")
        zp, nfev = sol.y[:, -1], sol.nfev
    z = torch.tensor(zp[:-B], dtype=dtype).reshape(shp)
    log_p = prior_logp(z) + torch.tensor(zp[-B:], dtype=dtype)
    return log_p, log_q_eps, aux, nfev


def ode_sample(drift_fn, emb, prior, rtol=1e-5, atol=1e-5, dtype=torch.float64):
    """sample_fn of get_sample_fn (ldm/notebook_utils.py:411-441) with the embedding and the prior draw given:
    solve_ivp over (1, 0) of the drift alone (the reference's divergence value is discarded).  -> (z, nfev)"""
    import numpy as np
    from scipy import integrate
    shp = (prior.shape[0], 32, 32, 3)

    def ode_func(t, y):
        xt = torch.tensor(y, dtype=torch.float32).to(dtype).reshape(shp)
        with torch.no_grad():
            return drift_fn(xt, emb, t).reshape(-1).numpy()

    sol = integrate.solve_ivp(ode_func, (1, 0), prior.reshape(-1).to(dtype).numpy(), rtol=rtol, atol=atol, method="RK45")
    return torch.tensor(sol.y[:, -1], dtype=dtype).reshape(shp), sol.nfev


# ------------------------------------------------------------------------------ parameter trees
def tree_map(fn, tree):
    return {k: tree_map(fn, v) if isinstance(v, dict) else fn(v) for k, v in tree.items()}


def tree_leaves(tree, prefix=()):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from tree_leaves(v, prefix + (k,))
        else:
            yield prefix + (k,), v


def to_np_tuples(tree):
    """Flax-layout tree -> the tuple layout oracle/mulan_np.py consumes."""
    import numpy as np
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict) and all(not isinstance(x, dict) for x in v.values()):
            g = lambda a: np.asarray(a.detach().cpu().double().numpy() if torch.is_tensor(a) else a, dtype=np.float64)
            if "scale" in v:
                out[k] = (g(v["scale"]), g(v["bias"]))
            else:
                out[k] = (g(v["kernel"]), g(v["bias"]) if "bias" in v else 0.0)
        else:
            out[k] = to_np_tuples(v)
    return out


def init_params(cfg, seed=0, dtype=torch.float64, zero_init=False, latent=50):
    """Random Flax-layout parameter tree for a (small) config.  With zero_init=False the tensors the
    reference zero-initialises (conv2, cond_proj, proj_out, conv_out, dense_out_a) get small random
    values instead, so every gradient path is exercised."""
    g = torch.Generator().manual_seed(seed)
    E = cfg["n_embd"]

    def rnd(*shape, scale=None):
        fan_in = shape[-2] if len(shape) > 1 else shape[0]
        if len(shape) == 4:
            fan_in = shape[0] * shape[1] * shape[2]
        s = scale if scale is not None else 1.0 / math.sqrt(fan_in)
        return (torch.randn(*shape, generator=g, dtype=torch.float64) * s).to(dtype)

    def z_or_r(*shape, scale=None):
        return torch.zeros(*shape, dtype=dtype) if zero_init else rnd(*shape, scale=scale)

    def gn(C):
        return {"scale": (1.0 + 0.1 * torch.randn(C, generator=g, dtype=torch.float64)).to(dtype),
                "bias": rnd(C, scale=0.1)}

    def block(cin, cout, cond_dim):
        p = {"GroupNorm_0": gn(cin), "conv1": {"kernel": rnd(3, 3, cin, cout), "bias": rnd(cout, scale=0.1)},
             "cond_proj": {"kernel": z_or_r(cond_dim, cout)}, "GroupNorm_1": gn(cout),
             "conv2": {"kernel": z_or_r(3, 3, cout, cout), "bias": rnd(cout, scale=0.1)}}
        if cin != cout:
            p["nin_shortcut"] = {"kernel": rnd(cin, cout), "bias": rnd(cout, scale=0.1)}
        return p

    def attn(C):
        p = {"GroupNorm_0": gn(C)}
        for n in ("q", "k", "v"):
            p[n] = {"kernel": rnd(C, C), "bias": rnd(C, scale=0.1)}
        p["proj_out"] = {"kernel": z_or_r(C, C), "bias": rnd(C, scale=0.1)}
        return p

    def unet(n_layers, K, out_ch, with_up, temb_dim):
        p = {"dense0": {"kernel": rnd(temb_dim + K, 4 * E), "bias": rnd(4 * E, scale=0.1)},
             "dense1": {"kernel": rnd(4 * E, 4 * E), "bias": rnd(4 * E, scale=0.1)},
             "conv_in": {"kernel": rnd(3, 3, 15, E), "bias": rnd(E, scale=0.1)}}
        for i in range(n_layers):
            p[f"down.block_{i}"] = block(E, E, 4 * E)
            if cfg.get("with_attention", False):
                p[f"down.attn_{i}"] = attn(E)
        p["mid.block_1"] = block(E, E, 4 * E)
        p["mid.attn_1"] = attn(E)
        p["mid.block_2"] = block(E, E, 4 * E)
        if with_up:
            for i in range(n_layers + 1):
                p[f"up.block_{i}"] = block(2 * E, E, 4 * E)
                if cfg.get("with_attention", False):
                    p[f"up.attn_{i}"] = attn(E)
        p["GroupNorm_0"] = gn(E)
        p["conv_out"] = {"kernel": z_or_r(3, 3, E, out_ch), "bias": rnd(out_ch, scale=0.1)}
        return p

    per_pixel = cfg.get("unet_type", "vdm") == "ldm"
    score = unet(cfg["n_layer"], latent, 3, True, 3 * E if per_pixel else E)
    enc = unet(cfg["forward_n_layer"], 1, 1, False, E)
    enc["dense_layer_final"] = {"kernel": rnd(1024, latent), "bias": rnd(latent, scale=0.1)}
    Dm = 3072
    gamma = {"dense_1": {"kernel": rnd(latent, Dm), "bias": rnd(Dm, scale=0.1)},
             "dense_2": {"kernel": rnd(Dm, Dm), "bias": rnd(Dm, scale=0.1)},
             "dense_out_a": {"kernel": z_or_r(Dm, Dm), "bias": z_or_r(Dm, scale=0.1)},
             "dense_out_b": {"kernel": rnd(Dm, Dm), "bias": rnd(Dm, scale=0.1)},
             "dense_out_c": {"kernel": rnd(Dm, Dm), "bias": rnd(Dm, scale=0.1)}}
    return {"score_model": score, "encoder_model": enc, "gamma": gamma}


def nnet_gamma(p, t):
    """NoiseSchedule_NNet.__call__ (ldm/model_vdm.py:492-509) with DenseMonotone = |kernel| (:581-598); t [B]"""
    t2 = t.reshape(-1, 1)
    h = t2 @ torch.abs(p["l1"]["kernel"]) + p["l1"]["bias"]
    _h = (2. * (t2 - .5)) @ torch.abs(p["l2"]["kernel"]) + p["l2"]["bias"]
    _h = 2 * (torch.sigmoid(_h) - .5)
    _h = (_h @ torch.abs(p["l3"]["kernel"])) / p["l2"]["kernel"].shape[1]
    return (h + _h).squeeze(-1)


def plain_vdm_forward(params, cfg, x_u8, t0, eps_0, eps, gmin=GAMMA_MIN, gmax=GAMMA_MAX, dtype=torch.float64,
                      score_masks=None, keep=1.0):
    """model_vdm.VDM.__call__ (ldm/model_vdm.py:110-180) with gamma_type 'fixed' (:462-468), 'learnable_scalar'
    (:418-431) or 'learnable_nnet' (:471-509), epsilon prediction, T = 0 (:158-161) or T > 0 with the 'noise' weighting
    (:162-170).  score_masks / keep: the dropout keep-masks of the score U-Net in training mode (deterministic=False is
    handed to the score model at :153-157)."""
    B = x_u8.shape[0]
    x = x_u8.reshape(B, 32, 32, 3)
    T = cfg.get("n_timesteps", 0)
    t = torch.remainder(t0 + torch.arange(B, dtype=dtype) / B, 1.)
    if T > 0:
        t = torch.ceil(t * T) / T
    if "gamma" in params and "l1" in params["gamma"]:          # gamma_type learnable_nnet
        gamma = lambda tt: nnet_gamma(params["gamma"], tt)
        # jax.jvp(self.gamma, (t,), (ones,)) (ldm/model_vdm.py:160): the t-derivative, differentiable in the parameters
        tt_ = t.detach().clone().requires_grad_(True)
        slope = torch.autograd.grad(gamma(tt_).sum(), tt_, create_graph=True)[0]
    elif "gamma" in params:
        w, b = torch.abs(params["gamma"]["w"]), params["gamma"]["b"]
        gamma = lambda tt: b + w * tt
        slope = w.expand(B)
    else:
        gamma = lambda tt: gmin + (gmax - gmin) * tt
        slope = torch.full((B,), gmax - gmin, dtype=dtype)
    g_0, g_1, g_t = gamma(torch.zeros(B, dtype=dtype)), gamma(torch.ones(B, dtype=dtype)), gamma(t)
    bc = lambda g: g[:, None, None, None]
    f = encode(x.to(dtype))
    var_0, var_1, var_t = torch.sigmoid(g_0), torch.sigmoid(g_1), torch.sigmoid(g_t)
    z_0 = f + torch.exp(0.5 * bc(g_0)) * eps_0
    loss_recon = -logprob(x, z_0, bc(g_0) * torch.ones_like(f))
    v1 = bc(var_1) * torch.ones_like(f)
    loss_klz = 0.5 * ((1. - v1) * f * f + v1 - torch.log(v1) - 1.).reshape(B, -1).sum(dim=1)
    z_t = torch.sqrt(1. - bc(var_t)) * f + torch.sqrt(bc(var_t)) * eps
    eps_hat = score_unet(z_t, g_t, torch.zeros(B, 1, dtype=dtype), params["score_model"], cfg["n_embd"], cfg["n_layer"],
                         gmin=gmin, gmax=gmax, masks=score_masks, keep=keep)
    mse = ((eps - eps_hat) ** 2).reshape(B, -1).sum(dim=1)
    if T == 0:
        loss_diff = .5 * slope * mse
    else:
        loss_diff = .5 * T * torch.expm1(g_t - gamma(t - 1. / T)) * mse
    r = 1. / (3072 * math.log(2.))
    return dict(loss_recon=loss_recon, loss_klz=loss_klz, loss_diff=loss_diff, var_0=var_0.mean(), var_1=var_1.mean(),
                bpd=(loss_recon.mean() + loss_klz.mean() + loss_diff.mean()) * r)
