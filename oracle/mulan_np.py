"""CPU ORACLE (test infrastructure only) -- NumPy restatement of the MuLAN train / eval-BPD hot path.

  * PARITY UNPINNED: the reference (s-sahoo/MuLAN, JAX/Flax) ships no tests, golden vectors or
    fixtures, and JAX/Flax/Optax are not installable here, so this restatement cannot be checked
    against outputs of the reference itself.  It is pinned only by the analytic known-answer tests
    in tests/test_oracle_kat.py and by agreement with the independent torch restatement
    (oracle/torch_ref.py).  Third-party defaults it hard-codes (Flax 0.7.0 GroupNorm: 32 groups,
    eps 1e-6, fast variance; nn.Conv SAME/HWIO; nn.Dropout 1/keep scaling; optax.adamw) are listed
    in DESIGN.md.
  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
    The product (mulan_amd/, ldm/) never does.

Every function cites the reference lines it follows (paths relative to the reference checkout).
All randomness is an explicit input.  `dt` selects float64 (ideal) or float32 (mimics the
reference's fp32 rounding points for the large-argument sin/cos embeddings).
"""
import numpy as np

GAMMA_MIN, GAMMA_MAX = -13.3, 5.0


# ------------------------------------------------------------------------------ small helpers
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def swish(x):  # flax nn.swish = x * sigmoid(x)
    return x * sigmoid(x)


def softplus(x):  # jax.nn.softplus = logaddexp(x, 0)
    return np.logaddexp(x, 0.0)


def log_softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    return (x - m) - np.log(np.exp(x - m).sum(axis=axis, keepdims=True))


def softmax(x, axis=-1):
    return np.exp(log_softmax(x, axis))


# ------------------------------------------------------------------------------ Philox4x32-10
def philox4x32_10(seed, counter):
    """(seed u64, counter u64 array) -> [..., 4] uint32; identical to csrc/common.h."""
    counter = np.asarray(counter, dtype=np.uint64)
    c0 = (counter & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    c1 = (counter >> np.uint64(32)).astype(np.uint64)
    c2 = np.zeros_like(c0)
    c3 = np.zeros_like(c0)
    k0 = np.uint64(int(seed) & 0xFFFFFFFF)
    k1 = np.uint64((int(seed) >> 32) & 0xFFFFFFFF)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & mask
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def dropout_mask(shape, keep, seed, offset):
    """Keep-mask of flax nn.Dropout (ldm/model_vdm.py:644) as drawn by the HIP GroupNorm kernel:
    element e of the flattened tensor uses word e%4 of Philox(seed, offset + e//4)."""
    n = int(np.prod(shape))
    assert n % 4 == 0
    r = philox4x32_10(seed, np.uint64(offset) + np.arange(n // 4, dtype=np.uint64)).reshape(-1)
    thr = np.uint32(np.float64(np.float32(keep)) * 4294967296.0)
    return (r < thr).reshape(shape)


# ------------------------------------------------------------------------------ A.1 EncDec
def encode(x, vocab_size=256):
    """ldm/model_vdm.py:274-280"""
    x = np.round(np.asarray(x, dtype=np.float64))
    return 2 * ((x + .5) / vocab_size) - 1


def decode_logprobs(z, g_0, vocab_size=256):
    """ldm/model_vdm.py:282-294 -> [..., 256] log-probs"""
    g_0 = np.asarray(g_0, dtype=np.float64)
    if g_0.ndim > 0:
        g_0 = g_0[..., None]
    x_vals = encode(np.arange(vocab_size))
    inv_stdev = np.exp(-0.5 * g_0)
    logits = -0.5 * np.square((z[..., None] - x_vals) * inv_stdev)
    return log_softmax(logits)


def logprob(x, z, g_0):
    """ldm/model_vdm.py:296-303 -> [B]"""
    lp = decode_logprobs(z, g_0)
    idx = np.round(x).astype(np.int64)
    sel = np.take_along_axis(lp, idx[..., None], axis=-1)[..., 0]
    return sel.reshape(sel.shape[0], -1).sum(axis=1)


# ------------------------------------------------------------------------------ A.2 schedule
def poly_coefficients(emb, p):
    """ldm/model_mulan_epsilon.py:531-538.  p: dict of dense_{1,2,out_a,out_b,out_c} -> (kernel[in,out], bias)"""
    h = swish(emb @ p["dense_1"][0] + p["dense_1"][1])
    h = swish(h @ p["dense_2"][0] + p["dense_2"][1])
    a = h @ p["dense_out_a"][0] + p["dense_out_a"][1]
    b = h @ p["dense_out_b"][0] + p["dense_out_b"][1]
    c = 1e-3 + softplus(h @ p["dense_out_c"][0] + p["dense_out_c"][1])
    return a, b, c


def poly_gamma(a, b, c, t, gmin=GAMMA_MIN, gmax=GAMMA_MAX):
    """_eval_polynomial, ldm/model_mulan_epsilon.py:514-529 (grad_min_epsilon = 0, :491).  t: [B] or scalar"""
    t = np.reshape(np.asarray(t, dtype=a.dtype) * np.ones(a.shape[0], dtype=a.dtype), (-1, 1))
    poly = ((a ** 2) * (t ** 5) / 5.0 + (b ** 2 + 2 * a * c) * (t ** 3) / 3.0 + a * b * (t ** 4) / 2.0
            + b * c * (t ** 2) + (c ** 2) * t)
    scale = (a ** 2) / 5.0 + (b ** 2 + 2 * a * c) / 3.0 + a * b / 2.0 + b * c + c ** 2
    return gmin + (gmax - gmin) * poly / scale


def poly_gamma_grad_t(a, b, c, t, gmin=GAMMA_MIN, gmax=GAMMA_MAX):
    """_grad_t, ldm/model_mulan_epsilon.py:540-555 (== jvp of gamma wrt t, model_mulan_velocity.py:251-254)"""
    t = np.reshape(np.asarray(t, dtype=a.dtype) * np.ones(a.shape[0], dtype=a.dtype), (-1, 1))
    poly = ((a ** 2) * (t ** 4) + (b ** 2 + 2 * a * c) * (t ** 2) + a * b * (t ** 3) * 2.0 + b * c * t * 2 + c ** 2)
    scale = (a ** 2) / 5.0 + (b ** 2 + 2 * a * c) / 3.0 + a * b / 2.0 + b * c + c ** 2
    return (gmax - gmin) * poly / scale


# ------------------------------------------------------------------------------ A.3 latent
def gumbel_kl_loss(logits):
    """ldm/model_mulan_velocity.py:78-83"""
    q = softmax(logits)
    return np.sum(q * (log_softmax(logits) - np.log(1.0 / logits.shape[1])), axis=1)


def gamma_noise(raw, k, gamma_tau=10.0):
    """ldm/model_mulan_velocity.py:94-104; raw = Gamma(1/k) draws [10,B,L]"""
    beta = k / np.arange(1.0, 11.0)
    s = (raw / beta[:, None, None]).sum(axis=0)
    s = s - np.log(10.0)
    return gamma_tau * (s / k)


def topk_embedding_and_loss(logits, raw_gamma, k):
    """ldm/model_mulan_velocity.py:106-120 -> (embedding, kl, soft)"""
    kl = gumbel_kl_loss(logits)
    l = logits + gamma_noise(raw_gamma, k)
    l = l - l.mean(axis=1, keepdims=True)
    soft = l / np.linalg.norm(l, axis=1, keepdims=True)
    thr = np.sort(l, axis=1)[:, ::-1][:, k - 1]
    hard = (l >= thr[:, None]).astype(l.dtype)
    return (hard - soft) + soft, kl, soft


# ------------------------------------------------------------------------------ A.5 nets
def timestep_embedding(t, dim, dt=np.float64):
    """ldm/model_vdm.py:391-413 (timesteps *= 1000 inside)"""
    half = dim // 2
    w = np.exp(np.arange(half, dtype=dt) * dt(-(np.log(10000) / (half - 1))))
    e = (np.asarray(t, dtype=dt) * dt(1000.))[:, None] * w[None, :]
    return np.concatenate([np.sin(e), np.cos(e)], axis=1).astype(dt)


def fourier_features(z, dt=np.float64):
    """Base2FourierFeatures(start=6, stop=8), ldm/model_vdm.py:812-829 -> [..., 12]"""
    z = np.asarray(z, dtype=dt)
    w = (dt(2.) ** np.asarray([6, 7], dtype=dt)) * dt(2) * dt(np.pi)
    w = np.tile(w[None, :], (1, z.shape[-1]))[0]
    h = np.repeat(z, 2, axis=-1) * w
    return np.concatenate([np.sin(h), np.cos(h)], axis=-1)


def conv3x3(x, w, b=None):
    """flax nn.Conv((3,3)), SAME, NHWC, HWIO (ldm/model_vdm.py:633-634).  x [B,H,W,C], w [3,3,C,N]"""
    B, H, W, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    y = np.zeros((B, H, W, w.shape[3]), dtype=x.dtype)
    for kh in range(3):
        for kw in range(3):
            y += np.einsum("bhwc,cn->bhwn", xp[:, kh:kh + H, kw:kw + W, :], w[kh, kw])
    return y if b is None else y + b


def group_norm(x, scale, bias, groups=32, eps=1e-6):
    """flax nn.GroupNorm() defaults (ldm/model_vdm.py:622): stats over (H,W,C/G), var = E[x^2]-E[x]^2 >= 0"""
    B, H, W, C = x.shape
    xg = x.reshape(B, H * W, groups, C // groups)
    mean = xg.mean(axis=(1, 3), keepdims=True)
    var = np.maximum(0., np.square(xg).mean(axis=(1, 3), keepdims=True) - np.square(mean))
    y = (xg - mean) / np.sqrt(var + eps)
    return y.reshape(B, H, W, C) * scale + bias


def attn_block(x, p):
    """AttnBlock, ldm/model_vdm.py:660-701 + dot_product_attention :704-802 (1 head)"""
    B, H, W, C = x.shape
    h = group_norm(x, *p["GroupNorm_0"])
    q = h @ p["q"][0] + p["q"][1]
    k = h @ p["k"][0] + p["k"][1]
    v = h @ p["v"][0] + p["v"][1]
    q, k, v = (a.reshape(B, H * W, C) for a in (q, k, v))
    wgt = softmax(np.einsum("bqc,bkc->bqk", q / np.sqrt(C), k), axis=-1)
    o = np.einsum("bqk,bkc->bqc", wgt, v).reshape(B, H, W, C)
    return x + (o @ p["proj_out"][0] + p["proj_out"][1])


def resnet_block(x, cond, p, keep_mask=None, keep=1.0):
    """ResnetBlock, ldm/model_vdm.py:610-657 (cond [B,4E]) and ldm/ldm_unet.py:10-61 (cond [B,H,W,4E])"""
    h = swish(group_norm(x, *p["GroupNorm_0"]))
    h = conv3x3(h, *p["conv1"])
    cb = cond @ p["cond_proj"][0]
    h = h + (cb[:, None, None, :] if cb.ndim == 2 else cb)
    h = swish(group_norm(h, *p["GroupNorm_1"]))
    if keep_mask is not None:
        h = np.where(keep_mask, h / keep, 0.0)
    h = conv3x3(h, *p["conv2"])
    if "nin_shortcut" in p:
        x = x @ p["nin_shortcut"][0] + p["nin_shortcut"][1]
    return x + h


def unet_stem(z, t, conditioning, p, n_embd, n_layers, per_pixel=False, dt=np.float64, masks=None, keep=1.0):
    """Shared body of ScoreUNet (ldm/model_vdm.py:314-371), ldm UNet (ldm/ldm_unet.py:69-125) and
    UnetEncoder (ldm/model_mulan_epsilon.py:105-141): returns (h after middle, skip list, cond)."""
    B = z.shape[0]
    masks = masks or {}
    if per_pixel:
        temb = timestep_embedding(t.reshape(-1), n_embd, dt).reshape(B, 32, 32, 3 * n_embd)
        cnd = np.broadcast_to(conditioning[:, None, None, :], (B, 32, 32, conditioning.shape[1]))
        cond = np.concatenate([temb, cnd], axis=-1)
    else:
        cond = np.concatenate([timestep_embedding(t, n_embd, dt), conditioning], axis=1)
    cond = swish(cond @ p["dense0"][0] + p["dense0"][1])
    cond = swish(cond @ p["dense1"][0] + p["dense1"][1])
    h = np.concatenate([z, fourier_features(z, dt)], axis=-1)
    h = conv3x3(h, *p["conv_in"])
    hs = [h]
    for i in range(n_layers):
        name = f"down.block_{i}"
        h = resnet_block(hs[-1], cond, p[name], masks.get(name), keep)
        hs.append(h)
    h = hs[-1]
    h = resnet_block(h, cond, p["mid.block_1"], masks.get("mid.block_1"), keep)
    h = attn_block(h, p["mid.attn_1"])
    h = resnet_block(h, cond, p["mid.block_2"], masks.get("mid.block_2"), keep)
    return h, hs, cond


def score_unet(z, g_t, conditioning, p, n_embd, n_layers, per_pixel=False, gmin=GAMMA_MIN, gmax=GAMMA_MAX,
               dt=np.float64, masks=None, keep=1.0):
    """ScoreUNet.__call__ ldm/model_vdm.py:314-388 / ldm_unet.UNet.__call__ ldm/ldm_unet.py:69-142"""
    masks = masks or {}
    t = (np.asarray(g_t, dtype=dt) - dt(gmin)) / dt(gmax - gmin)
    h, hs, cond = unet_stem(z, t, conditioning, p, n_embd, n_layers, per_pixel, dt, masks, keep)
    for i in range(n_layers + 1):
        name = f"up.block_{i}"
        h = resnet_block(np.concatenate([h, hs.pop()], axis=-1), cond, p[name], masks.get(name), keep)
    assert not hs
    h = swish(group_norm(h, *p["GroupNorm_0"]))
    return conv3x3(h, *p["conv_out"]) + z


def unet_encoder(f, p, n_embd, n_layers, dt=np.float64, masks=None, keep=1.0):
    """UnetEncoder.__call__ ldm/model_mulan_epsilon.py:101-154 (t = 0, conditioning = 0)"""
    B = f.shape[0]
    h, _, _ = unet_stem(f, np.zeros(B), np.zeros((B, 1)), p, n_embd, n_layers, False, dt, masks, keep)
    h = swish(group_norm(h, *p["GroupNorm_0"]))
    h = conv3x3(h, *p["conv_out"])
    h = swish(h.reshape(B, -1))
    return h @ p["dense_layer_final"][0] + p["dense_layer_final"][1]


# ------------------------------------------------------------------------------ A.4 ELBO
def antithetic_t(t0, n):
    """ldm/model_mulan_velocity.py:196-198"""
    return np.mod(t0 + np.arange(0., 1., step=1. / n), 1.)


def elbo_pre(x, g_0, g_1, g_t, eps_0, eps):
    """ldm/model_mulan_velocity.py:208,219-236 -> (f, z_t, loss_recon, loss_klz, var0, var1)"""
    f = encode(x)
    var_t, var_0, var_1 = sigmoid(g_t), sigmoid(g_0), sigmoid(g_1)
    z_0_rescaled = f + np.exp(0.5 * g_0) * eps_0
    loss_recon = -logprob(x, z_0_rescaled, g_0)
    loss_klz = 0.5 * np.sum(((1. - var_1) * np.square(f) + var_1 - np.log(var_1) - 1.).reshape(x.shape[0], -1), axis=1)
    z_t = np.sqrt(1. - var_t) * f + np.sqrt(var_t) * eps
    return f, z_t, loss_recon, loss_klz, var_0.mean(), var_1.mean()


def diffusion_loss_velocity(f, g_t, g_t_grad, eps, z_t, net, velocity_from_epsilon=False):
    """ldm/model_mulan_velocity.py:246-260"""
    var_t = sigmoid(g_t)
    v_hat = net
    if velocity_from_epsilon:
        v_hat = -np.exp(0.5 * g_t) * z_t + np.sqrt(1 + np.exp(g_t)) * net
    v_target = np.sqrt(1. - var_t) * eps - np.sqrt(var_t) * f
    return .5 * np.sum(((1 - var_t) * g_t_grad * np.square(v_target - v_hat)).reshape(f.shape[0], -1), axis=1)


def diffusion_loss_epsilon(g_t_grad, eps, eps_hat):
    """ldm/model_mulan_epsilon.py:338-347 (T = 0); also model_vdm.py:156-161 with scalar g_t_grad"""
    return .5 * np.sum((g_t_grad * np.square(eps - eps_hat)).reshape(eps.shape[0], -1), axis=1)


def bpd(loss_recon, loss_klz, loss_diff, n_dims=3072):
    """Experiment_VDM.loss_fn, ldm/experiment_vdm.py:62-66"""
    r = 1. / (n_dims * np.log(2.))
    return (loss_recon.mean() + loss_klz.mean() + loss_diff.mean()) * r


def mulan_forward(params, cfg, x, t0, raw_gamma, eps_0, eps, dt=np.float64, enc_masks=None, score_masks=None,
                  keep=1.0):
    """VDM.__call__ of ldm/model_mulan_velocity.py:188-268 / ldm/model_mulan_epsilon.py:280-363 (T = 0).
    cfg keys: vdm_type, n_embd, n_layer, forward_n_layer, latent_k, unet_type, velocity_from_epsilon."""
    B = x.shape[0]
    x = x.reshape(B, 32, 32, 3)
    t = antithetic_t(t0, B)
    f = encode(x)
    logits = unet_encoder(f, params["encoder_model"], cfg["n_embd"], cfg["forward_n_layer"], dt, enc_masks, keep)
    emb, kl_z, _ = topk_embedding_and_loss(logits, raw_gamma, cfg["latent_k"])
    a, b, c = poly_coefficients(emb, params["gamma"])
    g_0 = poly_gamma(a, b, c, np.zeros(B)).reshape(f.shape)
    g_1 = poly_gamma(a, b, c, np.ones(B)).reshape(f.shape)
    g_t = poly_gamma(a, b, c, t).reshape(f.shape)
    g_p = poly_gamma_grad_t(a, b, c, t).reshape(f.shape)
    f, z_t, loss_recon, loss_klz, var0, var1 = elbo_pre(x, g_0, g_1, g_t, eps_0, eps)
    per_pixel = cfg.get("unet_type", "vdm") == "ldm"
    g_in = g_t if per_pixel else g_t.reshape(B, -1).mean(axis=1)
    net = score_unet(z_t, g_in, emb, params["score_model"], cfg["n_embd"], cfg["n_layer"], per_pixel, dt=dt,
                     masks=score_masks, keep=keep)
    if cfg["vdm_type"] == "mulan_velocity":
        loss_diff = diffusion_loss_velocity(f, g_t, g_p, eps, z_t, net, cfg.get("velocity_from_epsilon", False))
    else:
        loss_diff = diffusion_loss_epsilon(g_p, eps, net)
    out = dict(loss_recon=loss_recon, loss_klz=kl_z + loss_klz, loss_diff=loss_diff, var_0=var0, var_1=var1)
    out["bpd"] = bpd(out["loss_recon"], out["loss_klz"], out["loss_diff"])
    out["aux"] = dict(logits=logits, emb=emb, z_t=z_t, net=net, g_t=g_t, g_p=g_p)
    return out


# ------------------------------------------------------------------------------ A.6 optimiser
def lr_schedule(step, lr=2e-4, warmup=100, decay=False, total=None):
    """ldm/experiment.py:106-129.  optax.linear_schedule(init, end, n) is polynomial_schedule with power 1:
    value(count) = (init - end) * (1 - clip(count, 0, n) / n) + end, and the CONSTANT init when n <= 0.  Without
    lr_decay the schedule is the warm-up ramp alone; with it, join_schedules switches at `warmup` to the decay ramp
    evaluated at count - warmup."""
    def linear(init, end, n, count):
        if n <= 0:
            return init
        frac = 1.0 - min(max(count, 0), n) / n
        return (init - end) * frac + end
    if decay and step >= warmup:
        return linear(lr, 0.0, total - warmup, step - warmup)
    return linear(0.0, lr, warmup, step)


def adamw_ema_step(p, g, m, v, ema, lr, step, decay_mask, b1=0.9, b2=0.99, eps=1e-8, wd=0.01, ema_rate=0.9999):
    """optax.adamw (scale_by_adam -> add_decayed_weights(mask) -> scale(-lr)), ldm/experiment.py:132-182,
    and the EMA of ldm/train_state.py:88-95.  `step` is the 1-based Adam count."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    u = (m / (1 - b1 ** step)) / (np.sqrt(v / (1 - b2 ** step)) + eps)
    u = u + wd * p * decay_mask
    p = p - lr * u
    ema = ema + (1. - ema_rate) * (p - ema)
    return p, m, v, ema
