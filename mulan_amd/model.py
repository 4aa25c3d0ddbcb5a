"""MuLAN / VDM models over the HIP kernels (host-side mirror of the reference's Flax modules).

Functional, Flax-like API: `VDM(config).init(rng)` returns a parameter tree whose names and shapes
are the reference's (the only layout difference is conv_in, stored with a 16th
zero input channel for aligned loads), `VDM.apply(params, images, labels, conditioning, step, rngs,
deterministic)` returns a VDMOutput.  Mirrors:
  VDMConfig / VDMOutput ............ ldm/model_vdm.py:33-92
  ScoreUNet / ldm UNet / UnetEncoder  ldm/model_vdm.py:309-388, ldm/ldm_unet.py:64-142,
                                     ldm/model_mulan_epsilon.py:101-154
  ResnetBlock / AttnBlock .......... ldm/model_vdm.py:610-701, ldm/ldm_unet.py:10-61
  NoiseSchedule_polynomial_fixedend  ldm/model_mulan_epsilon.py:481-613
  NoiseSchedule_{Scalar,FixedLinear,NNet} (plain VDM) ldm/model_vdm.py:418-509
  VDM.__call__ ...................... ldm/model_mulan_velocity.py:188-268,
                                     ldm/model_mulan_epsilon.py:280-363, ldm/model_vdm.py:110-180
All tensor math runs in libmulan_hip.so (mulan_amd.ops); there is no CPU path.
"""
import dataclasses
import functools
import math
import os
import numpy as np
from typing import Any, Optional

import torch

from . import ops
from .rng import Key

HW = 1024
D = 3072


# ----------------------------------------------------------------------------- config / output
@dataclasses.dataclass(frozen=True)
class VDMConfig:
    """Same fields as the reference dataclass (ldm/model_vdm.py:33-82); unknown keys raise TypeError."""
    vocab_size: int
    sample_softmax: bool
    antithetic_time_sampling: bool
    with_fourier_features: bool
    with_attention: bool
    gamma_type: str
    gamma_min: float
    gamma_max: float
    sm_n_timesteps: int
    sm_n_embd: int
    sm_n_layer: int
    sm_pdrop: float
    sm_kernel_init: Any = None
    forward_n_layer: int = 4
    forward_type: int = 1
    sigma_type: str = 'learnable_scalar'
    sigma_min: float = 0
    sigma_max: float = 20.0
    sm_mult: float = 1.0
    sigma_prior: float = 1.0
    blur_noise: bool = False
    sigma_recons_type: str = 'sigmoid'
    loss_type: str = 'recons'
    reparam_type: str = 'noise'
    nn_input: str = 'gamma'
    condition: str = 'label'
    latent_size: int = 10
    epsilon: float = 0.0
    encoder: str = 'cnn'
    model_time: bool = False
    monotone_layer: str = 'dense_monotone'
    latent_type: str = 'gumbel'
    z_conditioning: bool = False
    importance_sampling: bool = False
    velocity_from_epsilon: bool = False
    unet_type: str = 'vdm'
    topk_noise_type: str = 'gamma'
    latent_k: int = 15
    trace_matching: bool = False


@dataclasses.dataclass
class VDMOutput:
    loss_recon: torch.Tensor   # [B]
    loss_klz: torch.Tensor     # [B]
    loss_diff: torch.Tensor    # [B]
    var_0: torch.Tensor
    var_1: torch.Tensor


# ----------------------------------------------------------------------------- initialisers
def _lecun_normal(gen, shape, fan_in):
    """jax.nn.initializers.lecun_normal (flax Dense/Conv default): truncated normal, var = 1/fan_in"""
    std = math.sqrt(1.0 / fan_in) / .87962566103423978
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return t


def _dense(gen, cin, cout, bias=True, zero=False):
    p = {"kernel": torch.zeros(cin, cout) if zero else _lecun_normal(gen, (cin, cout), cin)}
    if bias:
        p["bias"] = torch.zeros(cout)
    return p


def _conv(gen, cin, cout, zero=False, pad_in=0):
    k = torch.zeros(3, 3, cin, cout) if zero else _lecun_normal(gen, (3, 3, cin, cout), 9 * cin)
    if pad_in:
        k = torch.cat([k, torch.zeros(3, 3, pad_in, cout)], dim=2)
    return {"kernel": k, "bias": torch.zeros(cout)}


def _gn(C):
    return {"scale": torch.ones(C), "bias": torch.zeros(C)}


def _resblock_init(gen, cin, cout, cond_dim):
    p = {"GroupNorm_0": _gn(cin), "conv1": _conv(gen, cin, cout), "cond_proj": _dense(gen, cond_dim, cout, False, True),
         "GroupNorm_1": _gn(cout), "conv2": _conv(gen, cout, cout, zero=True)}
    if cin != cout:
        p["nin_shortcut"] = _dense(gen, cin, cout)
    return p


def _attn_init(gen, C):
    return {"GroupNorm_0": _gn(C), "q": _dense(gen, C, C), "k": _dense(gen, C, C), "v": _dense(gen, C, C),
            "proj_out": _dense(gen, C, C, zero=True)}


def _unet_init(gen, E, n_layers, cond_in, out_ch, with_up, with_attention=False):
    p = {"dense0": _dense(gen, cond_in, 4 * E), "dense1": _dense(gen, 4 * E, 4 * E),
         "conv_in": _conv(gen, 15, E, pad_in=1)}
    for i in range(n_layers):
        p[f"down.block_{i}"] = _resblock_init(gen, E, E, 4 * E)
        if with_attention:                      # AttnBlock after every block (ldm/model_vdm.py:356-357, 371-372)
            p[f"down.attn_{i}"] = _attn_init(gen, E)
    p["mid.block_1"] = _resblock_init(gen, E, E, 4 * E)
    p["mid.attn_1"] = _attn_init(gen, E)
    p["mid.block_2"] = _resblock_init(gen, E, E, 4 * E)
    if with_up:
        for i in range(n_layers + 1):
            p[f"up.block_{i}"] = _resblock_init(gen, 2 * E, E, 4 * E)
            if with_attention:
                p[f"up.attn_{i}"] = _attn_init(gen, E)
    p["GroupNorm_0"] = _gn(E)
    p["conv_out"] = _conv(gen, E, out_ch, zero=True)
    return p


# ----------------------------------------------------------------------------- blocks
class _Drop:
    """Per-apply dropout context: one Philox seed, a distinct counter window per dropout site."""

    def __init__(self, key: Optional[Key], rate: float):
        self.on = key is not None and rate > 0.0
        self.keep = 1.0 - rate if self.on else 1.0
        # a Key bound to a device slot (rng.Key.dev, graph capture) hands the kernels the slot instead of the value
        self.seed = 0 if key is None else (key.dev if getattr(key, "dev", None) is not None else key.v)
        self.site = 0

    def next(self):
        self.site += 1
        return self.keep, self.seed, self.site << 34


SAMPLER_GRAPH = os.environ.get("MULAN_SAMPLER_GRAPH", "1") == "1"


class GraphedReverseStep:
    """One reverse step of the ancestral sampler (VDM.sample / conditional_sample, ldm/model_mulan_velocity.py:281-350)
    captured as a HIP graph and replayed T times.  What changes from step to step reaches the kernels through static
    buffers written before each replay: z_t, the step's noise (drawn by the same Philox call as the eager step, into the
    buffer) and the two times t, s; the replayed step is bit-identical to conditional_sample
    (tests/test_gpu_sampler.py::test_replayed_reverse_step_equals_the_eager_step).  The weights must stay as they are
    while the stepper lives (the caller holds the ParamPacker refresh, as the eager loop does)."""

    def __init__(self, model, params, B, device, embedding, conditioning, coeffs, T):
        self.model, self.T, self.B = model, T, B
        f32 = dict(device=device, dtype=torch.float32)
        self.z_in, self.eps = torch.zeros((B, D), **f32), torch.zeros((B, D), **f32)
        self.t, self.s = torch.full((B,), 1.0, **f32), torch.full((B,), 1.0 - 1.0 / T, **f32)
        with torch.no_grad():
            for _ in range(2):         # eager first: every kernel configured, the allocator warm
                model._reverse_step(params, self.z_in, self.eps, self.t, self.s, embedding, conditioning, coeffs)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # (thread_local: the RCCL watchdog thread of a multi-rank job keeps polling its events during the capture)
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.z_out = model._reverse_step(params, self.z_in, self.eps, self.t, self.s, embedding, conditioning, coeffs)

    def step(self, i, z, rng):
        B, T = self.B, self.T
        self.z_in.copy_(z.reshape(B, D))
        ops.randn(None, rng.fold_in(i).v, 0, self.z_in.device, out=self.eps)
        self.t.fill_(float(np.float32((T - i) / T)))
        self.s.fill_(float(np.float32((T - i - 1) / T)))
        self.graph.replay()
        return self.z_out.view(z.shape)


ODE_GRAPH = os.environ.get("MULAN_ODE_GRAPH", "1") == "1"      # A/B switch: 0 = every function evaluation issued eagerly


class GraphedOdeFunction:
    """One function evaluation of the probability-flow ODE -- VDM.reverse_ode and, for the likelihood, the Hutchinson term
    hutch^T (d drift / d x) hutch through the U-Net's input gradient (ldm/model_mulan_velocity.py:393-421,
    ldm/notebook_utils.py:203-215) -- captured as a HIP graph and replayed at every stage of every Dormand-Prince step.
    Eagerly an evaluation is ~1100 dependent launches that the host needs ~25 ms to issue whatever the batch (measured:
    25.1 ms per evaluation at 16 AND at 64 images); replayed, the device's own time is left.  What changes between
    evaluations reaches the kernels through static buffers: the state x, the probe, the time (a [B] tensor read by
    mulan_poly_gamma).  Same kernels, same order: results are bit-identical to the eager evaluation
    (tests/test_gpu_ode.py::test_replayed_ode_function_equals_the_eager_one).  The weights must stay as they are while the
    object lives (the caller holds the ParamPacker refresh)."""

    def __init__(self, model, params, ctx, B, device, with_div, high_precision=False):
        self.model, self.B, self.with_div, self.params = model, B, with_div, params
        f32 = dict(device=device, dtype=torch.float32)
        # the per-batch context (embedding, schedule coefficients) in buffers of its own: set_context() re-targets a
        # captured graph at the next batch / importance sample instead of capturing again
        self.ctx = dict(emb=ctx["emb"].detach().clone(), coeffs=tuple(c.detach().clone() for c in ctx["coeffs"]))
        ctx = self.ctx
        self.x, self.tt = torch.zeros((B, D), **f32), torch.full((B,), 0.5, **f32)
        self.probe = torch.ones((B, D), **f32) if with_div else None
        self.drift = torch.empty((B, D), **f32)
        self.div = torch.empty((B,), **f32) if with_div else None
        run = lambda: model.reverse_ode(params, self.x, ctx, None, self.probe, drift_out=self.drift, div_out=self.div,
                                        tt=self.tt, **({"high_precision": True} if high_precision else {}))
        for _ in range(2):             # eager first: every kernel configured, the allocator warm
            run()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            run()

    def set_context(self, ctx):
        self.ctx["emb"].copy_(ctx["emb"])
        for dst, src in zip(self.ctx["coeffs"], ctx["coeffs"]):
            dst.copy_(src)

    def __call__(self, t, x, probe, drift_out, div_out=None):
        self.x.copy_(x.reshape(self.B, D))
        if self.with_div:
            self.probe.copy_(probe.reshape(self.B, D))
        self.tt.fill_(float(np.float32(t)))
        self.graph.replay()
        drift_out.copy_(self.drift.view_as(drift_out))
        if self.with_div:
            div_out.copy_(self.div.view_as(div_out))


def ode_function(model, params, ctx, B, device, with_div, graph=None, cache=None, high_precision=False):
    """-> f(t, x [B, 3072], probe | None, drift_out, div_out | None): VDM.reverse_ode as the ODE solvers call it; a
    replayed HIP graph by default (MULAN_ODE_GRAPH), eager where the capture fails (logged).  cache (a dict the caller
    keeps while the weights stay as they are): the captured graph of a batch size is re-used for the next batch / the
    next importance sample with its context re-targeted (the packed weights live in persistent buffers, ops.ParamPacker)"""
    if graph is None:
        graph = ODE_GRAPH
    if graph and torch.device(device).type == "cuda":
        # the key names what the captured kernels were chosen by: batch, with / without the divergence term and the
        # arithmetic mode (as GraphedStep.matches does; the developer tuning words of mulan_set_tuning are not product
        # state: a caller that flips them drops the cache); the parameter tree is held by the entry and compared by identity -- an id() alone could be re-used by a new tree once the old one is freed
        # (ADVICE r05).  drop_ode_graphs(cache) releases the graph's private pool (a forward + backward pass of
        # activations) when the caller goes back to training.
        key = (B, bool(with_div), ops.CONV_MODE, bool(high_precision))
        hit = cache.get(key) if cache is not None else None
        if hit is not None and hit.params is params:
            hit.set_context(ctx)
            return hit
        try:
            g = GraphedOdeFunction(model, params, ctx, B, device, with_div, high_precision)
            g.params = params
            if cache is not None:
                cache.clear()            # (one graph at a time: its pool holds a forward + backward pass of activations)
                cache[key] = g
            return g
        except Exception as e:      # noqa: BLE001  the replay is an optimisation: fall back loudly
            import logging
            logging.getLogger("mulan").warning("HIP-graph capture of the ODE function evaluation failed (%s: %s); "
                                               "evaluating eagerly", type(e).__name__, e)
    hp = {"high_precision": True} if high_precision else {}
    return lambda t, x, probe, drift_out, div_out=None: model.reverse_ode(params, x, ctx, t, probe, drift_out=drift_out,
                                                                          div_out=div_out, **hp)


def drop_ode_graphs(cache):
    """Release the captured ODE function evaluations of `cache` (the dict handed to ode_function) and their private
    memory pools: call it when the likelihood evaluation is over and training continues."""
    if cache:
        cache.clear()
        torch.cuda.empty_cache()


def resnet_block(p, x1, x2, cond, drop):
    """ResnetBlock.__call__ (ldm/model_vdm.py:618-657 / ldm/ldm_unet.py:18-61) on [x1|x2]."""
    # Each (norm + swish [+ dropout]) -> conv pair is one op: the normalised tensor goes from the GroupNorm kernel to the
    # convolution as split fp16 planes (ops.GnConv3x3Fn; two ops with an fp32 tensor in between where that does not
    # apply).  s1 / s2 alias x1 / x2 for the skip path: their gradients are added inside the GroupNorm backward kernel.
    cb = ops.cond_proj(cond, p["cond_proj"]["kernel"])       # [B,E] or [B,1024,E]
    h, s1, s2 = ops.gn_conv3x3(x1, x2, p["GroupNorm_0"]["scale"], p["GroupNorm_0"]["bias"], p["conv1"]["kernel"],
                               p["conv1"]["bias"], cbias=cb, act=True, skip=True)
    keep, seed, off = drop.next()
    if "nin_shortcut" in p:
        if x2 is None:
            res = ops.linear(s1, p["nin_shortcut"]["kernel"], p["nin_shortcut"]["bias"])
        else:
            res = ops.linear2(s1, s2, p["nin_shortcut"]["kernel"], p["nin_shortcut"]["bias"])
    else:
        res = s1
    # (h = conv1's output has no other consumer: its gradient may travel as split planes only, ops.GRAD_PLANES)
    return ops.gn_conv3x3(h, None, p["GroupNorm_1"]["scale"], p["GroupNorm_1"]["bias"], p["conv2"]["kernel"],
                          p["conv2"]["bias"], res=res, act=True, keep=keep, seed=seed, offset=off, x1_grad_planes=True)


def attn_block(p, x):
    """AttnBlock.__call__ with num_heads = 1 (ldm/model_vdm.py:668-701)."""
    h = ops.group_norm(x, None, p["GroupNorm_0"]["scale"], p["GroupNorm_0"]["bias"], act=False)
    q = ops.linear(h, p["q"]["kernel"], p["q"]["bias"])
    k = ops.linear(h, p["k"]["kernel"], p["k"]["bias"])
    v = ops.linear(h, p["v"]["kernel"], p["v"]["bias"])
    o = ops.attention(q, k, v)
    return ops.linear(o, p["proj_out"]["kernel"], p["proj_out"]["bias"], res=x)


def _unet_stem(p, z, t, conditioning, E, n_layers, per_pixel, with_attention, drop, with_skips=True):
    """conv_in .. mid.block_2 shared by ScoreUNet, ldm UNet and UnetEncoder."""
    if per_pixel:
        # ldm_unet.py:82-90: temb over t.reshape(-1) -> [B,32,32,3E]; dense0(concat[temb, c]) is evaluated as
        # temb @ W[:3E] + broadcast(c @ W[3E:] + b) so the per-pixel concat is never materialised.
        B = z.shape[0]
        none = torch.empty((B * D, 0), device=z.device, dtype=torch.float32)
        temb3 = ops.cond_input(t.reshape(-1), none, E).view(B * HW, 3 * E)
        w0 = p["dense0"]["kernel"]
        cs = ops.linear(conditioning, w0[3 * E:], p["dense0"]["bias"])
        cond = ops.silu(ops.linear(temb3, w0[:3 * E], None, res=ops.row_broadcast(cs, HW)))
    else:
        cond = ops.silu(ops.linear(ops.cond_input(t, conditioning, E), p["dense0"]["kernel"], p["dense0"]["bias"]))
    cond = ops.silu(ops.linear(cond, p["dense1"]["kernel"], p["dense1"]["bias"]))
    if per_pixel:
        cond = cond.view(z.shape[0], HW, -1)
    h = ops.conv3x3(ops.fourier_features(z), p["conv_in"]["kernel"], p["conv_in"]["bias"])
    # every level output feeds the next block and (in the U-Nets with an up path) a skip connection: ops.tee /
    # ops.tee_take arrange for the two gradients to be summed inside the next block's GroupNorm backward kernel
    h, skip = ops.tee(h) if with_skips else (h, h)
    hs = [skip]
    for i in range(n_layers):
        h = resnet_block(p[f"down.block_{i}"], h, None, cond, drop)
        if with_attention:
            h = attn_block(p[f"down.attn_{i}"], h)
        h, skip = ops.tee(h) if with_skips else (h, h)
        hs.append(skip)
    h = resnet_block(p["mid.block_1"], h, None, cond, drop)
    h = attn_block(p["mid.attn_1"], h)
    h = resnet_block(p["mid.block_2"], h, None, cond, drop)
    return h, hs, cond


def score_unet(p, cfg, z, g_t, conditioning, drop, time=False):
    """ScoreUNet.__call__ (ldm/model_vdm.py:314-388) / ldm_unet.UNet.__call__ (ldm/ldm_unet.py:69-142).
    z [B,1024,3]; g_t [B] (vdm) or [B,1024,3] (ldm); conditioning [B,K]."""
    per_pixel = cfg.unet_type == 'ldm'
    E, L = cfg.sm_n_embd, cfg.sm_n_layer
    t = g_t if (time and not per_pixel) else (g_t - cfg.gamma_min) / (cfg.gamma_max - cfg.gamma_min)
    h, hs, cond = _unet_stem(p, z, t, conditioning, E, L, per_pixel, cfg.with_attention, drop)
    for i in range(L + 1):
        h = resnet_block(p[f"up.block_{i}"], h, ops.tee_take(hs.pop()), cond, drop)
        if cfg.with_attention:
            h = attn_block(p[f"up.attn_{i}"], h)
    assert not hs
    h = ops.group_norm(h, None, p["GroupNorm_0"]["scale"], p["GroupNorm_0"]["bias"], act=True)
    return ops.conv3x3(h, p["conv_out"]["kernel"], p["conv_out"]["bias"], res=z)


def unet_encoder(p, cfg, f, drop):
    """UnetEncoder.__call__ (ldm/model_mulan_epsilon.py:101-154): f [B,1024,3] -> logits [B,latent_size]"""
    B = f.shape[0]
    E = cfg.sm_n_embd
    t = torch.zeros(B, device=f.device)
    conditioning = torch.zeros((B, 1), device=f.device)
    h, _, _ = _unet_stem(p, f, t, conditioning, E, cfg.forward_n_layer, False, cfg.with_attention, drop, with_skips=False)
    h = ops.group_norm(h, None, p["GroupNorm_0"]["scale"], p["GroupNorm_0"]["bias"], act=True)
    h = ops.conv3x3(h, p["conv_out"]["kernel"], p["conv_out"]["bias"])          # [B,1024,1]
    h = ops.silu(h.view(B, HW))
    return ops.linear(h, p["dense_layer_final"]["kernel"], p["dense_layer_final"]["bias"])


def poly_coefficients(p, emb):
    """NoiseSchedule_polynomial_fixedend._compute_coefficients (ldm/model_mulan_epsilon.py:531-538)"""
    h = ops.silu(ops.linear(emb, p["dense_1"]["kernel"], p["dense_1"]["bias"]))
    h = ops.silu(ops.linear(h, p["dense_2"]["kernel"], p["dense_2"]["bias"]))
    a = ops.linear(h, p["dense_out_a"]["kernel"], p["dense_out_a"]["bias"])
    b = ops.linear(h, p["dense_out_b"]["kernel"], p["dense_out_b"]["bias"])
    c = ops.softplus_shift(ops.linear(h, p["dense_out_c"]["kernel"], p["dense_out_c"]["bias"]), 1e-3)
    return a, b, c


def encode_images(images_u8):
    """EncDec.encode (ldm/model_vdm.py:274-280) as a device tensor [B,1024,3] (exact in fp32)."""
    return ops.encode_u8(images_u8.reshape(images_u8.shape[0], HW, 3))


# ----------------------------------------------------------------------------- VDM variants
class _VDMBase:
    def __init__(self, config: VDMConfig):
        self.config = config

    def _noise(self, rngs, noise, B, device, need_gamma):
        """Draw (t0, raw Gamma, eps_0, eps) in the reference's make_rng('sample') order
        (ldm/model_mulan_velocity.py:195, :95-96 via :210, :223, :235) unless given explicitly."""
        noise = dict(noise or {})
        key = rngs.get("sample") if rngs else None
        gkey = "gumbel" if self.config.topk_noise_type == 'gumbel' else "gamma_raw"
        tkey = "t0" if self.config.antithetic_time_sampling else "t"
        if any(k not in noise for k in (tkey, "eps_0", "eps")) or (need_gamma and gkey not in noise):
            if key is None:
                raise ValueError("VDM.apply needs rngs={'sample': Key} or explicit noise")
            k_t, k_g, k_0, k_e = key.split(4)
            noise.setdefault("t0", k_t.uniform())
            if not self.config.antithetic_time_sampling:    # t ~ U[0,1)^B (ldm/model_mulan_velocity.py:199-200)
                noise.setdefault("t", ops.noise((B,), k_t.v, 0, device, "uniform"))
            if need_gamma:
                cfg = self.config
                if cfg.topk_noise_type == 'gumbel':
                    noise.setdefault("gumbel", ops.noise((B, cfg.latent_size), k_g.v, 0, device, "gumbel"))
                else:
                    noise.setdefault("gamma_raw", k_g.gamma(1.0 / cfg.latent_k, (10, B, cfg.latent_size), device))
            noise.setdefault("eps_0", k_0.normal((B, D), device))
            noise.setdefault("eps", k_e.normal((B, D), device))
        return noise

    def _times(self, noise, B, device):
        cfg = self.config
        T = cfg.sm_n_timesteps
        if not cfg.antithetic_time_sampling:
            t = noise["t"].to(device=device, dtype=torch.float32).reshape(B)
            return torch.ceil(t * T) / T if T > 0 else t
        t0 = noise["t0"]
        # t = mod(t0 + arange(0, 1, 1/B), 1)  (ldm/model_mulan_velocity.py:196-198), built in fp32 like jnp
        # (evaluated on the device: the same IEEE fp32 multiply / add / floor as on the host, and no blocking
        # host-to-device copy, which would make the host wait for the whole previous step)
        # (t0 may be a 0-dim fp32 device tensor: a stream-ordered parameter under HIP-graph replay)
        t0v = t0.to(device=device, dtype=torch.float32) if torch.is_tensor(t0) else float(np.float32(t0))
        t = torch.remainder(torch.arange(B, dtype=torch.float32, device=device) * float(np.float32(1.0 / B)) + t0v, 1.0)
        if T > 0:
            t = torch.ceil(t * T) / T
        return t

    def __call__(self, params, *a, **kw):
        return self.apply(params, *a, **kw)


class MulanVDM(_VDMBase):
    """model_mulan_velocity.VDM / model_mulan_epsilon.VDM selected by `parameterization`."""

    def __init__(self, config: VDMConfig, parameterization: str):
        super().__init__(config)
        assert parameterization in ("velocity", "epsilon")
        self.parameterization = parameterization
        c = config
        if c.latent_type != 'topk' or c.encoder != 'unet' or c.gamma_type != 'poly_fixedend':
            raise NotImplementedError(
                "hot path covers latent_type=topk, encoder=unet, gamma_type=poly_fixedend (the shipped configs); "
                f"got {c.latent_type}/{c.encoder}/{c.gamma_type}")
        if c.topk_noise_type not in ('gamma', 'gumbel') or (c.topk_noise_type == 'gumbel' and parameterization != "epsilon"):
            raise ValueError("topk_noise_type: 'gamma' (both models) or 'gumbel' (model_mulan_epsilon only, "
                             "ldm/model_mulan_epsilon.py:236-239)")
        if parameterization == "velocity" and c.sm_n_timesteps != 0:
            raise AssertionError("model_mulan_velocity asserts T == 0 (ldm/model_mulan_velocity.py:255)")

    def init(self, rng: Key):
        c = self.config
        gen = torch.Generator().manual_seed(rng.v & ((1 << 63) - 1))
        E = c.sm_n_embd
        K = c.latent_size if c.z_conditioning else 1
        temb = 3 * E if c.unet_type == 'ldm' else E
        score = _unet_init(gen, E, c.sm_n_layer, temb + K, 3, True, c.with_attention)
        enc = _unet_init(gen, E, c.forward_n_layer, E + 1, 1, False, c.with_attention)
        enc["dense_layer_final"] = _dense(gen, HW, c.latent_size)
        lat = c.latent_size if c.reparam_type == 'true' else 10
        gamma = {"dense_1": _dense(gen, lat, D), "dense_2": _dense(gen, D, D),
                 "dense_out_a": _dense(gen, D, D, zero=True), "dense_out_b": _dense(gen, D, D),
                 "dense_out_c": _dense(gen, D, D)}
        return {"score_model": score, "encoder_model": enc, "gamma": gamma}

    def apply(self, params, images, labels=None, conditioning=None, step=0, rngs=None, deterministic=True,
              noise=None, return_aux=False, same_image=False):
        """same_image (not in the reference): the caller vouches that every row of `images` is the same image (the dense
        variational-bound evaluator tiles one test image n_timesteps times, ldm/notebook_utils.py:181-186).  The encoder
        U-Net sees neither t nor the noise, so in evaluation mode its logits are identical for all rows: it then runs on
        one row and the logits are broadcast (7 % of the evaluator's FLOPs; same bits)."""
        cfg = self.config
        dev = images.device
        x = images.reshape(-1, D).contiguous()
        if x.dtype != torch.uint8:
            x = torch.round(x).to(torch.uint8)
        B = x.shape[0]
        noise = self._noise(rngs, noise, B, dev, cfg.reparam_type == 'true')
        t = self._times(noise, B, dev)
        f = encode_images(x)
        # rngs['dropout_pair']: the two sub-keys already split (and bound to device slots) by a graphed train step
        pair = None if deterministic else (rngs or {}).get("dropout_pair")
        drop_key = None if deterministic else (rngs or {}).get("dropout")
        if not deterministic and drop_key is None and pair is None:
            raise ValueError("training mode needs rngs['dropout']")
        k_enc, k_score = pair if pair is not None else (drop_key.split(2) if drop_key is not None else (None, None))
        if cfg.reparam_type == 'true':
            if same_image and deterministic and B > 1:
                logits = unet_encoder(params["encoder_model"], cfg, f[:1].contiguous(), _Drop(None, 0.0))
                logits = logits.expand(B, logits.shape[1]).contiguous()
            else:
                logits = unet_encoder(params["encoder_model"], cfg, f, _Drop(k_enc, cfg.sm_pdrop))
            if cfg.topk_noise_type == 'gumbel':
                emb, kl_z = ops.topk_embedding(logits, noise["gumbel"], cfg.latent_k, tau=-1.0)
            else:
                emb, kl_z = ops.topk_embedding(logits, noise["gamma_raw"], cfg.latent_k)
        else:   # ldm/model_mulan_velocity.py:212-214
            emb = torch.nn.functional.one_hot(labels.long(), 10).to(torch.float32)
            kl_z = torch.zeros(B, device=dev)
        a, b, c = poly_coefficients(params["gamma"], emb)
        g0, g1, gt, gp = ops.poly_gamma(a, b, c, t, cfg.gamma_min, cfg.gamma_max)
        zt, gbar, loss_recon, loss_klz, v0, v1 = ops.qsample(x, g0, g1, gt, noise["eps_0"], noise["eps"])
        if cfg.z_conditioning:
            cond = emb
        else:
            cond = conditioning.reshape(B, 1).to(torch.float32)
        g_in = gt.view(B, HW, 3) if cfg.unet_type == 'ldm' else gbar
        net = score_unet(params["score_model"], cfg, zt.view(B, HW, 3), g_in, cond, _Drop(k_score, cfg.sm_pdrop))
        net = net.reshape(B, D)
        T = cfg.sm_n_timesteps
        if self.parameterization == "velocity":
            mode = 1 if cfg.velocity_from_epsilon else 0
            loss_diff = ops.diffusion_loss(mode, x, gt, gp, noise["eps"], zt, net)
        elif T == 0:
            loss_diff = ops.diffusion_loss(2, x, gt, gp, noise["eps"], zt, net)
        else:
            # finite depth T (ldm/model_mulan_epsilon.py:348-355): s = t - 1/T, weight T * expm1(g_t - g_s)
            _, _, gs, _ = ops.poly_gamma(a, b, c, t - 1.0 / T, cfg.gamma_min, cfg.gamma_max)
            loss_diff = ops.diffusion_loss(2, x, gt, ops.expm1_weight(gt, gs, T), noise["eps"], zt, net)
        out = VDMOutput(loss_recon=loss_recon, loss_klz=kl_z + loss_klz, loss_diff=loss_diff, var_0=v0.mean(),
                        var_1=v1.mean())
        if return_aux:
            return out, dict(emb=emb, zt=zt, net=net, gt=gt, gp=gp, t=t, logits=logits if cfg.reparam_type == 'true' else None)
        return out


    # ---- ancestral sampler (ldm/model_mulan_velocity.py:270-368, ldm/model_mulan_epsilon.py:365-460) ----------
    def deterministic_embedding(self, B, device):
        """_get_deterministic_embedding for latent_type = topk: the first latent_k entries set"""
        c = self.config
        emb = torch.zeros((B, c.latent_size), device=device, dtype=torch.float32)
        emb[:, :c.latent_k] = 1.0
        return emb

    def sample_coefficients(self, params, embedding):
        """(a, b, c) of the per-pixel schedule for an embedding: constant over the T steps of a sampling run"""
        with torch.no_grad():
            return poly_coefficients(params["gamma"], embedding)

    def _gamma_at(self, coeffs, t_value, B, device):
        cfg = self.config
        t = torch.full((B,), float(np.float32(t_value)), device=device, dtype=torch.float32)
        return self._gamma_of(coeffs, t)                                   # [B, 3072]

    def _gamma_of(self, coeffs, t):
        """gamma at the times of the device tensor t [B] (the sampler's replayed step reads t from a static buffer)"""
        cfg = self.config
        _, _, gt, _ = ops.poly_gamma(coeffs[0], coeffs[1], coeffs[2], t, cfg.gamma_min, cfg.gamma_max)
        return gt

    def _reverse_step(self, params, z, eps, t, s, embedding, conditioning, coeffs):
        """the device work of one reverse step t -> s: z, eps [B, 3072]; t, s [B] device tensors"""
        cfg = self.config
        B = z.shape[0]
        g_t = self._gamma_of(coeffs, t)
        g_s = self._gamma_of(coeffs, s)
        cond = embedding if cfg.z_conditioning else conditioning.reshape(B, 1).to(torch.float32)
        g_in = g_t.view(B, HW, 3) if cfg.unet_type == 'ldm' else ops.rowmean(g_t)
        net = score_unet(params["score_model"], cfg, z.view(B, HW, 3), g_in, cond, _Drop(None, 0.0)).reshape(B, D)
        return ops.ancestral_step(z, net, g_t, g_s, eps, 0 if self.parameterization == "velocity" else 1)

    def conditional_sample(self, params, i, T, z_t, embedding, conditioning, rng, coeffs=None):
        """one reverse step t = (T-i)/T -> s = (T-i-1)/T given the latent embedding; z_t [B,32,32,3] (any layout with
        B x 3072 elements); rng: Key, folded with i like the reference"""
        with torch.no_grad():
            B = z_t.shape[0]
            z = z_t.reshape(B, D).contiguous()
            eps = rng.fold_in(i).normal((B, D), z.device)
            if coeffs is None:
                coeffs = self.sample_coefficients(params, embedding)
            t = torch.full((B,), float(np.float32((T - i) / T)), device=z.device, dtype=torch.float32)
            s = torch.full((B,), float(np.float32((T - i - 1) / T)), device=z.device, dtype=torch.float32)
            z_s = self._reverse_step(params, z, eps, t, s, embedding, conditioning, coeffs)
        return z_s.view(z_t.shape)

    def reverse_stepper(self, params, B, device, embedding, conditioning, coeffs, T, graph=None):
        """-> step(i, z, rng) running conditional_sample's reverse step; with `graph` (default: MULAN_SAMPLER_GRAPH, on)
        as a replayed HIP graph (GraphedReverseStep): five launches per step from the host instead of ~450.  Measured on
        an idle host it buys nothing (9.66 vs 9.67 ms per step at 16 images, 10.9 vs 10.9 at 64, 18.6 vs 18.7 at 128): below
        ~100 images the step is bound by the latency of its ~450 dependent launches (every one a single round of at most
        256 blocks), not by the host; the replay keeps it that way when the host is busy (data loading, other ranks)."""
        if graph is None:
            graph = SAMPLER_GRAPH
        if graph and torch.device(device).type == "cuda":
            try:
                return GraphedReverseStep(self, params, B, device, embedding, conditioning, coeffs, T).step
            except Exception as e:      # noqa: BLE001  the replay is an optimisation: fall back loudly
                import logging
                logging.getLogger("mulan").warning("HIP-graph capture of the sampler's reverse step failed (%s: %s); "
                                                   "sampling eagerly", type(e).__name__, e)
        return lambda i, z, rng: self.conditional_sample(params, i, T, z, embedding, conditioning, rng, coeffs)

    def sample(self, params, i, T, z_t, conditioning, rng, coeffs=None):
        emb = self.deterministic_embedding(z_t.shape[0], z_t.device)
        return self.conditional_sample(params, i, T, z_t, emb, conditioning, rng, coeffs)

    def generate_x(self, params, z_0, coeffs=None, rng=None):
        """argmax of the decoder logits, or with sample_softmax a categorical draw (rng: the 'sample' Key)"""
        cfg = self.config
        if cfg.sample_softmax and rng is None:
            raise ValueError("sample_softmax=True needs rng (the reference's make_rng('sample'))")
        with torch.no_grad():
            B = z_0.shape[0]
            if coeffs is None:
                coeffs = self.sample_coefficients(params, self.deterministic_embedding(B, z_0.device))
            g_0 = self._gamma_at(coeffs, 0.0, B, z_0.device)
            if cfg.sample_softmax:
                return ops.decode_sample(z_0.reshape(B, D), g_0, rng.v).view(B, 32, 32, 3)
            return ops.decode_argmax(z_0.reshape(B, D), g_0).view(B, 32, 32, 3)


    # ---- probability-flow ODE (ldm/model_mulan_velocity.py:51-53, 393-421; ldm/model_mulan_epsilon.py:459-478) ----
    def apply_encoder(self, params, images_u8):
        """VDM.apply_encoder: encoder logits [B, latent_size] of integer images"""
        x = images_u8.reshape(-1, D).contiguous()
        with torch.no_grad():
            return unet_encoder(params["encoder_model"], self.config, encode_images(x), _Drop(None, 0.0))

    def ode_context(self, params, images_u8):
        """per batch, constant along the ODE: hard top-k embedding of the encoder logits
        (notebook_utils.logits_to_embeddings), its KL term (_gumbel_kl_loss) and the schedule coefficients"""
        if not self.config.z_conditioning:
            raise NotImplementedError("reverse_ode hands the embedding to the score model (z_conditioning=True)")
        logits = self.apply_encoder(params, images_u8)
        emb, kl = ops.topk_hard(logits, self.config.latent_k)
        with torch.no_grad():
            coeffs = poly_coefficients(params["gamma"], emb)
        return dict(emb=emb, kl=kl, coeffs=coeffs, logits=logits)

    def ode_context_from_embedding(self, params, emb):
        """the same context for a given k-hot embedding (the ODE sampler draws it from random logits)"""
        with torch.no_grad():
            return dict(emb=emb, kl=None, coeffs=poly_coefficients(params["gamma"], emb), logits=None)

    def _ode_mode(self, high_precision=False):
        hp = 4 if high_precision else 0       # mulan_ode_drift / mulan_ode_div: mode | 4 = the high_precision selects
        if self.parameterization == "velocity":
            return (1 if self.config.velocity_from_epsilon else 0) | hp
        return 2 | hp

    def reverse_ode(self, params, x, ctx, t, hutch=None, drift_out=None, div_out=None, tt=None, high_precision=False):
        """VDM.reverse_ode at time t for x [B, 3072]; with `hutch` also the Hutchinson estimate
        hutch^T (d drift / d x) hutch per sample (notebook_utils._get_value_div_fn): returns (drift, div | None).
        tt (optional, [B] fp32 device tensor): the time as a stream-ordered parameter (GraphedOdeFunction) instead of t.
        high_precision: the alpha / sigma selects of ldm/model_mulan_velocity.py:410-417, model_mulan_epsilon.py:472-475"""
        cfg = self.config
        mode = self._ode_mode(high_precision)
        B = x.shape[0]
        a, b, c = ctx["coeffs"]
        if tt is None:
            tt = torch.full((B,), float(np.float32(t)), device=x.device, dtype=torch.float32)
        with torch.no_grad():
            _, _, gt, gp = ops.poly_gamma(a, b, c, tt, cfg.gamma_min, cfg.gamma_max)
            g_in = gt.view(B, HW, 3) if cfg.unet_type == 'ldm' else ops.rowmean(gt)
        xin = x.detach().reshape(B, D).contiguous()
        if hutch is None:
            with torch.no_grad():
                net = score_unet(params["score_model"], cfg, xin.view(B, HW, 3), g_in, ctx["emb"], _Drop(None, 0.0))
                drift, _ = ops.ode_drift(net.reshape(B, D), xin, gt, gp, None, mode, drift_out)
            return drift, None
        xin.requires_grad_(True)
        with torch.enable_grad():
            net = score_unet(params["score_model"], cfg, xin.view(B, HW, 3), g_in, ctx["emb"], _Drop(None, 0.0))
        drift, cot = ops.ode_drift(net.detach().reshape(B, D), xin.detach(), gt, gp, hutch, mode, drift_out)
        (gx,) = torch.autograd.grad(net, xin, cot.view_as(net))
        div = ops.ode_div(gx.reshape(B, D), gt, gp, hutch, mode, div_out)
        return drift, div


class PlainVDM(_VDMBase):
    """model_vdm.VDM (ldm/model_vdm.py:95-180): scalar noise schedule, epsilon prediction, T = 0 or T > 0
    with reparam_type 'noise' | 'input'.  gamma_type in {'fixed', 'learnable_scalar', 'learnable_nnet'}."""

    N_FEATURES = 1024      # NoiseSchedule_NNet.n_features

    def __init__(self, config: VDMConfig):
        super().__init__(config)
        if config.gamma_type not in ('fixed', 'learnable_scalar', 'learnable_nnet'):
            raise NotImplementedError(f"model_vdm.VDM gamma_type={config.gamma_type} "
                                      "(supported: fixed, learnable_scalar, learnable_nnet; the reference raises on "
                                      "poly_fixedend, ldm/model_vdm.py:101-108)")
        if config.unet_type != 'vdm':
            raise NotImplementedError("model_vdm.VDM always uses ScoreUNet")
        if config.sm_n_timesteps > 0 and config.reparam_type not in ('noise', 'input'):
            raise ValueError(f"model_vdm.VDM with T > 0 needs reparam_type noise | input, got {config.reparam_type} "
                             "(the reference leaves loss_diff undefined there, ldm/model_vdm.py:166-169)")

    def init(self, rng: Key):
        c = self.config
        gen = torch.Generator().manual_seed(rng.v & ((1 << 63) - 1))
        E = c.sm_n_embd
        p = {"score_model": _unet_init(gen, E, c.sm_n_layer, E + 1, 3, True, c.with_attention)}
        if c.gamma_type == 'learnable_scalar':   # NoiseSchedule_Scalar, ldm/model_vdm.py:418-431
            p["gamma"] = {"w": torch.tensor([c.gamma_max - c.gamma_min], dtype=torch.float32),
                          "b": torch.tensor([c.gamma_min], dtype=torch.float32)}
        elif c.gamma_type == 'learnable_nnet':   # NoiseSchedule_NNet, ldm/model_vdm.py:471-509 (normal init: stddev 0.01)
            F = self.N_FEATURES
            p["gamma"] = {"l1": {"kernel": torch.full((1, 1), c.gamma_max - c.gamma_min, dtype=torch.float32),
                                 "bias": torch.tensor([c.gamma_min], dtype=torch.float32)},
                          "l2": {"kernel": torch.randn((1, F), generator=gen) * 0.01, "bias": torch.zeros(F)},
                          "l3": {"kernel": torch.randn((F, 1), generator=gen) * 0.01}}
        return p

    def _gamma(self, params, t):
        """(gamma(t), d gamma / d t) for t [B]"""
        c = self.config
        if c.gamma_type == 'fixed':
            return c.gamma_min + (c.gamma_max - c.gamma_min) * t, torch.full_like(t, c.gamma_max - c.gamma_min)
        if c.gamma_type == 'learnable_nnet':
            # monotone network of one scalar input (DenseMonotone = |kernel|): [B, 1024] host-side autograd glue like the
            # two scalars below; the derivative the reference takes by jax.jvp is written out
            g = params["gamma"]
            F = self.N_FEATURES
            t2 = t.reshape(-1, 1)
            w1, w2, w3 = torch.abs(g["l1"]["kernel"]), torch.abs(g["l2"]["kernel"]), torch.abs(g["l3"]["kernel"])
            sg = torch.sigmoid((2.0 * (t2 - 0.5)) * w2 + g["l2"]["bias"])
            h = t2 * w1 + g["l1"]["bias"] + ((2.0 * (sg - 0.5)) @ w3) / F
            dh = w1 + ((4.0 * sg * (1.0 - sg) * w2) @ w3) / F
            return h.reshape(-1), dh.reshape(-1)
        w, b = params["gamma"]["w"], params["gamma"]["b"]
        aw = torch.abs(w)     # tiny [1]-element host-side autograd glue for the 2 schedule scalars
        return b + aw * t, aw.expand_as(t)

    def apply(self, params, images, labels=None, conditioning=None, step=0, rngs=None, deterministic=True,
              noise=None, return_aux=False):
        cfg = self.config
        dev = images.device
        x = images.reshape(-1, D).contiguous()
        if x.dtype != torch.uint8:
            x = torch.round(x).to(torch.uint8)
        B = x.shape[0]
        noise = self._noise(rngs, noise, B, dev, False)
        t = self._times(noise, B, dev)
        ones = torch.ones(B, device=dev)
        g0, _ = self._gamma(params, 0.0 * ones)
        g1, _ = self._gamma(params, ones)
        gt, gp = self._gamma(params, t)
        T = cfg.sm_n_timesteps
        if T > 0:   # ldm/model_vdm.py:162-170
            gs, _ = self._gamma(params, t - 1.0 / T)
            gp = T * torch.expm1(gt - gs)
            if cfg.reparam_type == 'input':
                gp = gp * torch.exp(-gt)
        zt, gbar, loss_recon, loss_klz, v0, v1 = ops.qsample(x, g0.contiguous(), g1.contiguous(), gt.contiguous(),
                                                             noise["eps_0"], noise["eps"])
        drop_key = None if deterministic else (rngs or {}).get("dropout")
        if not deterministic and drop_key is None:
            raise ValueError("training mode needs rngs['dropout']")
        cond = conditioning.reshape(B, 1).to(torch.float32)
        net = score_unet(params["score_model"], cfg, zt.view(B, HW, 3), gt, cond, _Drop(drop_key, cfg.sm_pdrop))
        loss_diff = ops.diffusion_loss(2, x, gt.contiguous(), gp.contiguous(), noise["eps"], zt, net.reshape(B, D))
        out = VDMOutput(loss_recon=loss_recon, loss_klz=loss_klz, loss_diff=loss_diff, var_0=v0.mean(), var_1=v1.mean())
        if return_aux:
            return out, dict(zt=zt, net=net, gt=gt, gp=gp, t=t)
        return out


def _plain_sample(self, params, i, T, z_t, conditioning, rng, coeffs=None):
    """model_vdm.VDM.sample (ldm/model_vdm.py:182-210)"""
    cfg = self.config
    with torch.no_grad():
        B = z_t.shape[0]
        z = z_t.reshape(B, D).contiguous()
        eps = rng.fold_in(i).normal((B, D), z.device)
        ones = torch.ones(B, device=z.device)
        g_t, _ = self._gamma(params, float(np.float32((T - i) / T)) * ones)
        g_s, _ = self._gamma(params, float(np.float32((T - i - 1) / T)) * ones)
        cond = conditioning.reshape(B, 1).to(torch.float32)
        net = score_unet(params["score_model"], cfg, z.view(B, HW, 3), g_t.contiguous(), cond, _Drop(None, 0.0))
        z_s = ops.ancestral_step(z, net.reshape(B, D), g_t.contiguous(), g_s.contiguous(), eps,
                                 2 if cfg.reparam_type == 'input' else 1)
    return z_s.view(z_t.shape)


def _plain_generate_x(self, params, z_0, coeffs=None, rng=None):
    if self.config.sample_softmax and rng is None:
        raise ValueError("sample_softmax=True needs rng (the reference's make_rng('sample'))")
    with torch.no_grad():
        B = z_0.shape[0]
        g_0, _ = self._gamma(params, torch.zeros(B, device=z_0.device))
        if self.config.sample_softmax:
            return ops.decode_sample(z_0.reshape(B, D), g_0.contiguous(), rng.v).view(B, 32, 32, 3)
        return ops.decode_argmax(z_0.reshape(B, D), g_0.contiguous()).view(B, 32, 32, 3)


def _plain_apply_encoder(self, params, images_u8):
    """model_vdm.VDM.apply_encoder (ldm/model_vdm.py:240-241): zeros"""
    return torch.zeros((images_u8.reshape(-1, D).shape[0], 50), device=images_u8.device, dtype=torch.float32)


def _plain_ode_context(self, params, images_u8):
    logits = self.apply_encoder(params, images_u8)
    emb, kl = ops.topk_hard(logits, 15)          # all-equal logits: every entry >= the 15th largest -> ones; KL = 0
    return dict(emb=emb, kl=kl, coeffs=None, logits=logits)


def _plain_reverse_ode(self, params, x, ctx, t, hutch=None, drift_out=None, div_out=None):
    """model_vdm.VDM.reverse_ode (ldm/model_vdm.py:243-260): drift - 0.5 g^2 score with score = -eps_hat / sigma;
    the score model is conditioned on embeddings[:, :1] like the reference"""
    cfg = self.config
    B = x.shape[0]
    with torch.no_grad():
        gt, gp = self._gamma(params, torch.full((B,), float(np.float32(t)), device=x.device))
        gt, gp = gt.contiguous(), gp.contiguous()
    cond = ctx["emb"][:, :1].contiguous()
    xin = x.detach().reshape(B, D).contiguous()
    if hutch is None:
        with torch.no_grad():
            net = score_unet(params["score_model"], cfg, xin.view(B, HW, 3), gt, cond, _Drop(None, 0.0))
            drift, _ = ops.ode_drift(net.reshape(B, D), xin, gt, gp, None, 2, drift_out)
        return drift, None
    xin.requires_grad_(True)
    with torch.enable_grad():
        net = score_unet(params["score_model"], cfg, xin.view(B, HW, 3), gt, cond, _Drop(None, 0.0))
    drift, cot = ops.ode_drift(net.detach().reshape(B, D), xin.detach(), gt, gp, hutch, 2, drift_out)
    (gx,) = torch.autograd.grad(net, xin, cot.view_as(net))
    return drift, ops.ode_div(gx.reshape(B, D), gt, gp, hutch, 2, div_out)


PlainVDM.ode_context_from_embedding = lambda self, params, emb: dict(emb=emb, kl=None, coeffs=None, logits=None)
PlainVDM.apply_encoder = _plain_apply_encoder
PlainVDM.ode_context = _plain_ode_context
PlainVDM.reverse_ode = _plain_reverse_ode
PlainVDM.sample = _plain_sample
PlainVDM.generate_x = _plain_generate_x
PlainVDM.sample_coefficients = lambda self, params, embedding: None


# ----------------------------------------------------------------------------- reference-shaped module surface
class _Module:
    """Flax-style handle of one sub-network: Module(config).apply(params, *args) with the reference's call signature, or
    Module(config).bind(params)(*args).  `params` is that sub-network's tree (e.g. state.params['score_model']);
    `rngs={'dropout': Key}` is needed with deterministic=False, like Flax's apply(..., rngs=...)."""

    def __init__(self, config: VDMConfig):
        self.config = config

    def bind(self, params, rngs=None):
        return functools.partial(self.apply, params, rngs=rngs)

    def _drop(self, deterministic, rngs):
        if deterministic:
            return _Drop(None, 0.0)
        key = (rngs or {}).get("dropout")
        if key is None:
            raise ValueError("deterministic=False needs rngs={'dropout': Key}")
        return _Drop(key, self.config.sm_pdrop)


def _pixels(z, ch):
    """[B,32,32,ch] (reference layout) or [B,1024,ch] -> [B,1024,ch] fp32"""
    return z.reshape(z.shape[0], HW, ch).to(torch.float32)


class ScoreUNet(_Module):
    """model_vdm.ScoreUNet (ldm/model_vdm.py:309-388): __call__(z, g_t, conditioning, deterministic=True, time=False)
    -> eps_hat with the shape of z.  z [B,32,32,3]; g_t [B] (or a scalar); conditioning [B,K]."""
    unet_type = 'vdm'

    def apply(self, params, z, g_t, conditioning, deterministic=True, time=False, rngs=None):
        cfg = self.config if self.config.unet_type == self.unet_type else dataclasses.replace(self.config, unet_type=self.unet_type)
        B = z.shape[0]
        g = torch.as_tensor(g_t, dtype=torch.float32, device=z.device)
        if self.unet_type == 'ldm':
            g = g.expand(z.shape).reshape(B, HW, 3).contiguous() if g.dim() < 4 and g.numel() != z.numel() \
                else g.reshape(B, HW, 3)
        else:
            g = g.expand(B).contiguous() if g.dim() == 0 else g.reshape(B)    # model_vdm.py:329-332: g_t * ones(B)
        cond = torch.as_tensor(conditioning, dtype=torch.float32, device=z.device).reshape(B, -1)
        out = score_unet(params, cfg, _pixels(z, 3), g, cond, self._drop(deterministic, rngs), time=time)
        return out.reshape(z.shape)

    __call__ = apply


class UNet(ScoreUNet):
    """ldm_unet.UNet (ldm/ldm_unet.py:64-142): the same topology with per-pixel FiLM conditioning; g_t [B,32,32,3]."""
    unet_type = 'ldm'


class UnetEncoder(_Module):
    """model_mulan_epsilon.UnetEncoder (ldm/model_mulan_epsilon.py:101-154): __call__(z, deterministic=True) ->
    logits [B, latent_size]; z = EncDec.encode(images) [B,32,32,3]."""

    def apply(self, params, z, deterministic=True, rngs=None):
        return unet_encoder(params, self.config, _pixels(z, 3), self._drop(deterministic, rngs))

    __call__ = apply


class NoiseSchedule_polynomial_fixedend(_Module):
    """model_mulan_epsilon.NoiseSchedule_polynomial_fixedend (ldm/model_mulan_epsilon.py:481-613):
    __call__(embedding [B,latent], t [B] or scalar) -> gamma [B,3072]; grad_t(embedding, t) = d gamma / dt (:540-555)."""

    def _eval(self, params, embedding, t):
        emb = torch.as_tensor(embedding, dtype=torch.float32)
        B = emb.shape[0]
        t = torch.as_tensor(t, dtype=torch.float32, device=emb.device)
        t = t.expand(B).contiguous() if t.dim() == 0 else t.reshape(B)
        a, b, c = poly_coefficients(params, emb)
        return ops.poly_gamma(a, b, c, t, self.config.gamma_min, self.config.gamma_max)

    def apply(self, params, embedding, t, rngs=None):
        return self._eval(params, embedding, t)[2]

    def grad_t(self, params, embedding, t):
        return self._eval(params, embedding, t)[3]

    __call__ = apply


class EncDec:
    """model_vdm.EncDec (ldm/model_vdm.py:265-303): no parameters.  encode(x uint8) -> f in (-1, 1);
    decode(z, g_0) -> [..., 256] decoder log-probabilities; logprob(x, z, g_0) -> [B] sum over sub-pixels of the
    log-probability of x's bin (the fused q-sample kernel, which never builds the 256-bin table)."""

    def __init__(self, config: VDMConfig):
        self.config = config

    def __call__(self, x, g_0):
        """EncDec.__call__ (:269-272): decode(encode(x), g_0)"""
        return self.decode(self.encode(x), g_0)

    def encode(self, x):
        return encode_images(x.reshape(x.shape[0], D).to(torch.uint8)).reshape(*x.shape)

    def _g(self, g_0, z):
        """g_0 as the reference accepts it -- a scalar, [B], or the shape of z -- as [B] or [B, D] fp32"""
        B = z.shape[0]
        g = torch.as_tensor(g_0, dtype=torch.float32, device=z.device)
        if g.numel() == 1:
            return g.reshape(1).expand(B).contiguous()
        if g.numel() == B:
            return g.reshape(B).contiguous()
        if g.numel() != z.numel():
            raise ValueError(f"g_0 of shape {tuple(g.shape)} does not broadcast over z {tuple(z.shape)}")
        return g.reshape(B, D).contiguous()

    def decode(self, z, g_0):
        """EncDec.decode (:282-296): log_softmax over the 256 bins of -0.5 ((z - v_j) exp(-g_0 / 2))^2, shape
        z.shape + (256,)"""
        return ops.decode_logprobs(z.reshape(z.shape[0], D), self._g(g_0, z)).reshape(*z.shape, self.config.vocab_size)

    def logprob(self, x, z, g_0):
        """EncDec.logprob (:298-303): sum over (H, W, C) of decode(z, g_0)[x] -> [B].  Evaluated by mulan_qsample_fwd
        (its reconstruction term at z_0 = f + exp(g_0 / 2) eps_0 with eps_0 chosen so that z_0 = z), no [.., 256]
        tensor.  z may carry a gradient (the kernel's backward covers g_0; z is a leaf here, like in the train path
        where z_0 is sampled, not learned)."""
        B = z.shape[0]
        xu = x.reshape(B, D).round().to(torch.uint8).contiguous()
        g = self._g(g_0, z)
        g = g if g.dim() == 2 else g[:, None].expand(B, D).contiguous()
        f = encode_images(xu).reshape(B, D)
        eps_0 = (z.reshape(B, D).to(torch.float32) - f) * torch.exp(-0.5 * g)
        _, _, recon, _, _, _ = ops.qsample(xu, g, g, g, eps_0, torch.zeros_like(eps_0))
        return -recon

    def decode_argmax(self, z, g_0):
        """argmax over the 256 bins of decode(z, g_0) (:282-293); g_0 per sample [B] or per element"""
        B = z.shape[0]
        g = torch.as_tensor(g_0, dtype=torch.float32, device=z.device)
        g = g.expand(B).contiguous() if g.dim() == 0 else g.reshape(B, -1).squeeze(-1) if g.numel() == B else g.reshape(B, D)
        return ops.decode_argmax(z.reshape(B, D).to(torch.float32).contiguous(), g.contiguous()).reshape(z.shape)


def make_vdm(vdm_type: str, config: VDMConfig):
    """Experiment_VDM.get_model_and_params dispatch (ldm/experiment_vdm.py:32-38)."""
    if vdm_type == 'mulan_velocity':
        return MulanVDM(config, "velocity")
    if vdm_type == 'mulan_epsilon':
        return MulanVDM(config, "epsilon")
    if vdm_type == 'vdm':
        return PlainVDM(config)
    raise KeyError(vdm_type)


# ----------------------------------------------------------------------------- tree utilities
def tree_leaves(tree, prefix=()):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from tree_leaves(v, prefix + (k,))
        else:
            yield prefix + (k,), v


def tree_map(fn, tree):
    return {k: tree_map(fn, v) if isinstance(v, dict) else fn(v) for k, v in tree.items()}


def tree_set(tree, path, value):
    for k in path[:-1]:
        tree = tree[k]
    tree[path[-1]] = value


def to_flax_layout(tree):
    """Product tree -> reference (Flax) layout: drops the zero 16th input channel of conv_in."""
    def fix(path, v):
        if path[-2:] == ("conv_in", "kernel") and v.shape[2] == 16:
            return v[:, :, :15, :]
        return v
    out = {}
    for path, v in tree_leaves(tree):
        d = out
        for k in path[:-1]:
            d = d.setdefault(k, {})
        d[path[-1]] = fix(path, v)
    return out


def from_flax_layout(flax_tree, like):
    """Copies a reference-layout tree into the leaves of `like` (in place, under no_grad)."""
    with torch.no_grad():
        for path, dst in tree_leaves(like):
            src = flax_tree
            for k in path:
                src = src[k]
            src = torch.as_tensor(src, dtype=torch.float32)
            if path[-2:] == ("conv_in", "kernel") and src.shape[2] == 15:
                src = torch.cat([src, torch.zeros(3, 3, 1, src.shape[3])], dim=2)
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"shape mismatch at {'/'.join(path)}: {tuple(src.shape)} vs {tuple(dst.shape)}")
            dst.copy_(src.to(dst.device))
    return like
