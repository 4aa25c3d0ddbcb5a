"""Generates tests/golden/*.npz from the CPU oracle (run in the build container: `python tests/golden/make_golden.py`).

The reference ships no golden vectors and cannot be imported here (JAX/Flax absent), so these fixtures are
OUTPUTS OF THE ORACLE ITSELF on seeded inputs: they pin the oracle against accidental edits and give the
GPU parity tests a second, file-based target.  They are data only (inputs + expected outputs).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mulan_np as onp  # noqa: E402
from oracle import torch_ref as tr  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def closed_forms():
    rng = np.random.default_rng(20240101)
    B = 3
    a = rng.standard_normal((B, 3072)) * 0.5
    b = rng.standard_normal((B, 3072)) * 0.5
    c = 1e-3 + np.logaddexp(rng.standard_normal((B, 3072)), 0)
    t = onp.antithetic_t(0.4321, B)
    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    e0 = rng.standard_normal((B, 32, 32, 3))
    e = rng.standard_normal((B, 32, 32, 3))
    net = rng.standard_normal((B, 32, 32, 3))
    shp = (B, 32, 32, 3)
    g0 = onp.poly_gamma(a, b, c, np.zeros(B)).reshape(shp)
    g1 = onp.poly_gamma(a, b, c, np.ones(B)).reshape(shp)
    gt = onp.poly_gamma(a, b, c, t).reshape(shp)
    gp = onp.poly_gamma_grad_t(a, b, c, t).reshape(shp)
    f, zt, recon, klz, v0, v1 = onp.elbo_pre(x, g0, g1, gt, e0, e)
    logits = rng.standard_normal((B, 50)) * 2
    raw = rng.gamma(1.0 / 15, size=(10, B, 50))
    emb, kl, soft = onp.topk_embedding_and_loss(logits, raw, 15)
    np.savez_compressed(
        os.path.join(HERE, "closed_forms.npz"), a=a, b=b, c=c, t=t, x=x, eps_0=e0, eps=e, net=net,
        g_t=gt, g_prime=gp, z_t=zt, loss_recon=recon, loss_klz=klz, var_0=v0, var_1=v1,
        loss_diff_velocity=onp.diffusion_loss_velocity(f, gt, gp, e, zt, net, False),
        loss_diff_vfe=onp.diffusion_loss_velocity(f, gt, gp, e, zt, net, True),
        loss_diff_epsilon=onp.diffusion_loss_epsilon(gp, e, net),
        logits=logits, gamma_raw=raw, embedding=emb, kl_z=kl,
        temb_t=np.array([0.0, 0.25, 0.731, 1.0]),
        temb=onp.timestep_embedding(np.array([0.0, 0.25, 0.731, 1.0]), 128),
        fourier_z=np.linspace(-2, 2, 12).reshape(4, 3), fourier=onp.fourier_features(np.linspace(-2, 2, 12).reshape(4, 3)))


def tiny_model():
    """Full MuLAN forward (E=128, 1+2+2 score blocks, 1+2 encoder blocks) with oracle-seeded parameters:
    only seeds, inputs and the expected scalars are stored (the 38 M-parameter tree is rebuilt from the seed)."""
    B = 2
    rng = np.random.default_rng(77)
    x = rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8)
    raw = rng.gamma(1.0 / 15, size=(10, B, 50))
    e0 = rng.standard_normal((B, 32, 32, 3))
    e = rng.standard_normal((B, 32, 32, 3))
    out = {}
    for name, vt, ut, vfe in (("velocity", "mulan_velocity", "vdm", False), ("epsilon", "mulan_epsilon", "vdm", False),
                              ("vfe", "mulan_velocity", "vdm", True), ("ldm", "mulan_velocity", "ldm", False)):
        cfg = dict(vdm_type=vt, n_embd=128, n_layer=1, forward_n_layer=1, latent_k=15, unet_type=ut,
                   velocity_from_epsilon=vfe)
        P = tr.init_params(cfg, seed=11, dtype=torch.float64)
        r = tr.mulan_forward(P, cfg, torch.tensor(x), 0.1234, torch.tensor(raw), torch.tensor(e0), torch.tensor(e))
        out[f"{name}_bpd"] = float(r["bpd"])
        out[f"{name}_recon"] = r["loss_recon"].detach().numpy()
        out[f"{name}_klz"] = r["loss_klz"].detach().numpy()
        out[f"{name}_diff"] = r["loss_diff"].detach().numpy()
        if name == "velocity":   # numpy restatement must agree with the torch one
            r2 = onp.mulan_forward(tr.to_np_tuples(P), cfg, x, 0.1234, raw, e0, e)
            assert abs(r2["bpd"] - out["velocity_bpd"]) < 1e-10
    np.savez_compressed(os.path.join(HERE, "tiny_model.npz"), x=x, gamma_raw=raw, eps_0=e0, eps=e, t0=0.1234,
                        param_seed=11, **out)


def sampler_and_ode():
    """closed forms of the ancestral sampler and of the probability-flow ODE evaluator (oracle/torch_ref.py)"""
    rng = np.random.default_rng(20240202)
    n = 512
    z, net, eps = (torch.tensor(rng.standard_normal(n)) for _ in range(3))
    g_t = torch.tensor(rng.uniform(-13.3, 5.0, n))
    g_s = g_t - torch.tensor(rng.uniform(1e-3, 1.0, n))
    g_p = torch.tensor(rng.uniform(1.0, 40.0, n))
    out = dict(z=z.numpy(), net=net.numpy(), eps=eps.numpy(), g_t=g_t.numpy(), g_s=g_s.numpy(), g_p=g_p.numpy())
    for kind in ("velocity", "epsilon", "input"):
        out[f"step_{kind}"] = tr.ancestral_step(z, net, g_t, g_s, eps, kind).numpy()
    for kind in ("velocity", "vfe", "epsilon"):
        out[f"drift_{kind}"] = tr.ode_drift(net, z, g_t, g_p, kind).numpy()
    z0 = torch.tensor(rng.uniform(-1.2, 1.2, n)) * 0.01
    g0 = torch.tensor(rng.uniform(-13.3, -9.0, n))
    out.update(z0=z0.numpy(), g0=g0.numpy(), decoded=tr.decode_argmax(z0, g0).numpy())
    logits = torch.tensor(rng.standard_normal((6, 50)))
    out.update(logits=logits.numpy(), hard_topk=tr.logits_to_embeddings(logits).numpy(),
               kl=tr.gumbel_kl_loss(logits).numpy(), prior_logp=tr.prior_logp(z.reshape(2, 16, 16, 1)).numpy(),
               bpd_offsets=np.array([tr.bpd_offset("uniform", 1), tr.bpd_offset("tn", 1), tr.bpd_offset("tn", 20)]))
    # Dormand-Prince on a prescribed grid: y' = -y (1 + t) + sin(3 t)
    grid = [0.0, 0.1, 0.25, 0.5, 0.8, 1.0]
    y1 = tr.dopri5_fixed(lambda t, y: -y * (1 + t) + np.sin(3 * t), np.array([1.0, -0.5, 2.0]), grid)
    out.update(dopri_grid=np.array(grid), dopri_y1=y1)
    np.savez_compressed(os.path.join(HERE, "sampler_ode.npz"), **out)


if __name__ == "__main__":
    closed_forms()
    tiny_model()
    sampler_and_ode()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
