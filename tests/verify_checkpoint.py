#!/usr/bin/env python3
"""One-command pin of the oracle (and of the HIP path) against a RELEASED MuLAN checkpoint.

The reference publishes BPD only for its released checkpoints (/root/reference README.md:18,42-59: CIFAR-10 ckpt 223
-> 2.55, ImageNet-32 ckpt 220 -> 3.67 by the exact-likelihood evaluator; the variational bound is evaluated the same
way with --bpd_eval_method=dense).  Neither the checkpoints nor JAX are available where this repository is built, so
the float64 oracle is "parity unpinned".  Whoever has a checkpoint closes that gap with:

    python tools/verify_checkpoint.py --ckpt <dir or ckpt-223.flax> --config ldm/configs/cifar10-conditioned.py \\
        --data cifar10 [--data-dir <dir with cifar-10-batches-py>] [--n-images 8] [--n-timesteps 128]

which (1) loads the Flax msgpack through mulan_amd.checkpoint.load_flax, (2) prints the parameter-tree diff against
the model's own init tree (names and shapes; the gamma network's dense_* / l* aliasing is resolved automatically,
ldm/model_mulan_epsilon.py:493-512), (3) evaluates the dense variational bound of N test images with the float64
oracle (CPU) and -- when a HIP device is present -- with the HIP path on the SAME explicit noise, and prints both next
to the README target.  Exit code 0: trees match and |BPD_hip - BPD_oracle| <= 0.005 (the north-star bar).

This is test infrastructure (it imports oracle/): it lives under tests/; tools/verify_checkpoint.py only launches it.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

README_TARGETS = {"cifar10": ("CIFAR-10, ckpt 223", 2.55), "imagenet32": ("ImageNet-32, ckpt 220", 3.67)}


def tree_shapes(tree, prefix=()):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(tree_shapes(v, prefix + (str(k),)))
        else:
            out[prefix + (str(k),)] = tuple(np.shape(v))
    return out


def diff_trees(expected, got):
    """(missing, unexpected, mismatched) between two {path: shape} maps"""
    missing = sorted(p for p in expected if p not in got)
    unexpected = sorted(p for p in got if p not in expected)
    mismatched = sorted((p, expected[p], got[p]) for p in expected if p in got and expected[p] != got[p])
    return missing, unexpected, mismatched


def expected_tree(config):
    """names and shapes of the model's parameters in the reference (Flax) layout"""
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    vdm = M.make_vdm(config.vdm_type, M.VDMConfig(**config.model.to_dict()))
    return vdm, M.to_flax_layout(vdm.init(PRNGKey(0)))


def oracle_cfg(config):
    m = config.model
    return dict(vdm_type=config.vdm_type, n_embd=m.sm_n_embd, n_layer=m.sm_n_layer, forward_n_layer=m.forward_n_layer,
                latent_k=m.get("latent_k", 15), unet_type=m.unet_type, velocity_from_epsilon=bool(m.get("velocity_from_epsilon", False)),
                with_attention=bool(m.with_attention))


def write_synthetic_flax_checkpoint(path, config, seed=4, step=9):
    """A Flax msgpack written the way the reference might write it, from seeded oracle parameters: the gamma network
    under its attribute names l1 .. l3_c (ldm/model_mulan_epsilon.py:493-512), optax.chain(masked(adamw),
    masked(adamw)) optimizer state in CLU / Flax state-dict form with masked-out leaves as empty nodes
    (ldm/experiment.py:151-173), one array in Flax's chunked form; ema_params = 0.5 x params.  Returns (the state dict
    as written, the float64 oracle tree it was made from, the fp32 trees params / ema / mu / nu under the aliased names)."""
    import copy
    from mulan_amd import checkpoint as ck
    from oracle import torch_ref as tr
    ref = tr.init_params(oracle_cfg(config), seed=seed, dtype=torch.float64)
    as_np = lambda tree, scale=1.0: tr.tree_map(lambda t: (t.detach().numpy() * scale).astype(np.float32), tree)
    alias = {v: k for k, v in ck.GAMMA_NET_ALIASES.items()}

    def aliased(tree):
        out = dict(tree)
        out["gamma"] = {alias[k]: v for k, v in tree["gamma"].items()}
        return out

    def masked(tree, keep_score):                      # optax.masked: the other sub-trees become empty MaskedNodes
        empty = lambda t: {k: empty(v) for k, v in t.items()} if isinstance(t, dict) else {}
        return {k: (v if (k == "score_model") == keep_score else empty(v)) for k, v in tree.items()}

    params, ema = aliased(as_np(ref)), aliased(as_np(ref, 0.5))
    mu, nu = aliased(as_np(ref, 0.1)), aliased(as_np(ref, 0.01))
    adam = lambda keep: {"inner_state": {"0": {"count": np.int32(step), "mu": masked(mu, keep), "nu": masked(nu, keep)},
                                         "1": {}, "2": {}}}
    sd = {"step": np.int32(step), "params": params, "ema_params": copy.deepcopy(ema),
          "opt_state": {"0": adam(True), "1": adam(False)}}
    k = ema["score_model"]["dense0"]["kernel"]         # one leaf in Flax's chunked-array form (arrays above 2^30 bytes)
    sd["ema_params"]["score_model"]["dense0"]["kernel"] = {
        "__msgpack_chunked_array__": True, "shape": {"0": k.shape[0], "1": k.shape[1]},
        "chunks": {"0": k.reshape(-1)[:100].copy(), "1": k.reshape(-1)[100:].copy()}}
    ck.save_flax(path, sd)
    return sd, ref, dict(params=params, ema=ema, mu=mu, nu=nu)


def verify(ckpt, config, images, n_timesteps=16, which="ema_params", use_gpu=None, seed=0, log=print):
    """Returns a dict with the tree diff and, per image, the oracle's and (if a device is present) the HIP path's dense
    variational bound in bits/dim."""
    from mulan_amd import checkpoint as ck
    from oracle import torch_ref as tr
    sd = ck.restore_dict(ckpt) if isinstance(ckpt, str) else ckpt
    if which not in sd:
        raise KeyError(f"{which} not in checkpoint (has {sorted(sd)})")
    tree = ck.canonical_param_names(sd[which])
    vdm, tmpl = expected_tree(config)
    missing, unexpected, mismatched = diff_trees(tree_shapes(tmpl), tree_shapes(tree))
    log(f"checkpoint step {sd.get('step', '?')}; {which}: {len(tree_shapes(tree))} leaves, "
        f"{sum(int(np.prod(s)) for s in tree_shapes(tree).values()) / 1e6:.2f} M parameters")
    for p in missing:
        log("  MISSING     " + "/".join(p))
    for p in unexpected:
        log("  UNEXPECTED  " + "/".join(p))
    for p, e, g in mismatched:
        log(f"  SHAPE       {'/'.join(p)}: model {e} vs checkpoint {g}")
    tree_ok = not (missing or unexpected or mismatched)
    log("parameter tree: " + ("matches the model (names and shapes)" if tree_ok else "DOES NOT MATCH"))
    result = dict(tree_ok=tree_ok, missing=missing, unexpected=unexpected, mismatched=mismatched, oracle=[], hip=[])
    if not tree_ok or images is None or len(images) == 0:
        return result

    ocfg = oracle_cfg(config)
    ref_params = tr.tree_map(lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64), tree)
    T = n_timesteps
    rng = np.random.default_rng(seed)                 # one noise set for every image, like the evaluator's fixed key
    t0 = float(rng.random())
    raw = rng.gamma(1.0 / ocfg["latent_k"], size=(10, T, 50))
    e0, e = rng.standard_normal((T, 3072)), rng.standard_normal((T, 3072))
    if use_gpu is None:
        use_gpu = torch.cuda.is_available()
    params_dev = None
    if use_gpu:
        from mulan_amd import model as M
        from mulan_amd.rng import PRNGKey
        params_dev = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(0)))
        M.from_flax_layout(M.tree_map(lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32), tree), params_dev)
        noise = dict(t0=t0, gamma_raw=torch.tensor(raw, dtype=torch.float32).cuda(),
                     eps_0=torch.tensor(e0, dtype=torch.float32).cuda(), eps=torch.tensor(e, dtype=torch.float32).cuda())
    r = 1.0 / (3072 * np.log(2.0))
    for i, img in enumerate(images):
        x = np.repeat(np.asarray(img, dtype=np.uint8)[None], T, axis=0)
        with torch.no_grad():
            o = tr.mulan_forward(ref_params, ocfg, torch.tensor(x), t0, torch.tensor(raw), torch.tensor(e0).view(T, 32, 32, 3),
                                 torch.tensor(e).view(T, 32, 32, 3))
        result["oracle"].append(float(o["bpd"]))
        line = f"image {i}: oracle (float64) {float(o['bpd']):.5f} bits/dim"
        if use_gpu:
            with torch.no_grad():
                out = vdm.apply(params_dev, torch.tensor(x).cuda(), None, None, step=0, rngs=None, deterministic=True,
                                noise=noise)
            h = float((out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()) * r)
            result["hip"].append(h)
            line += f" | HIP {h:.5f} | diff {h - float(o['bpd']):+.2e}"
        log(line)
    return result


def main(argv=None):
    from mulan_amd.config import load_config_file
    from mulan_amd import data
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True, help="checkpoint file (ckpt-N.flax / .pt), ckpt-N stem or directory")
    ap.add_argument("--config", required=True, help="config file exposing get_config() (ldm/configs/*.py)")
    ap.add_argument("--set", action="append", default=[], metavar="a.b=v", help="config override, e.g. model.velocity_from_epsilon=True")
    ap.add_argument("--data", default=None, help="cifar10 | imagenet32 | npz:<file> (default: the config's data.dataset)")
    ap.add_argument("--data-dir", default=None, help="sets MULAN_DATA_DIR")
    ap.add_argument("--n-images", type=int, default=4)
    ap.add_argument("--n-timesteps", type=int, default=16, help="copies per image (the reference's dense evaluator uses 128..1000)")
    ap.add_argument("--which", default="ema_params", choices=["ema_params", "params"])
    ap.add_argument("--no-gpu", action="store_true", help="oracle only")
    a = ap.parse_args(argv)
    if a.data_dir:
        os.environ["MULAN_DATA_DIR"] = a.data_dir
    config = load_config_file(a.config)
    for kv in a.set:
        k, v = kv.split("=", 1)
        config.set_path(k, v)
    name = a.data or config.data.dataset
    images = None
    if a.n_images > 0:
        x, _ = data.load_arrays(name, train=False)
        if x is None:
            x = np.random.default_rng(0).integers(0, 256, (a.n_images, 32, 32, 3), dtype=np.uint8)
        images = x[:a.n_images]
    res = verify(a.ckpt, config, images, n_timesteps=a.n_timesteps, which=a.which, use_gpu=False if a.no_gpu else None)
    ok = res["tree_ok"]
    if res["oracle"]:
        print(f"mean over {len(res['oracle'])} images, T = {a.n_timesteps}: oracle {np.mean(res['oracle']):.4f} bits/dim"
              + (f", HIP {np.mean(res['hip']):.4f} bits/dim" if res["hip"] else ""))
        key = "imagenet32" if "imagenet" in name else ("cifar10" if "cifar" in name else None)
        if key:
            label, target = README_TARGETS[key]
            print(f"README target ({label}, whole test set, exact likelihood): {target:.2f} bits/dim "
                  f"(the variational bound of a few images is an upper bound with image-to-image spread)")
        if res["hip"]:
            worst = max(abs(h - o) for h, o in zip(res["hip"], res["oracle"]))
            print(f"max |HIP - oracle| = {worst:.2e} bits/dim (bar 0.005)")
            ok = ok and worst <= 0.005
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
