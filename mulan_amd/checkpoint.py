"""Checkpoint I/O: {step, params, ema_params, opt_state} (ldm/train_state.py:62-68).

Native format: `<dir>/ckpt-<N>.pt` (torch.save of nested dicts of CPU tensors).  The reference's CLU
checkpoints (`ckpt-<N>.flax` = Flax msgpack of the state dict, ldm/experiment.py:210-214,292-294;
ldm/notebook_utils.py:31-37) are read without Flax: ndarray = msgpack ExtType(1, packb((shape, dtype,
bytes))), large arrays chunked under '__msgpack_chunked_array__'.  `save_flax` writes the same layout.
"""
import json
import os
import re
import time

import numpy as np
import torch

_CKPT_RE = re.compile(r"ckpt-(\d+)")


def checkpoint_numbers(directory):
    """Numbers N of every file whose name contains 'ckpt' (ldm/eval_bpd.py:50-55)."""
    nums = set()
    if os.path.isdir(directory):
        for name in os.listdir(directory):
            if 'ckpt' in name:
                m = _CKPT_RE.search(name)
                if m:
                    nums.add(int(m.group(1)))
    return sorted(nums)


def _path_for(directory, n):
    for ext in (".pt", ".flax", ""):
        p = os.path.join(directory, f"ckpt-{n}{ext}")
        if os.path.isfile(p):
            return p
    return None


def latest_checkpoint(directory):
    nums = checkpoint_numbers(directory)
    return _path_for(directory, nums[-1]) if nums else None


def save(directory, state_dict, max_to_keep=100):
    os.makedirs(directory, exist_ok=True)
    nums = checkpoint_numbers(directory)
    n = (nums[-1] + 1) if nums else 1
    tmp = os.path.join(directory, f".tmp-{n}.pt")
    torch.save(state_dict, tmp)
    os.replace(tmp, os.path.join(directory, f"ckpt-{n}.pt"))
    for old in nums[:max(0, len(nums) + 1 - max_to_keep)]:
        p = _path_for(directory, old)
        if p:
            os.remove(p)
    return n


def restore_dict(path):
    """`path` is a checkpoint file, a 'ckpt-N' stem, or a directory (latest checkpoint)."""
    if os.path.isdir(path):
        p = latest_checkpoint(path)
        if p is None:
            raise FileNotFoundError(f"no ckpt-* in {path}")
        path = p
    elif not os.path.isfile(path):
        d, stem = os.path.dirname(path), os.path.basename(path)
        m = _CKPT_RE.fullmatch(stem)
        cand = _path_for(d, int(m.group(1))) if m else None
        if cand is None:
            raise FileNotFoundError(path)
        path = cand
    if path.endswith(".pt"):
        return torch.load(path, map_location="cpu", weights_only=False)
    return load_flax(path)


# ------------------------------------------------------------------ Flax msgpack (no flax needed)
def _np_from_ext(data):
    import msgpack
    shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
    dtype_name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()


def _ext_hook(code, data):
    import msgpack
    if code == 1:
        return _np_from_ext(data)
    if code == 3:   # numpy scalar
        return _np_from_ext(data)[()]
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if '__msgpack_chunked_array__' in tree:
            sh = tree['shape']       # flax stores the shape as _tuple_to_dict: {'0': d0, '1': d1, ...}
            shape = tuple(int(sh[str(i)]) for i in range(len(sh))) if isinstance(sh, dict) else tuple(int(d) for d in sh)
            chunks = tree['chunks']
            flat = np.concatenate([np.asarray(chunks[str(i)]).reshape(-1) for i in range(len(chunks))])
            return flat.reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def load_flax(path):
    import msgpack
    with open(path, "rb") as f:
        tree = msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)
    tree = _unchunk(tree)
    return _normalise_flax_state(tree)


# The polynomial schedule network declares its layers in setup() as attributes l1, l2, l3_a, l3_b, l3_c AND passes
# name='dense_1' ... 'dense_out_c' (ldm/model_mulan_epsilon.py:493-512): depending on the Flax version the parameter
# collection is keyed by the explicit names or by the attribute names.  Both are accepted; the build's canonical names
# are the explicit ones.
GAMMA_NET_ALIASES = {"l1": "dense_1", "l2": "dense_2", "l3_a": "dense_out_a", "l3_b": "dense_out_b", "l3_c": "dense_out_c"}


def canonical_param_names(tree):
    """A parameter-shaped tree (params, ema_params, Adam mu / nu) with the gamma network's attribute-style layer names
    mapped onto the canonical ones, and a single wrapping {'params': ...} level removed."""
    if not isinstance(tree, dict):
        return tree
    if set(tree.keys()) == {"params"} and isinstance(tree["params"], dict):
        tree = tree["params"]
    out = dict(tree)
    g = out.get("gamma")
    if isinstance(g, dict) and any(k in g for k in GAMMA_NET_ALIASES):
        clash = [k for k in g if k in GAMMA_NET_ALIASES and GAMMA_NET_ALIASES[k] in g]
        if clash:
            raise ValueError(f"gamma network holds both spellings of {clash}")
        out["gamma"] = {GAMMA_NET_ALIASES.get(k, k): v for k, v in g.items()}
    return out


def _normalise_flax_state(sd):
    """Maps the reference's opt_state (optax.chain of two optax.masked AdamW states, ldm/experiment.py:151-173: number 0
    over the score_model leaves, number 1 over the rest; as a Flax state dict
    {'0': {'inner_state': {'0': {count, mu, nu}, '1': {'inner_state': {}}, '2': {}}}, '1': {...}} with masked-out leaves
    serialised as empty nodes) onto {'mu','nu'} -- the two moment trees are merged leaf by leaf, whichever instance
    holds a leaf --, resolves the gamma-network name aliases, and leaves step untouched."""
    out = {k: canonical_param_names(sd[k]) if k != "step" else sd[k] for k in ("step", "params", "ema_params") if k in sd}
    if "step" in out:
        out["step"] = int(np.asarray(out["step"]))
    opt = sd.get("opt_state")
    mus, nus = {}, {}

    def walk(node):
        if isinstance(node, dict):
            if "mu" in node and "nu" in node:
                _merge(mus, node["mu"])
                _merge(nus, node["nu"])
            else:
                for v in node.values():
                    walk(v)
    if opt is not None:
        walk(opt)
        if mus:
            out["opt_state"] = {"mu": canonical_param_names(mus), "nu": canonical_param_names(nus)}
    return out


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict):
            _merge(dst.setdefault(k, {}), v)
        elif v is not None and np.ndim(v) > 0:   # masked-out leaves are serialised as empty/None
            dst[k] = v


def save_flax(path, state_dict):
    """Writes {step, params, ema_params} in the Flax msgpack layout (reference-loadable)."""
    import msgpack

    def enc(o):
        if torch.is_tensor(o):
            o = o.detach().cpu().numpy()
        if isinstance(o, np.ndarray):
            return msgpack.ExtType(1, msgpack.packb((o.shape, o.dtype.name, o.tobytes()), use_bin_type=True))
        if isinstance(o, np.generic):
            return enc(np.asarray(o))
        raise TypeError(type(o))
    with open(path, "wb") as f:
        f.write(msgpack.packb(state_dict, default=enc, use_bin_type=True, strict_types=True))


# ------------------------------------------------------------------ image grids
def generate_image_grids(images):
    """utils.generate_image_grids (ldm/utils.py:101-122): floor(sqrt(B))^2 images tiled into one picture, each row
    laid out right to left like the reference's hstack(...[::-1]); images uint8 [B, H, W, C] (tensor or array)"""
    a = images.detach().cpu().numpy() if torch.is_tensor(images) else np.asarray(images)
    n = int(np.floor(np.sqrt(a.shape[0])))
    rows = [np.hstack([a[r * n + c] for c in range(n)][::-1]) for r in range(n)]
    return np.vstack(rows)


# ------------------------------------------------------------------ scalar logging
class ScalarWriter:
    """CustomLoggingWriter look-alike (ldm/utils.py:168-202): one 'step,key=value,...' line per write."""

    def __init__(self, workdir):
        self.f = None
        if workdir is not None:
            os.makedirs(workdir, exist_ok=True)
            self.f = open(os.path.join(workdir, "metrics.csv"), "a")

    def write_hparams(self, hparams):
        if self.f:
            self.f.write("# hparams " + json.dumps(hparams, default=str) + "\n")
            self.f.flush()

    def write_scalars(self, step, scalars):
        line = f"step={step}," + ",".join(f"{k}={float(v):.6g}" for k, v in sorted(scalars.items()))
        print(time.strftime("%H:%M:%S"), line, flush=True)
        if self.f:
            self.f.write(line + "\n")
            self.f.flush()

    def write_images(self, step, images):
        """one binary PPM per entry (the reference hands [1, H, W, 3] uint8 grids to the clu writers)"""
        if not self.f:
            return
        for name, img in images.items():
            a = np.asarray(img, dtype=np.uint8)
            a = a.reshape(a.shape[-3:])                                   # [H, W, 3]
            with open(os.path.join(os.path.dirname(self.f.name), f"{name}_{int(step):08d}.ppm"), "wb") as g:
                g.write(b"P6\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
                g.write(np.ascontiguousarray(a).tobytes())

    def close(self):
        if self.f:
            self.f.close()
            self.f = None
