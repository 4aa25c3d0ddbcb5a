"""Writes tests/golden/tiny_state.flax: a Flax-0.7 msgpack state dict assembled BYTE BY BYTE from the msgpack
specification -- no call into mulan_amd.checkpoint.save_flax and no msgpack library on the writing side -- so that
`load_flax` is tested against bytes it did not produce itself.

What flax.serialization writes (flax 0.7.0, the version the reference pins; ldm/experiment.py:210-214 saves
`flax.serialization.to_state_dict(state)` through clu, ldm/notebook_utils.py:31-37 reads it back):
  * dict          -> msgpack map, keys are str (tuples / lists / NamedTuples become {'0': .., '1': ..});
  * ndarray       -> ext type 1 whose payload is packb((shape, dtype.name, tobytes('C')), use_bin_type=True);
  * numpy scalar  -> ext type 3, same payload with shape ();
  * arrays above 2**30 bytes -> {'__msgpack_chunked_array__': True, 'shape': {'0': d0, ..}, 'chunks': {'0': ext1, ..}};
  * optax.MaskedNode() (a masked-out leaf of optax.masked) -> empty map;
  * the optimizer state of ldm/experiment.py:160-170 (two optax.masked AdamW instances split by score_model / the rest):
    {'0': {'inner_state': {'0': {count, mu, nu}, '1': {'inner_state': {}}, '2': {}}}, '1': {... the same shape}}.
msgpack headers used: fixmap 0x80|n, fixarray 0x90|n, fixstr 0xa0|n, str8 0xd9, positive fixint, uint8 0xcc,
uint16 0xcd, true 0xc3, bin8 0xc4, bin16 0xc5, fixext4/8/16 0xd6/0xd7/0xd8, ext8 0xc7, ext16 0xc8.
Run in the build container: `python tests/golden/make_flax_fixture.py`.
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def p_str(s):
    b = s.encode()
    if len(b) < 32:
        return bytes([0xa0 | len(b)]) + b
    assert len(b) < 256
    return b"\xd9" + struct.pack("B", len(b)) + b


def p_uint(n):
    if n < 128:
        return struct.pack("B", n)
    if n < 256:
        return b"\xcc" + struct.pack("B", n)
    assert n < 65536
    return b"\xcd" + struct.pack(">H", n)


def p_bin(b):
    if len(b) < 256:
        return b"\xc4" + struct.pack("B", len(b)) + b
    assert len(b) < 65536
    return b"\xc5" + struct.pack(">H", len(b)) + b


def p_ext(code, payload):
    n = len(payload)
    fix = {1: 0xd4, 2: 0xd5, 4: 0xd6, 8: 0xd7, 16: 0xd8}
    if n in fix:
        return bytes([fix[n], code]) + payload
    if n < 256:
        return b"\xc7" + struct.pack("Bb", n, code) + payload
    assert n < 65536
    return b"\xc8" + struct.pack(">Hb", n, code) + payload


def p_array_payload(a):
    a = np.asarray(a)
    shape = bytes([0x90 | a.ndim]) + b"".join(p_uint(d) for d in a.shape)
    return b"\x93" + shape + p_str(a.dtype.name) + p_bin(a.tobytes("C"))


def p_ndarray(a):
    return p_ext(1, p_array_payload(a))


def p_npscalar(a):
    return p_ext(3, p_array_payload(np.asarray(a)))


def p_map(items):
    assert len(items) < 16
    return bytes([0x80 | len(items)]) + b"".join(p_str(k) + v for k, v in items)


def p_chunked(a, chunk):
    flat = np.asarray(a).reshape(-1)
    chunks = [flat[i:i + chunk] for i in range(0, flat.size, chunk)]
    return p_map([("__msgpack_chunked_array__", b"\xc3"),
                  ("shape", p_map([(str(i), p_uint(d)) for i, d in enumerate(a.shape)])),
                  ("chunks", p_map([(str(i), p_ndarray(c)) for i, c in enumerate(chunks)]))])


def expected_tree():
    """the content of the fixture, as plain numpy (what load_flax must return after its normalisation)"""
    f = np.float32
    params = {
        "gamma": {"dense_1": {"kernel": (np.arange(6, dtype=f).reshape(2, 3) - 2.5) * f(0.25),
                              "bias": np.array([0.5, -1.0, 3.0], dtype=f)}},
        "score_model": {"conv_out": {"kernel": np.arange(3 * 3 * 2 * 1, dtype=f).reshape(3, 3, 2, 1) * f(-0.125),
                                     "bias": np.array([1.5], dtype=f)},
                        "norm_out": {"scale": np.linspace(0.5, 2.0, 24).astype(f)}},
    }
    ema = {k: {kk: {k3: v3 * f(0.5) for k3, v3 in vv.items()} for kk, vv in v.items()} for k, v in params.items()}
    mu = {k: {kk: {k3: v3 * f(1e-3) for k3, v3 in vv.items()} for kk, vv in v.items()} for k, v in params.items()}
    nu = {k: {kk: {k3: v3 * v3 * f(1e-6) for k3, v3 in vv.items()} for kk, vv in v.items()} for k, v in params.items()}
    return {"step": 223, "params": params, "ema_params": ema, "opt_state": {"mu": mu, "nu": nu}}


def build_bytes():
    t = expected_tree()

    def tree(d, leaf=p_ndarray, keep=lambda path: True, names=None, path=()):
        items = []
        for k, v in d.items():
            kk = (names or {}).get(k, k)
            if isinstance(v, dict):
                items.append((kk, tree(v, leaf, keep, names, path + (k,))))
            else:
                items.append((kk, leaf(v) if keep(path + (k,)) else p_map([])))     # masked-out leaf: MaskedNode -> {}
        return p_map(items)

    # the gamma network under its attribute names (model_mulan_epsilon.py:493-512; checkpoint.GAMMA_NET_ALIASES)
    alias = {"dense_1": "l1"}
    # ldm/experiment.py:160-170: chain(masked(adamw, score_model leaves), masked(adamw, the rest)) -- the two optax.masked
    # states split the tree by its TOP-LEVEL module, not by the decay mask (that one sits inside each adamw)
    score = lambda path: path[0] == "score_model"
    not_score = lambda path: path[0] != "score_model"
    # norm_out/scale (96 bytes) goes out chunked, 16 elements per chunk, in ema_params only
    def ema_leaf(v):
        return p_chunked(v, 16) if v.size == 24 else p_ndarray(v)

    def adam(keep):
        return p_map([("count", p_ndarray(np.asarray(223, dtype=np.int32))),
                      ("mu", tree(t["opt_state"]["mu"], keep=keep, names=alias)),
                      ("nu", tree(t["opt_state"]["nu"], keep=keep, names=alias))])
    # optax.adamw(mask=decay_mask_fn) = chain(scale_by_adam, add_decayed_weights(mask), scale(-lr)): ScaleByAdamState,
    # MaskedState(inner_state=AddDecayedWeightsState()) -> {'inner_state': {}}, ScaleState() -> {}
    empty = p_map([])
    masked_decay = p_map([("inner_state", empty)])
    opt = p_map([("0", p_map([("inner_state", p_map([("0", adam(score)), ("1", masked_decay), ("2", empty)]))])),
                 ("1", p_map([("inner_state", p_map([("0", adam(not_score)), ("1", masked_decay), ("2", empty)]))]))])
    return p_map([("step", p_npscalar(np.int32(223))),
                  ("params", p_map([("params", tree(t["params"], names=alias))])),       # one wrapping level, as Flax variables
                  ("ema_params", tree(t["ema_params"], leaf=ema_leaf, names=alias)),
                  ("opt_state", opt)])


if __name__ == "__main__":
    b = build_bytes()
    with open(os.path.join(HERE, "tiny_state.flax"), "wb") as f:
        f.write(b)
    print("wrote tiny_state.flax:", len(b), "bytes")
