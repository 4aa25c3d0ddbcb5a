"""Counter-based RNG plumbing for the host side (stands in for jax.random keys).

A key is a 64-bit integer; `split` / `fold_in` derive independent keys with splitmix64 (the
reference uses jax.random.split / fold_in: ldm/experiment.py:336-337, ldm/experiment_vdm.py:48-52).
Tensors are drawn on the device: normals by the library's Philox kernel, Gamma(alpha) draws by
torch's device sampler (host plumbing, off the hot path: 10*B*50 values per step).
"""
import torch

_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class Key:
    """`dev` (optional): a 1-element int64 device tensor holding this key's value.  Kernels that take their seed from a
    key then read the slot when they run (stream-ordered parameter) instead of a value frozen into the launch -- what a
    captured HIP graph of the train step needs (mulan_amd.experiment.GraphedStep rewrites the slots every step)."""
    __slots__ = ("v", "dev")

    def __init__(self, v, dev=None):
        self.v = int(v) & _M64
        self.dev = dev

    def fold_in(self, data):
        return Key(_splitmix64(self.v ^ _splitmix64(int(data) & _M64)))

    def split(self, n=2):
        return tuple(Key(_splitmix64((self.v + 0x632BE59BD9B4E019 * (i + 1)) & _M64)) for i in range(n))

    def uniform(self):
        """one U[0,1) scalar on the host (t0 of the antithetic sampler, ldm/model_mulan_velocity.py:197)"""
        return (_splitmix64(self.v) >> 11) * (1.0 / (1 << 53))

    def normal(self, shape, device):
        from . import ops
        return ops.randn(shape, self.v, 0, device)

    def gamma(self, alpha, shape, device):
        g = torch.Generator(device=device)
        g.manual_seed(self.v & ((1 << 63) - 1))
        return torch._standard_gamma(torch.full(shape, float(alpha), device=device, dtype=torch.float32), generator=g)

    def __repr__(self):
        return f"Key({self.v:#018x})"


def PRNGKey(seed):
    return Key(_splitmix64(int(seed)))
