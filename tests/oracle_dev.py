"""Where the float64 oracle's torch code runs.

oracle/torch_ref.py is plain torch in float64: IEEE double arithmetic whichever device executes it (checked on every run by
tests/test_gpu_kernels.py::test_oracle_is_the_same_on_host_and_device: forward terms and gradients of one model agree to
1e-9 between the host and the GPU).  The -m gpu suite is bound by the oracle's host time (a full-depth float64 backward
pass takes 40 s on the host cores and under a second on the MI355X's fp64 units), so the heavy comparisons run the ORACLE'S
OWN CODE, unchanged, with its tensors on the device.  This is still the checker -- torch's generic float64 kernels
(rocBLAS dgemm, the native convolution), nothing of mulan_amd -- and MULAN_ORACLE_DEVICE=cpu puts it back on the host.

Independence from the device's libraries does not rest on that one agreement test alone: two whole-model comparisons of the
suite are PINNED TO THE HOST (`pin_oracle_to_host`) -- tests/test_gpu_model.py::test_full_depth_forward_bpd_parity (the
shipped 32 + 2 + 33 + 4-block depth, forward, the +-0.005 bits/dim bar) and ::test_deeper_stack_train_gradients (19 + 4
ResnetBlocks in training mode: losses and every parameter gradient through float64 autograd) -- so every run of the suite
holds the HIP path, forward and backward, against float64 arithmetic that never touched the GPU (35 + 10 s of the suite's
budget; the full-depth BACKWARD pass in float64 takes 40-100 s on the host cores and stays on the device).
"""
import os

import torch


def pin_oracle_to_host(monkeypatch):
    """this test's oracle runs on the host cores whatever MULAN_ORACLE_DEVICE says (MULAN_ORACLE_PIN_HOST=0: dev only)"""
    if os.environ.get("MULAN_ORACLE_PIN_HOST", "1") != "0":
        monkeypatch.setenv("MULAN_ORACLE_DEVICE", "cpu")


def oracle_device():
    return "cuda" if (torch.cuda.is_available() and os.environ.get("MULAN_ORACLE_DEVICE", "cuda") == "cuda") else "cpu"


def _map(tree, fn):
    if torch.is_tensor(tree):
        return fn(tree)
    if isinstance(tree, dict):
        return {k: _map(v, fn) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_map(v, fn) for v in tree)
    return tree


def _pairs(a, b):
    if torch.is_tensor(a):
        yield a, b
    elif isinstance(a, dict):
        for k in a:
            yield from _pairs(a[k], b[k])
    elif isinstance(a, (list, tuple)):
        for x, y in zip(a, b):
            yield from _pairs(x, y)


def run_oracle(fn, params, *args, backward=None, **kwargs):
    """out = fn(params, *args, **kwargs) with every tensor argument moved to oracle_device() and every tensor fn creates
    placed there; `backward`: key of a scalar in `out` to backpropagate -- the gradients land in the `.grad` of the HOST
    leaves of `params`, exactly as if `out[backward].backward()` had run on the host.  Returns `out` detached, on the host."""
    dev = oracle_device()
    if dev == "cpu":
        out = fn(params, *args, **kwargs)
        if backward is not None:
            out[backward].backward()
        return _map(out, lambda t: t.detach())
    move = lambda t: t.detach().to(dev).requires_grad_(t.requires_grad)
    gp, ga, gk = _map(params, move), _map(args, move), _map(kwargs, move)
    with torch.device(dev):
        out = fn(gp, *ga, **gk)
        if backward is not None:
            out[backward].backward()
    if backward is not None:
        for host, devt in _pairs(params, gp):
            if host.requires_grad and devt.grad is not None:      # accumulate, as backward() on the host leaves would
                g = devt.grad.cpu()
                host.grad = g if host.grad is None else host.grad + g
    return _map(out, lambda t: t.detach().cpu())


def params_on_device(tree):
    """a copy of an oracle parameter tree on oracle_device() (made once, reused by many on_device calls)"""
    dev = oracle_device()
    return tree if dev == "cpu" else _map(tree, lambda t: t.detach().to(dev))


def on_device(fn):
    """fn(*tensors) executed on oracle_device(): tensor arguments are moved in, results moved back to the host -- with
    `.to()` / `.cpu()`, which autograd differentiates through, so a caller that differentiates fn with respect to a HOST
    argument (the Hutchinson divergence of the oracle's ODE drift, oracle/torch_ref.value_div) works unchanged"""
    def wrapped(*args, **kwargs):
        dev = oracle_device()
        if dev == "cpu":
            return fn(*args, **kwargs)
        mv = lambda t: t.to(dev)
        with torch.device(dev):
            out = fn(*_map(args, mv), **_map(kwargs, mv))
        return _map(out, lambda t: t.cpu())
    return wrapped
