"""The 6-pass bf16 split convolution (fp32-equivalent products on the bf16 matrix cores) against the same oracle and
tolerances as the exact-fp32 MFMA kernel: bit-exact on integer data, and on random data an error vs float64 of the same
order as the fp32 kernel's own (2^-23-ish relative to sum |a||b|)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mulan_np as onp
from oracle import torch_ref as tr


@pytest.fixture()
def ops(monkeypatch):
    from mulan_amd import ops as _ops
    _ops.lib.load()
    monkeypatch.setattr(_ops, "CONV_MODE", "bf16x6")
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).float().cuda()


@pytest.mark.parametrize("B,C,N", [(2, 128, 128), (1, 256, 128), (1, 128, 256), (1, 16, 128), (3, 48, 128)])
def test_bf16x6_conv_exact_on_integers(ops, B, C, N):
    rng = np.random.default_rng(B + C + N)
    x = rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64)
    w = rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64)
    bias, cb = rng.integers(-3, 4, N).astype(np.float64), rng.integers(-3, 4, (B, N)).astype(np.float64)
    res = rng.integers(-3, 4, (B, 32, 32, N)).astype(np.float64)
    ref = onp.conv3x3(x, w, bias) + cb[:, None, None, :] + res
    y = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w), dev(bias), dev(cb), dev(res).view(B, 1024, N))
    assert np.array_equal(y.cpu().double().numpy().reshape(ref.shape), ref)
    cb2 = rng.integers(-3, 4, (B, 32, 32, N)).astype(np.float64)
    y2 = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w), None, dev(cb2).view(B, 1024, N), None)
    assert np.array_equal(y2.cpu().double().numpy().reshape(ref.shape), onp.conv3x3(x, w) + cb2)


@pytest.mark.parametrize("B,C,N", [(2, 128, 128), (1, 256, 128), (1, 128, 256)])
def test_bf16x6_dgrad_exact_on_integers(ops, B, C, N):
    rng = np.random.default_rng(C * 3 + N)
    x = torch.tensor(rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64), requires_grad=True)
    w = torch.tensor(rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64))
    dy = torch.tensor(rng.integers(-2, 3, (B, 32, 32, N)).astype(np.float64))
    tr.conv3x3(x, {"kernel": w}).backward(dy)
    dx = ops.conv3x3_dgrad_raw(dev(dy).view(B, 1024, N), dev(w))
    assert np.array_equal(dx.cpu().double().numpy().reshape(B, 32, 32, C), x.grad.numpy())


@pytest.mark.parametrize("B,C,N", [(2, 128, 128), (1, 256, 128), (3, 64, 64), (1, 16, 128), (2, 48, 96), (1, 128, 4)])
def test_bf16x6_wgrad_exact_on_integers(ops, B, C, N):
    rng = np.random.default_rng(C * 5 + N + B)
    x = torch.tensor(rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64))
    w = torch.tensor(rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64), requires_grad=True)
    dy = torch.tensor(rng.integers(-2, 3, (B, 32, 32, N)).astype(np.float64))
    tr.conv3x3(x, {"kernel": w}).backward(dy)
    dw = ops.conv3x3_wgrad_raw(dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N))
    assert np.array_equal(dw.cpu().double().numpy(), w.grad.numpy())


def test_bf16x6_wgrad_accuracy(ops, monkeypatch):
    rng = np.random.default_rng(1)
    B, C, N = 4, 128, 128
    x = rng.standard_normal((B, 32, 32, C)) * np.exp(rng.standard_normal((B, 32, 32, C)))
    dy = rng.standard_normal((B, 32, 32, N)) * np.exp(rng.standard_normal((B, 32, 32, N)))
    xt = torch.tensor(x.astype(np.float32).astype(np.float64))
    wt = torch.zeros(3, 3, C, N, dtype=torch.float64, requires_grad=True)
    tr.conv3x3(xt, {"kernel": wt}).backward(torch.tensor(dy.astype(np.float32).astype(np.float64)))
    ref = wt.grad.numpy()
    d6 = ops.conv3x3_wgrad_raw(dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N)).cpu().double().numpy()
    monkeypatch.setattr(ops, "CONV_MODE", "f32")
    d32 = ops.conv3x3_wgrad_raw(dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N)).cpu().double().numpy()
    scale = np.abs(ref).max()
    e6, e32 = np.abs(d6 - ref).max() / scale, np.abs(d32 - ref).max() / scale
    assert e6 < 1e-5 and e32 < 1e-5 and e6 < 2 * e32 + 1e-6, (e6, e32)


def test_bf16x6_accuracy_matches_fp32_kernel(ops, monkeypatch):
    """random data with a wide dynamic range: max error relative to sum_k |a_k b_k| for both kernels"""
    rng = np.random.default_rng(0)
    B, C, N = 2, 128, 128
    x = rng.standard_normal((B, 32, 32, C)) * np.exp(rng.standard_normal((B, 32, 32, C)))
    w = rng.standard_normal((3, 3, C, N)) * np.exp(rng.standard_normal((3, 3, C, N))) / math.sqrt(9 * C)
    ref = onp.conv3x3(x, w)
    mag = onp.conv3x3(np.abs(x), np.abs(w))             # sum |a||b| per output
    y6 = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w)).cpu().double().numpy().reshape(ref.shape)
    monkeypatch.setattr(ops, "CONV_MODE", "f32")
    y32 = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w)).cpu().double().numpy().reshape(ref.shape)
    # inputs were rounded to fp32 on the way in: compare against the float64 conv of the rounded inputs
    ref = onp.conv3x3(x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64))
    e6 = float((np.abs(y6 - ref) / mag).max())
    e32 = float((np.abs(y32 - ref) / mag).max())
    # K = 1152 products accumulated in fp32: both kernels sit at ~1e-6 of sum|a||b|; the split must not be worse
    assert e32 < 3e-6 and e6 < 3e-6 and e6 < 1.5 * e32 + 2e-7, (e6, e32)
    assert float(np.abs(y6 - ref).max() / np.abs(ref).max()) < 1e-5   # the bar of test_conv3x3_float_tolerance


def test_bf16x6_whole_model_parity(ops):
    """MuLAN train-mode step through the bf16x6 convolutions: same parity bars as the fp32 path"""
    from tests.test_gpu_model import run_case
    run_case("mulan_velocity", "vdm", False, train=True)


def test_f32_mfma_whole_model_parity(ops, monkeypatch):
    """the same step with MULAN_CONV_MODE=f32 (exact-fp32 MFMA kernels)"""
    from tests.test_gpu_model import run_case
    monkeypatch.setattr(ops, "CONV_MODE", "f32")
    run_case("mulan_epsilon", "vdm", False, train=True)
