"""Parity of every HIP kernel (called through the C ABI via mulan_amd.ops / mulan_amd.lib) against the CPU
oracle on the same seeded inputs.  Integer-valued inputs make the MFMA contractions bit-exact checks of
the fragment layouts; everything else is compared to float64 with the tolerance stated in the test."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mulan_np as onp
from oracle import torch_ref as tr


@pytest.fixture(scope="module")
def ops():
    """This module pins the exact-fp32 MFMA convolution kernels; tests/test_gpu_bf16x6.py covers the default
    6-pass bf16 split kernels against the same oracle."""
    from mulan_amd import ops as _ops
    _ops.lib.load()
    saved = _ops.CONV_MODE
    _ops.CONV_MODE = "f32"
    yield _ops
    _ops.CONV_MODE = saved


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def ints(rng, shape, lo=-3, hi=4):
    return rng.integers(lo, hi, size=shape).astype(np.float64)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ------------------------------------------------------------------------------ plumbing
def test_cpp_driver_calls_the_c_abi_without_torch():
    """tests/abi_driver.cpp -- plain C++ over include/mulan_hip.h + hipMalloc / hipMemcpy / its own stream: the exact and
    the f16x3 convolution, input and weight gradients bit for bit against host loops on integer data, the maxima
    by-product, the error convention, the signal word (the C++ unit-test driver of the boundary)"""
    import os
    import subprocess
    from mulan_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "_bin", "abi_driver")
    if not os.path.exists(exe):
        exe = build.build_abi_driver()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ", 0 failed" in r.stdout, r.stdout


def test_library_loads_and_runs(ops):
    assert "mulan_hip" in ops.lib.version()
    x = torch.arange(1000, dtype=torch.float32).cuda()
    y = torch.ones(1000).cuda()
    ops.call("mulan_axpby", ops.ptr(x), ops.ptr(y), 1000, 2.0, 3.0, ops.stream())
    assert torch.equal(y.cpu(), 2.0 * torch.arange(1000, dtype=torch.float32) + 3.0)


# ------------------------------------------------------------------------------ conv3x3
CONV_SHAPES = [(2, 128, 128), (1, 16, 128), (2, 128, 3), (1, 256, 128), (1, 128, 256), (1, 3, 128), (1, 128, 1),
               (3, 64, 64), (1, 48, 96)]


@pytest.mark.parametrize("B,C,N", CONV_SHAPES)
def test_conv3x3_fwd_exact(ops, B, C, N):
    rng = np.random.default_rng(B * 1000 + C + N)
    x = ints(rng, (B, 32, 32, C))
    w = ints(rng, (3, 3, C, N), -2, 3)
    bias = ints(rng, (N,))
    cb = ints(rng, (B, N))
    res = ints(rng, (B, 32, 32, N))
    ref = onp.conv3x3(x, w, bias) + cb[:, None, None, :] + res
    y = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w), dev(bias), dev(cb), dev(res).view(B, 1024, N))
    assert np.array_equal(y.cpu().numpy().reshape(ref.shape).astype(np.float64), ref)
    # per-pixel cond bias, no residual
    cb2 = ints(rng, (B, 32, 32, N))
    ref2 = onp.conv3x3(x, w, None) + cb2
    y2 = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w), None, dev(cb2).view(B, 1024, N), None)
    assert np.array_equal(y2.cpu().numpy().reshape(ref2.shape).astype(np.float64), ref2)


@pytest.mark.parametrize("B,C,N", CONV_SHAPES)
def test_conv3x3_grads_exact(ops, B, C, N):
    rng = np.random.default_rng(7 * B + C * N)
    x = torch.tensor(ints(rng, (B, 32, 32, C)), requires_grad=True)
    w = torch.tensor(ints(rng, (3, 3, C, N), -2, 3), requires_grad=True)
    dy = torch.tensor(ints(rng, (B, 32, 32, N), -2, 3))
    y = tr.conv3x3(x, {"kernel": w})
    y.backward(dy)
    dx = ops.conv3x3_dgrad_raw(dev(dy).view(B, 1024, N), dev(w.detach()))
    dw = ops.conv3x3_wgrad_raw(dev(x.detach()).view(B, 1024, C), dev(dy).view(B, 1024, N))
    assert np.array_equal(dx.cpu().numpy().reshape(B, 32, 32, C).astype(np.float64), x.grad.numpy())
    assert np.array_equal(dw.cpu().numpy().astype(np.float64), w.grad.numpy())


@pytest.mark.parametrize("B,C,N", [(2, 128, 3), (1, 256, 1), (3, 128, 4), (1, 256, 2), (1, 3, 128), (2, 1, 256), (1, 4, 128)])
def test_conv3x3_thin_ends_exact(ops, B, C, N):
    """conv_out (E -> 3 / 1, ldm/model_vdm.py:378-383) and its gradients on the vector-ALU kernels (exact fp32 FMAs, one pass
    over the wide tensor): integers exact with / without bias and residual, the same bits on random data as the MFMA
    kernels' float64-checked results to fp32 rounding, and really the thin kernels (dev switch 8 = 1 selects the old ones)."""
    rng = np.random.default_rng(11 * B + C + 7 * N)
    x, w = ints(rng, (B, 32, 32, C)), ints(rng, (3, 3, C, N), -2, 3)
    bias, res = ints(rng, (N,)), ints(rng, (B, 32, 32, N))
    for bb, rr in ((bias, res), (None, None), (bias, None)):
        ref = onp.conv3x3(x, w, bb) + (rr if rr is not None else 0)
        y = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w), dev(bb) if bb is not None else None, None,
                            dev(rr).view(B, 1024, N) if rr is not None else None)
        assert np.array_equal(y.cpu().numpy().reshape(ref.shape).astype(np.float64), ref)
    xf = rng.standard_normal((B, 32, 32, C))
    wf = rng.standard_normal((3, 3, C, N)) / math.sqrt(9 * C)
    ref = onp.conv3x3(xf, wf)
    y = ops.conv3x3_raw(dev(xf).view(B, 1024, C), dev(wf)).cpu().numpy().reshape(ref.shape)
    assert rel_err(y, ref) < 2e-6
    if N <= 4:       # the weight gradient of the thin-output layer
        dy = ints(rng, (B, 32, 32, N), -2, 3)
        xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True)
        tr.conv3x3(xt, {"kernel": wt}).backward(torch.tensor(dy))
        dw = ops.conv3x3_wgrad_raw(dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N))
        assert np.array_equal(dw.cpu().numpy().astype(np.float64), wt.grad.numpy())
        dx = ops.conv3x3_dgrad_raw(dev(dy).view(B, 1024, N), dev(w))
        assert np.array_equal(dx.cpu().numpy().reshape(B, 32, 32, C).astype(np.float64), xt.grad.numpy())
        ops.call("mulan_set_tuning", 8, 1)
        try:
            dw_old = ops.conv3x3_wgrad_raw(dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N))
        finally:
            ops.call("mulan_set_tuning", 8, 0)
        assert torch.equal(dw_old, dw)


def test_conv3x3_thin_ends_at_the_bench_batch(ops):
    """the same kernels at the bench batch (B = 128: 8 row blocks per image x 128 images, 64-bit offsets): forward, input
    gradient and weight gradient against the padded MFMA kernels (dev switch 8) on random data, fp32 rounding apart"""
    torch.manual_seed(4)
    B, C, N = 128, 128, 3
    x, w = torch.randn(B, 1024, C, device="cuda"), torch.randn(3, 3, C, N, device="cuda") * 0.05
    bias, res, dy = torch.randn(N, device="cuda"), torch.randn(B, 1024, N, device="cuda"), torch.randn(B, 1024, N, device="cuda")
    out = {}
    for old in (0, 1):
        ops.call("mulan_set_tuning", 8, old)
        try:
            out[old] = (ops.conv3x3_raw(x, w, bias, None, res), ops.conv3x3_dgrad_raw(dy, w), ops.conv3x3_wgrad_raw(x, dy))
        finally:
            ops.call("mulan_set_tuning", 8, 0)
    for a_, r_ in zip(out[0], out[1]):
        assert float((a_ - r_).abs().max()) <= 3e-6 * float(r_.abs().max())


def test_conv3x3_float_tolerance(ops):
    """random fp32 data: fp32 MFMA accumulation vs float64, rel 1e-5 of the output scale"""
    rng = np.random.default_rng(0)
    B, C, N = 2, 128, 128
    x = rng.standard_normal((B, 32, 32, C))
    w = rng.standard_normal((3, 3, C, N)) / math.sqrt(9 * C)
    ref = onp.conv3x3(x, w)
    y = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w)).cpu().numpy().reshape(ref.shape)
    assert rel_err(y, ref) < 1e-5


def test_conv3x3_autograd_function(ops):
    rng = np.random.default_rng(3)
    B, C, N = 2, 128, 128
    x = torch.tensor(ints(rng, (B, 1024, C)), dtype=torch.float64, requires_grad=True)
    w = torch.tensor(ints(rng, (3, 3, C, N), -1, 2), dtype=torch.float64, requires_grad=True)
    b = torch.tensor(ints(rng, (N,)), requires_grad=True)
    cb = torch.tensor(ints(rng, (B, N)), requires_grad=True)
    res = torch.tensor(ints(rng, (B, 1024, N)), requires_grad=True)
    dy = torch.tensor(ints(rng, (B, 1024, N), -1, 2))
    y = tr.conv3x3(x.view(B, 32, 32, C), {"kernel": w, "bias": b}).view(B, 1024, N) + cb[:, None, :] + res
    y.backward(dy)
    gx, gw, gb, gcb, gres = (t.detach().float().cuda().requires_grad_() for t in (x, w, b, cb, res))
    out = ops.conv3x3(gx, gw, gb, cbias=gcb, res=gres)
    out.backward(dev(dy))
    assert np.array_equal(out.detach().cpu().double().numpy(), y.detach().numpy())
    for g, r in ((gx, x), (gw, w), (gb, b), (gcb, cb), (gres, res)):
        assert np.array_equal(g.grad.cpu().double().numpy(), r.grad.numpy())


# ------------------------------------------------------------------------------ gemm
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (130, 50, 178), (2, 512, 129), (1024, 1024, 128), (64, 3072, 50),
                                   (7, 5, 3), (4096, 128, 256)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_exact(ops, M, N, K, ta, tb):
    rng = np.random.default_rng(M + 3 * N + 5 * K + ta * 2 + tb)
    A = ints(rng, (M, K))
    Bm = ints(rng, (K, N))
    bias = ints(rng, (N,))
    R = ints(rng, (M, N))
    ref = 2.0 * (A @ Bm) + bias + R
    a_dev = dev(A.T if ta else A)
    b_dev = dev(Bm.T if tb else Bm)
    out = ops.gemm_raw(a_dev, b_dev, M, N, K, bias=dev(bias), R=dev(R), transA=bool(ta), transB=bool(tb), alpha=2.0)
    assert np.array_equal(out.cpu().double().numpy(), ref)


@pytest.mark.parametrize("M,N,K,ta", [(256, 128, 131072, 1), (128, 128, 8192, 1), (50, 1024, 4096, 0), (128, 3, 65536, 1),
                                      (128, 128, 512, 0), (128, 128, 512, 1), (64, 256, 320, 0), (2, 128, 512, 0),
                                      (128, 3072, 3072, 0), (128, 1024, 1024, 0)])
def test_gemm_split_k_exact(ops, M, N, K, ta):
    """long-K weight-gradient shapes, and few-tile shapes with a few hundred k steps (the FiLM projections), take the
    split-K path (slab + fixed-order reduce): still bit exact"""
    rng = np.random.default_rng(K + M)
    A = ints(rng, (M, K), -2, 3)
    Bm = ints(rng, (K, N), -2, 3)
    bias = ints(rng, (N,))
    ref = A @ Bm + bias
    out = ops.gemm_raw(dev(A.T if ta else A), dev(Bm), M, N, K, bias=dev(bias), transA=bool(ta))
    assert np.array_equal(out.cpu().double().numpy(), ref)


def test_colsum_long_segment(ops):
    rng = np.random.default_rng(21)
    x = ints(rng, (131072, 128), -2, 3)
    assert np.array_equal(ops.colsum_raw(dev(x), 1, 131072, 128).cpu().double().numpy()[0], x.sum(axis=0))
    x3 = ints(rng, (4096, 3), -2, 3)
    assert np.array_equal(ops.colsum_raw(dev(x3), 1, 4096, 3).cpu().double().numpy()[0], x3.sum(axis=0))


def test_gemm_batched_attention_shapes(ops):
    rng = np.random.default_rng(11)
    Bt, S, C = 3, 1024, 128
    q, k = ints(rng, (Bt, S, C), -2, 3), ints(rng, (Bt, S, C), -2, 3)
    ref = np.einsum("bqc,bkc->bqk", q, k)
    out = ops.gemm_raw(dev(q), dev(k), S, S, C, transB=True, batch=Bt, sA=S * C, sB=S * C)
    assert np.array_equal(out.cpu().double().numpy(), ref)
    p = ints(rng, (Bt, S, S), 0, 2)
    ref2 = np.einsum("bqk,bqc->bkc", p, q)      # P^T @ Q  (transA)
    out2 = ops.gemm_raw(dev(p), dev(q), S, C, S, transA=True, batch=Bt, sA=S * S, sB=S * C)
    assert np.array_equal(out2.cpu().double().numpy(), ref2)


def test_linear_functions(ops):
    rng = np.random.default_rng(5)
    x1 = torch.tensor(ints(rng, (2, 1024, 128)), requires_grad=True)
    x2 = torch.tensor(ints(rng, (2, 1024, 128)), requires_grad=True)
    w = torch.tensor(ints(rng, (256, 128), -1, 2), requires_grad=True)
    b = torch.tensor(ints(rng, (128,)), requires_grad=True)
    dy = torch.tensor(ints(rng, (2, 1024, 128), -1, 2))
    y = torch.cat([x1, x2], dim=-1) @ w + b
    y.backward(dy)
    g = [t.detach().float().cuda().requires_grad_() for t in (x1, x2, w, b)]
    out = ops.linear2(*g)
    out.backward(dev(dy))
    assert np.array_equal(out.detach().cpu().double().numpy(), y.detach().numpy())
    for a, r in zip(g, (x1, x2, w, b)):
        assert np.array_equal(a.grad.cpu().double().numpy(), r.grad.numpy())


# ------------------------------------------------------------------------------ group norm
@pytest.mark.parametrize("C1,C2,act,keep", [(128, 0, 1, 1.0), (128, 128, 1, 1.0), (128, 0, 0, 1.0), (256, 0, 1, 0.9),
                                            (256, 256, 1, 1.0), (128, 0, 1, 0.9)])
def test_groupnorm_fwd_bwd(ops, C1, C2, act, keep):
    """fp32 kernel vs float64 torch autograd: 2e-5 relative on outputs and gradients"""
    rng = np.random.default_rng(C1 + C2 + act)
    B, Ct = 2, C1 + C2
    x = torch.tensor(rng.standard_normal((B, 32, 32, Ct)) * 2 + 0.5, requires_grad=True)
    p = {"scale": torch.tensor(1 + 0.2 * rng.standard_normal(Ct), requires_grad=True),
         "bias": torch.tensor(0.3 * rng.standard_normal(Ct), requires_grad=True)}
    dy = torch.tensor(rng.standard_normal((B, 32, 32, Ct)))
    seed, offset = 0x1234ABCD5678, 5 << 34
    y = tr.group_norm(x, p)
    if act:
        y = tr.swish(y)
    mask = None
    if keep < 1:
        mask = torch.tensor(onp.dropout_mask((B, 32, 32, Ct), keep, seed, offset))
        y = torch.where(mask, y / np.float64(np.float32(keep)), torch.zeros_like(y))
    y.backward(dy)
    x1 = x.detach()[..., :C1].reshape(B, 1024, C1).float().contiguous().cuda().requires_grad_()
    x2 = x.detach()[..., C1:].reshape(B, 1024, C2).float().contiguous().cuda().requires_grad_() if C2 else None
    gs, gb = p["scale"].detach().float().cuda().requires_grad_(), p["bias"].detach().float().cuda().requires_grad_()
    out = ops.group_norm(x1, x2, gs, gb, act=bool(act), keep=keep, seed=seed, offset=offset)
    out.backward(dev(dy).view(B, 1024, Ct))
    o = out.detach().cpu().numpy().reshape(B, 32, 32, Ct)
    if mask is not None:   # the Philox keep-mask must match the oracle bit for bit
        assert np.array_equal(o != 0, (mask.numpy() & (y.detach().numpy() != 0)))
    assert rel_err(o, y.detach().numpy()) < 2e-5
    gx = x.grad.numpy()
    assert rel_err(x1.grad.cpu().numpy().reshape(B, 32, 32, C1), gx[..., :C1]) < 2e-5
    if C2:
        assert rel_err(x2.grad.cpu().numpy().reshape(B, 32, 32, C2), gx[..., C1:]) < 2e-5
    assert rel_err(gs.grad.cpu().numpy(), p["scale"].grad.numpy()) < 2e-5
    assert rel_err(gb.grad.cpu().numpy(), p["bias"].grad.numpy()) < 2e-5


# ------------------------------------------------------------------------------ small kernels
def test_activations_colsum_softmax(ops):
    rng = np.random.default_rng(1)
    x = torch.tensor(rng.standard_normal((300, 70)) * 3, requires_grad=True)
    dy = torch.tensor(rng.standard_normal((300, 70)))
    for fn_ref, fn in ((tr.swish, ops.silu), (lambda v: 1e-3 + torch.nn.functional.softplus(v),
                                              lambda v: ops.softplus_shift(v, 1e-3))):
        x.grad = None
        y = fn_ref(x)
        y.backward(dy)
        g = x.detach().float().cuda().requires_grad_()
        o = fn(g)
        o.backward(dev(dy))
        assert rel_err(o.detach().cpu().numpy(), y.detach().numpy()) < 1e-6
        assert rel_err(g.grad.cpu().numpy(), x.grad.numpy()) < 1e-6
    xi = ints(rng, (6 * 1024, 130))
    cs = ops.colsum_raw(dev(xi), 6, 1024, 130).cpu().double().numpy()
    assert np.array_equal(cs, xi.reshape(6, 1024, 130).sum(axis=1))
    s = rng.standard_normal((64, 1024)) * 4
    ds = rng.standard_normal((64, 1024))
    st = torch.tensor(s, requires_grad=True)
    pt = torch.softmax(st, dim=-1)
    pt.backward(torch.tensor(ds))
    p = torch.empty(64, 1024).cuda()
    ops.call("mulan_softmax_fwd", ops.ptr(dev(s)), ops.ptr(p), 64, 1024, ops.stream())
    assert rel_err(p.cpu().numpy(), pt.detach().numpy()) < 1e-6
    g = torch.empty(64, 1024).cuda()
    ops.call("mulan_softmax_bwd", ops.ptr(p), ops.ptr(dev(ds)), ops.ptr(g), 64, 1024, ops.stream())
    assert rel_err(g.cpu().numpy(), st.grad.numpy()) < 1e-5


def test_attention(ops):
    """softmax(q k^T / sqrt(C)) v fwd + bwd vs float64 autograd, 1e-5 relative"""
    rng = np.random.default_rng(2)
    B, S, C = 2, 1024, 128
    q, k, v = (torch.tensor(rng.standard_normal((B, S, C)), requires_grad=True) for _ in range(3))
    do = torch.tensor(rng.standard_normal((B, S, C)))
    o = torch.einsum("bqk,bkc->bqc", torch.softmax(torch.einsum("bqc,bkc->bqk", q / math.sqrt(C), k), -1), v)
    o.backward(do)
    g = [t.detach().float().cuda().requires_grad_() for t in (q, k, v)]
    out = ops.attention(*g)
    out.backward(dev(do))
    assert rel_err(out.detach().cpu().numpy(), o.detach().numpy()) < 1e-5
    for a, r in zip(g, (q, k, v)):
        assert rel_err(a.grad.cpu().numpy(), r.grad.numpy()) < 1e-5


def test_fourier_and_timestep_embedding(ops):
    rng = np.random.default_rng(4)
    B = 2
    z = rng.standard_normal((B, 1024, 3)) * 1.5
    zt = torch.tensor(z, requires_grad=True)
    ref = torch.cat([zt, tr.fourier_features(zt)], dim=-1)
    dout = rng.standard_normal((B, 1024, 16))
    ref.backward(torch.tensor(dout[..., :15]))
    g = dev(z).requires_grad_()
    out = ops.fourier_features(g)
    out.backward(dev(dout))
    o = out.detach().cpu().numpy()
    assert np.all(o[..., 15] == 0)
    # arguments reach ~2000 rad: compare with the fp32-rounded oracle tightly and float64 loosely
    f32 = np.concatenate([z.astype(np.float32), onp.fourier_features(z.astype(np.float32), np.float32)], axis=-1)
    assert np.abs(o[..., :15] - f32).max() < 2e-6
    assert np.abs(o[..., :15] - ref.detach().numpy()).max() < 5e-4
    # d/dz multiplies the ~1e-4 fp32 argument-rounding error of sin/cos(804 z) by w = 804
    assert rel_err(g.grad.cpu().numpy(), zt.grad.numpy()) < 1e-3
    # timestep embedding + conditioning concat
    t = rng.uniform(0, 1, size=8)
    cond = rng.standard_normal((8, 50))
    tt = torch.tensor(t, requires_grad=True)
    ct = torch.tensor(cond, requires_grad=True)
    ref = torch.cat([tr.timestep_embedding(tt, 128), ct], dim=1)
    dd = rng.standard_normal((8, 178))
    ref.backward(torch.tensor(dd))
    gt_, gc = dev(t).requires_grad_(), dev(cond).requires_grad_()
    out = ops.cond_input(gt_, gc, 128)
    out.backward(dev(dd))
    e32 = onp.timestep_embedding(t.astype(np.float32), 128, np.float32)
    assert np.abs(out.detach().cpu().numpy()[:, :128] - e32).max() < 2e-4
    assert np.abs(out.detach().cpu().numpy()[:, :128] - ref.detach().numpy()[:, :128]).max() < 5e-4
    assert np.array_equal(out.detach().cpu().numpy()[:, 128:], cond.astype(np.float32))
    assert rel_err(gt_.grad.cpu().numpy(), tt.grad.numpy()) < 2e-3
    assert rel_err(gc.grad.cpu().numpy(), ct.grad.numpy()) < 1e-6


# ------------------------------------------------------------------------------ MuLAN closed forms
def _abc(rng, B):
    a = rng.standard_normal((B, 3072)) * 0.5
    b = rng.standard_normal((B, 3072)) * 0.5
    c = 1e-3 + np.logaddexp(rng.standard_normal((B, 3072)), 0)
    return a, b, c


def test_poly_gamma(ops):
    rng = np.random.default_rng(6)
    B = 4
    a, b, c = _abc(rng, B)
    t = rng.uniform(0, 1, B)
    ta, tb, tc = (torch.tensor(v, requires_grad=True) for v in (a, b, c))
    tt = torch.tensor(t)
    gt = tr.poly_gamma(ta, tb, tc, tt)
    gp = tr.poly_gamma_grad_t(ta, tb, tc, tt)
    d1, d2 = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    (gt * torch.tensor(d1)).sum().backward(retain_graph=True)
    (gp * torch.tensor(d2)).sum().backward()
    ga, gb, gc = (dev(v).requires_grad_() for v in (a, b, c))
    g0, g1, ogt, ogp = ops.poly_gamma(ga, gb, gc, dev(t), -13.3, 5.0)
    (ogt * dev(d1)).sum().backward(retain_graph=True)
    (ogp * dev(d2)).sum().backward()
    assert np.abs(g0.cpu().numpy() + 13.3).max() < 1e-5 and np.abs(g1.cpu().numpy() - 5.0).max() < 1e-5
    # fp32 evaluation of the degree-5 polynomial ratio (same operation order as the reference) vs float64
    assert rel_err(ogt.detach().cpu().numpy(), gt.detach().numpy()) < 5e-5
    assert rel_err(ogp.detach().cpu().numpy(), gp.detach().numpy()) < 5e-5
    for g, r in ((ga, ta), (gb, tb), (gc, tc)):
        assert rel_err(g.grad.cpu().numpy(), r.grad.numpy()) < 1e-4


@pytest.mark.parametrize("per_elem", [True, False])
def test_qsample(ops, per_elem):
    rng = np.random.default_rng(8)
    B = 3
    x = rng.integers(0, 256, (B, 3072)).astype(np.uint8)
    shape = (B, 3072) if per_elem else (B,)
    g0 = -13.3 + 0.5 * rng.standard_normal(shape)
    g1 = 5.0 + 0.5 * rng.standard_normal(shape)
    gt = rng.uniform(-12, 4, shape)
    e0, e = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    T = lambda v: torch.tensor(v, requires_grad=True)
    t0, t1, tt = T(g0), T(g1), T(gt)
    bc = (lambda v: v) if per_elem else (lambda v: v[:, None])
    f = tr.encode(torch.tensor(x, dtype=torch.float64))
    xi = torch.tensor(x.astype(np.int64))
    recon = -tr.logprob(xi, f + torch.exp(0.5 * bc(t0)) * torch.tensor(e0), bc(t0) * torch.ones(B, 3072, dtype=torch.float64))
    v1 = torch.sigmoid(bc(t1)) * torch.ones(B, 3072, dtype=torch.float64)
    klz = 0.5 * ((1 - v1) * f * f + v1 - torch.log(v1) - 1).sum(1)
    vt = torch.sigmoid(bc(tt)) * torch.ones(B, 3072, dtype=torch.float64)
    zt = torch.sqrt(1 - vt) * f + torch.sqrt(vt) * torch.tensor(e)
    gbar = (bc(tt) * torch.ones(B, 3072, dtype=torch.float64)).mean(1)
    dz, dgb = rng.standard_normal((B, 3072)), rng.standard_normal(B)
    dr, dk = rng.standard_normal(B), rng.standard_normal(B)
    ((zt * torch.tensor(dz)).sum() + (gbar * torch.tensor(dgb)).sum() + (recon * torch.tensor(dr)).sum()
     + (klz * torch.tensor(dk)).sum()).backward()
    d0, d1, dt = (dev(v).requires_grad_() for v in (g0, g1, gt))
    ozt, ogbar, orec, oklz, ov0, ov1 = ops.qsample(torch.tensor(x).cuda(), d0, d1, dt, dev(e0), dev(e))
    ((ozt * dev(dz)).sum() + (ogbar * dev(dgb)).sum() + (orec * dev(dr)).sum() + (oklz * dev(dk)).sum()).backward()
    assert rel_err(ozt.detach().cpu().numpy(), zt.detach().numpy()) < 1e-6
    assert rel_err(ogbar.detach().cpu().numpy(), gbar.detach().numpy()) < 1e-6
    assert rel_err(orec.detach().cpu().numpy(), recon.detach().numpy()) < 1e-5
    assert rel_err(oklz.detach().cpu().numpy(), klz.detach().numpy()) < 1e-5
    assert abs(float(ov1.mean().detach()) - float(v1.mean().detach())) < 1e-6
    assert rel_err(dt.grad.cpu().numpy(), tt.grad.numpy()) < 1e-4
    assert rel_err(d0.grad.cpu().numpy(), t0.grad.numpy()) < 1e-3
    assert rel_err(d1.grad.cpu().numpy(), t1.grad.numpy()) < 1e-4


@pytest.mark.parametrize("g0_mean", [-13.3, -9.0, -6.0, -2.0, 3.0])
def test_qsample_bin_window_is_the_sum_over_all_bins(ops, g0_mean):
    """the reconstruction term's softmax over the 256 bins (ldm/model_vdm.py:269-296 under :100-108) is summed over the
    bins whose term is not exactly 0.0f only (vdm_loss.hip bin_window): bit for bit the sum over all bins (dev switch
    tune[24]) -- forward values and the gamma_0 gradient, from the reference's gamma_0 = -13.3 (bins 6 standard deviations
    apart) to noise wider than the whole range, with outliers far outside [-1, 1]"""
    rng = np.random.default_rng(int(-g0_mean * 10) + 200)
    B = 5
    x = rng.integers(0, 256, (B, 3072)).astype(np.uint8)
    x[0, :16] = 0
    x[0, 16:32] = 255
    g0 = g0_mean + 0.7 * rng.standard_normal((B, 3072))
    g1, gt = 5.0 + 0.5 * rng.standard_normal((B, 3072)), rng.uniform(-12, 4, (B, 3072))
    e0, e = rng.standard_normal((B, 3072)), rng.standard_normal((B, 3072))
    e0[0, :32] *= 50.0                                        # z_0 far outside the bins
    e0[1, :8] = 0.0
    dr = rng.standard_normal(B)

    def run(all_bins):
        ops.call("mulan_set_tuning", 24, all_bins)
        d0, d1, dt = (dev(v).requires_grad_() for v in (g0, g1, gt))
        out = ops.qsample(torch.tensor(x).cuda(), d0, d1, dt, dev(e0), dev(e))
        (out[2] * dev(dr)).sum().backward()
        return [o.detach().clone() for o in out] + [d0.grad.clone()]

    try:
        ref = run(1)
        got = run(0)
    finally:
        ops.call("mulan_set_tuning", 24, 0)
    assert bool(torch.isfinite(ref[2]).all())
    for i, (a, r) in enumerate(zip(got, ref)):
        assert torch.equal(a, r), (i, float((a - r).abs().max()))


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("per_elem", [True, False])
def test_diffusion_loss(ops, mode, per_elem):
    rng = np.random.default_rng(9 + mode)
    B = 3
    x = rng.integers(0, 256, (B, 3072)).astype(np.uint8)
    shape = (B, 3072) if per_elem else (B,)
    gt = rng.uniform(-10, 4, shape)
    gp = rng.uniform(5, 30, shape)
    e, zt, net = (rng.standard_normal((B, 3072)) for _ in range(3))
    T = lambda v: torch.tensor(v, requires_grad=True)
    tg, tp, tz, tn = T(gt), T(gp), T(zt), T(net)
    bc = (lambda v: v) if per_elem else (lambda v: v[:, None])
    f = tr.encode(torch.tensor(x, dtype=torch.float64))
    te = torch.tensor(e)
    if mode == 2:
        loss = 0.5 * (bc(tp) * (te - tn) ** 2).sum(1)
    else:
        vt = torch.sigmoid(bc(tg))
        vh = tn if mode == 0 else -torch.exp(0.5 * bc(tg)) * tz + torch.sqrt(1 + torch.exp(bc(tg))) * tn
        vs = torch.sqrt(1 - vt) * te - torch.sqrt(vt) * f
        loss = 0.5 * ((1 - vt) * bc(tp) * (vs - vh) ** 2).sum(1)
    dl = rng.standard_normal(B)
    (loss * torch.tensor(dl)).sum().backward()
    dg, dp, dz, dn = (dev(v).requires_grad_() for v in (gt, gp, zt, net))
    out = ops.diffusion_loss(mode, torch.tensor(x).cuda(), dg, dp, dev(e), dz, dn)
    (out * dev(dl)).sum().backward()
    assert rel_err(out.detach().cpu().numpy(), loss.detach().numpy()) < 1e-5
    assert rel_err(dn.grad.cpu().numpy(), tn.grad.numpy()) < 1e-5
    assert rel_err(dp.grad.cpu().numpy(), tp.grad.numpy()) < 1e-5
    if mode != 2:
        assert rel_err(dg.grad.cpu().numpy(), tg.grad.numpy()) < 1e-4
    if mode == 1:
        assert rel_err(dz.grad.cpu().numpy(), tz.grad.numpy()) < 1e-5


def test_topk_embedding(ops):
    rng = np.random.default_rng(10)
    B, L, k = 16, 50, 15
    logits = rng.standard_normal((B, L)) * 2
    raw = rng.gamma(1.0 / k, size=(10, B, L))
    tl = torch.tensor(logits, requires_grad=True)
    emb, kl = tr.topk_embedding_and_loss(tl, torch.tensor(raw), k)
    de, dk = rng.standard_normal((B, L)), rng.standard_normal(B)
    ((emb * torch.tensor(de)).sum() + (kl * torch.tensor(dk)).sum()).backward()
    gl = dev(logits).requires_grad_()
    oe, ok = ops.topk_embedding(gl, dev(raw), k)
    ((oe * dev(de)).sum() + (ok * dev(dk)).sum()).backward()
    o = oe.detach().cpu().numpy()
    assert np.array_equal(np.round(o), np.round(emb.detach().numpy()))       # same hard top-k set
    assert np.all(np.round(o).sum(axis=1) == k)
    assert np.abs(o - emb.detach().numpy()).max() < 1e-6
    assert rel_err(ok.detach().cpu().numpy(), kl.detach().numpy()) < 1e-5
    assert rel_err(gl.grad.cpu().numpy(), tl.grad.numpy()) < 1e-4


def test_adamw_ema_matches_oracle(ops):
    rng = np.random.default_rng(12)
    n, n_decay = 10007, 6000
    p, g = rng.standard_normal(n), rng.standard_normal(n)
    m, v = 0.1 * rng.standard_normal(n), np.abs(rng.standard_normal(n)) * 0.01
    ema = p + 0.01 * rng.standard_normal(n)
    mask = (np.arange(n) < n_decay).astype(np.float64)
    rp, rm, rv, re_ = onp.adamw_ema_step(p, 0.5 * g, m, v, ema, 2e-4, 7, mask)
    n_pad = (n + 3) // 4 * 4
    bufs = []
    for a in (p, g, m, v, ema):
        t = torch.zeros(n_pad).cuda()
        t[:n] = dev(a)
        bufs.append(t)
    ops.adamw_ema_step(bufs[0][:n], bufs[1][:n], bufs[2][:n], bufs[3][:n], bufs[4][:n], n_decay, 2e-4, 0.9, 0.99, 1e-8,
                       0.01, 7, 0.9999, grad_scale=0.5)
    for got, ref in zip((bufs[0], bufs[2], bufs[3], bufs[4]), (rp, rm, rv, re_)):
        assert rel_err(got[:n].cpu().numpy(), ref) < 1e-6


@pytest.mark.parametrize("clip", [0.3, 1e6])
def test_adamw_with_global_norm_clipping(ops, clip):
    """optax.chain(clip_by_global_norm(c), adamw): g <- g min(1, c / ||g||) with the norm of the rank-averaged
    gradient (pre-scale 1/world = 0.5 here); clip active and inactive"""
    rng = np.random.default_rng(13)
    n, n_decay = 40000, 30000
    p, g = rng.standard_normal(n), rng.standard_normal(n) * 0.01
    m, v = 0.01 * rng.standard_normal(n), np.abs(rng.standard_normal(n)) * 1e-4
    ema = p + 0.01 * rng.standard_normal(n)
    mask = (np.arange(n) < n_decay).astype(np.float64)
    gm = 0.5 * g
    norm = np.sqrt((gm ** 2).sum())
    factor = min(1.0, clip / norm)
    rp, rm, rv, re_ = onp.adamw_ema_step(p, gm * factor, m, v, ema, 2e-4, 3, mask)
    bufs = [dev(a) for a in (p, g, m, v, ema)]
    out = ops.adamw_ema_step(bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], n_decay, 2e-4, 0.9, 0.99, 1e-8, 0.01, 3,
                             0.9999, grad_scale=0.5, clip_norm=clip)
    got_factor, got_norm = out.cpu().numpy()
    assert abs(got_norm - norm) < 1e-5 * norm and abs(got_factor - factor) < 1e-5
    assert (factor < 1.0) == (clip < 1.0)
    for got, ref in zip((bufs[0], bufs[2], bufs[3], bufs[4]), (rp, rm, rv, re_)):
        assert rel_err(got.cpu().numpy(), ref) < 2e-6


def test_randn_moments(ops):
    z = ops.randn((1 << 20,), 1234, 0, "cuda").cpu().double().numpy()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs(((z ** 4).mean()) - 3) < 5e-2
    z2 = ops.randn((1 << 20,), 1234, 0, "cuda").cpu().double().numpy()
    assert np.array_equal(z, z2)
    z3 = ops.randn((1 << 20,), 1235, 0, "cuda").cpu().double().numpy()
    assert abs(np.corrcoef(z, z3)[0, 1]) < 5e-3


def test_oracle_is_the_same_on_host_and_device(monkeypatch):
    """tests/oracle_dev.run_oracle runs the float64 oracle's own torch code with its tensors on the GPU (the heavy parity
    tests are bound by the oracle's host time).  IEEE double either way: forward terms and every parameter gradient of one
    training-mode pass agree between host and device to 1e-9 of their scale -- seven orders below any bar a HIP kernel is
    held to."""
    from oracle import torch_ref as tr
    from tests.oracle_dev import run_oracle
    from tests.test_gpu_model import block_names, make_cfg, oracle_masks
    from mulan_amd.rng import PRNGKey
    _, ocfg = make_cfg("mulan_velocity", "vdm", True)
    B = 2
    rng = np.random.default_rng(5)
    x = torch.tensor(rng.integers(0, 256, (B, 32, 32, 3)).astype(np.uint8))
    raw = torch.tensor(rng.gamma(1.0 / 15, size=(10, B, 50)))
    e0, e = (torch.tensor(rng.standard_normal((B, 32, 32, 3))) for _ in range(2))
    k_enc, k_score = PRNGKey(7).split(2)
    masks = dict(enc_masks=oracle_masks(block_names(1, False), k_enc, B, 128, 0.9),
                 score_masks=oracle_masks(block_names(1, True), k_score, B, 128, 0.9))
    keep = float(np.float32(0.9))
    res = {}
    for dev_name in ("cpu", "cuda"):
        monkeypatch.setenv("MULAN_ORACLE_DEVICE", dev_name)
        params = tr.init_params(ocfg, seed=21, dtype=torch.float64)
        for _, leaf in tr.tree_leaves(params):
            leaf.requires_grad_(True)
        out = run_oracle(lambda P, *a, **k: tr.mulan_forward(P, ocfg, *a, keep=keep, **k), params, x, 0.3, raw, e0, e,
                         backward="bpd", **masks)
        res[dev_name] = (out, {"/".join(p): l.grad.clone() for p, l in tr.tree_leaves(params) if l.grad is not None})
    (oc, gc), (od, gd) = res["cpu"], res["cuda"]
    for k in ("bpd", "loss_recon", "loss_klz", "loss_diff"):
        assert float((oc[k] - od[k]).abs().max()) <= 1e-9 * float(oc[k].abs().max()), k
    assert sorted(gc) == sorted(gd) and len(gc) > 50
    gmax = max(float(v.abs().max()) for v in gc.values())
    for k in gc:      # (1e-13 of the largest gradient: the key bias of the attention block has a gradient that is zero
        #                 identically -- softmax ignores a constant added to a row of scores -- and holds rounding noise)
        assert float((gc[k] - gd[k]).abs().max()) <= 1e-9 * float(gc[k].abs().max()) + 1e-13 * gmax, k
