"""The 3-pass fp16 split convolution (two power-of-two-scaled fp16 pieces per fp32 operand; fp32-equivalent products on
the fp16 matrix cores) against the same oracle and tolerances as the exact-fp32 MFMA kernel: bit-exact on integer data,
on random data an error vs float64 of the same order as the fp32 kernel's own (2^-23-ish relative to sum |a||b|), and
no loss for images of very different magnitude inside one batch (per-image scales)."""
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
    monkeypatch.setattr(_ops, "CONV_MODE", "f16x3")
    return _ops


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).float().cuda()


@pytest.mark.parametrize("B,C,N", [(2, 128, 128), (1, 256, 128), (1, 128, 256), (1, 16, 128), (3, 48, 128)])
def test_f16x3_conv_exact_on_integers(ops, B, C, N):
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
def test_f16x3_dgrad_exact_on_integers(ops, B, C, N):
    rng = np.random.default_rng(C * 3 + N)
    x = torch.tensor(rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64), requires_grad=True)
    w = torch.tensor(rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64))
    dy = torch.tensor(rng.integers(-2, 3, (B, 32, 32, N)).astype(np.float64))
    tr.conv3x3(x, {"kernel": w}).backward(dy)
    dx = ops.conv3x3_dgrad_raw(dev(dy).view(B, 1024, N), dev(w))
    assert np.array_equal(dx.cpu().double().numpy().reshape(B, 32, 32, C), x.grad.numpy())


@pytest.mark.parametrize("B,C,N", [(2, 128, 128), (1, 256, 128), (3, 64, 64), (1, 16, 128), (2, 48, 96), (1, 128, 4)])
def test_f16x3_wgrad_exact_on_integers(ops, B, C, N):
    rng = np.random.default_rng(C * 5 + N + B)
    x = torch.tensor(rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64))
    w = torch.tensor(rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64), requires_grad=True)
    dy = torch.tensor(rng.integers(-2, 3, (B, 32, 32, N)).astype(np.float64))
    tr.conv3x3(x, {"kernel": w}).backward(dy)
    dw = ops.conv3x3_wgrad_raw(dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N))
    assert np.array_equal(dw.cpu().double().numpy(), w.grad.numpy())


def test_f16x3_wgrad_accuracy(ops, monkeypatch):
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


def test_f16x3_accuracy_matches_fp32_kernel(ops, monkeypatch):
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


def test_f16x3_whole_model_parity(ops):
    """MuLAN train-mode step through the f16x3 convolutions: same parity bars as the fp32 path"""
    from tests.test_gpu_model import run_case
    run_case("mulan_velocity", "vdm", False, train=True)


def test_absmax_rows(ops):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((5, 1024, 48)).astype(np.float32)
    x[1] *= 1e-20
    x[2] = 0.0
    x[3, 1023, 47] = -77.0
    x[4, 0, 0] = 1e30
    got = ops.absmax_rows(dev(x)).cpu().numpy().view(np.float32)
    assert got.shape == (5, 16) and np.array_equal(got.max(1), np.abs(x).reshape(5, -1).max(1))


def test_groupnorm_leaves_output_maxima(ops):
    """the fused GroupNorm kernel's by-product (per-image maxima of its output) equals a pass over the output, with
    and without dropout, for 128 and 128+128 channels, and the convolution picks it up"""
    torch.manual_seed(3)
    for C1, C2, keep in ((128, 0, 1.0), (128, 128, 0.9)):
        x1 = torch.randn(3, 1024, C1, device="cuda") * torch.tensor([1.0, 1e-3, 50.0], device="cuda").view(3, 1, 1)
        x2 = torch.randn(3, 1024, C2, device="cuda") if C2 else None
        g, b = torch.randn(C1 + C2, device="cuda"), torch.randn(C1 + C2, device="cuda")
        y = ops.GroupNormFn.apply(x1, x2, g, b, 32, 1e-6, 1, keep, 7, 0)
        ymax, ver = y._absmax
        assert ver == y._version
        got = ymax.cpu().numpy().view(np.float32).max(1)
        assert np.array_equal(got, y.abs().reshape(3, -1).amax(1).cpu().numpy())
        assert ops.cached_absmax(y) is ymax
        y.add_(1.0)                                       # modified in place: the cached maxima are stale
        assert ops.cached_absmax(y) is not ymax


def test_f16x3_per_image_dynamic_range(ops):
    """images whose magnitudes differ by 10^14 in one batch: every image keeps fp32-level accuracy (fwd and dgrad)"""
    rng = np.random.default_rng(11)
    B, C, N = 4, 128, 128
    x = rng.standard_normal((B, 32, 32, C))
    x *= np.array([1e-8, 1.0, 1e6, 3e-3])[:, None, None, None]
    w = rng.standard_normal((3, 3, C, N)) / math.sqrt(9 * C)
    x32, w32 = x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64)
    ref, mag = onp.conv3x3(x32, w32), onp.conv3x3(np.abs(x32), np.abs(w32))
    y = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w)).cpu().double().numpy().reshape(ref.shape)
    for b in range(B):
        assert float((np.abs(y[b] - ref[b]) / mag[b]).max()) < 3e-6, b
    wt = torch.tensor(w32)
    xt = torch.tensor(x32, requires_grad=True)
    dy = torch.tensor(np.ascontiguousarray(np.broadcast_to(x32[..., :1], x32.shape)))   # same per-image magnitudes
    tr.conv3x3(xt, {"kernel": wt}).backward(dy)
    dx = ops.conv3x3_dgrad_raw(dev(dy.numpy()).view(B, 1024, N), dev(w)).cpu().double().numpy().reshape(x.shape)
    for b in range(B):
        err = np.abs(dx[b] - xt.grad[b].numpy()).max() / np.abs(xt.grad[b].numpy()).max()
        assert err < 1e-5, (b, err)


def test_f16x3_zero_operands(ops):
    """all-zero weights (the reference zero-initialises the second ResBlock convolution) and all-zero inputs"""
    B, C, N = 2, 128, 128
    x = torch.randn(B, 1024, C, device="cuda")
    z = ops.conv3x3_raw(x, torch.zeros(3, 3, C, N, device="cuda"))
    assert torch.count_nonzero(z) == 0
    z = ops.conv3x3_raw(torch.zeros_like(x), torch.randn(3, 3, C, N, device="cuda"))
    assert torch.count_nonzero(z) == 0
    z = ops.conv3x3_wgrad_raw(torch.zeros_like(x), torch.randn(B, 1024, N, device="cuda"))
    assert torch.count_nonzero(z) == 0


@pytest.mark.parametrize("side", [False, True])
def test_f16x3_slab_reductions_folded_into_the_next_weight_gradient(ops, side, monkeypatch):
    """mulan_conv3x3_wgrad_f16x3_planes_fold (round 6): inside a weight_gradient_stream() scope -- the backward pass of a
    train step -- a 3x3 weight gradient writes its slabs and leaves them to the NEXT one, whose blocks sum them in their
    prologue (up to two sets per launch); what is pending at the end of the scope goes to mulan_slab_reduce.  A chain of
    four weight gradients of three shapes (and one extra flush in the middle, as a completed gradient bucket forces it)
    gives the very bits of the four stand-alone launches with their own reduction kernels -- same slabs, same summation
    order -- with the weight-gradient stream on and off; on integer data both are exact."""
    import mulan_amd.ops as O
    rng = np.random.default_rng(5)
    B = 3
    cases = []
    for C, N in ((128, 128), (256, 128), (128, 128), (128, 256)):
        x = rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64)
        dy = rng.integers(-2, 3, (B, 32, 32, N)).astype(np.float64)
        w = rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64)
        xd, dyd, wd = dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N), dev(w)
        xmax, dymax = ops.absmax_rows(xd), ops.absmax_rows(dyd)
        _, xs = ops.conv3x3_raw(xd, wd, xmax=xmax, planes=True)
        _, dys = ops.conv3x3_dgrad_raw(dyd, wd, dymax=dymax, planes=True)
        xt, wt = torch.tensor(x), torch.tensor(w, requires_grad=True)
        tr.conv3x3(xt, {"kernel": wt}).backward(torch.tensor(dy))
        cases.append((xs, xmax, dys, dymax, C, N, wt.grad.numpy()))
    monkeypatch.setattr(O, "SIDE_STREAM", side)
    monkeypatch.setattr(O, "FOLD_SLAB_REDUCE", True)      # (opt-in: measured +-0 on the train step, ops.FOLD_SLAB_REDUCE)
    monkeypatch.setattr(O, "SIDE_WGRAD_SHARE", False)     # the same split count inside and outside the scope
    alone = [ops.conv3x3_wgrad_planes_raw(xs, xm, dys, dm, B, C, N, out=torch.empty(3, 3, C, N, device="cuda"))
             for xs, xm, dys, dm, C, N, _ in cases]
    outs = [torch.full((3, 3, C, N), float("nan"), device="cuda") for _, _, _, _, C, N, _ in cases]
    names = []
    orig = O.call
    monkeypatch.setattr(O, "call", lambda name, *a: (names.append(name), orig(name, *a))[1])
    with O.weight_gradient_stream():
        for i, ((xs, xm, dys, dm, C, N, _), o) in enumerate(zip(cases, outs)):
            launch = lambda xs=xs, xm=xm, dys=dys, dm=dm, C=C, N=N, o=o: ops.conv3x3_wgrad_planes_raw(xs, xm, dys, dm, B, C, N, out=o)
            if side:
                O._on_side(launch, (xs, xm, dys, dm))
            else:
                launch()
            if i == 1:
                O.flush_slab_reductions()           # (a completed gradient bucket: parallel.GradReducer._launch)
                assert not O._SLAB_PENDING
        assert len(O._SLAB_PENDING) == 1            # the third set went into the fourth launch; the fourth is pending
    torch.cuda.synchronize()
    assert not O._SLAB_PENDING and not O._SLAB_KEEP
    assert names.count("mulan_conv3x3_wgrad_f16x3_planes_fold") == 4 and names.count("mulan_slab_reduce") == 2
    assert "mulan_conv3x3_wgrad_f16x3_planes" not in names
    for (_, _, _, _, C, N, ref), a, o in zip(cases, alone, outs):
        assert torch.equal(a, o)
        assert np.array_equal(o.cpu().double().numpy(), ref)


@pytest.mark.parametrize("B,C,N,ints", [(2, 128, 128, True), (3, 256, 128, True), (1, 128, 256, True),
                                         (4, 128, 128, False),
                                         # round 6 (eight-wave block, one-dimensional grid): split counts that are not a
                                         # multiple of 8 -- S = 85 at one tile (the remainder group of w8_decode), 42 at
                                         # two, 21 at four -- with ragged pixel ranges that cross image boundaries
                                         (6, 128, 128, True), (11, 256, 128, True), (7, 256, 256, True), (9, 128, 128, False)])
def test_f16x3_plane_fed_wgrad(ops, B, C, N, ints):
    """weight gradient from the split planes that the forward / input-gradient convolutions write as a by-product:
    bit exact on integers; on random data with very different per-image magnitudes (exercises the per-image ->
    tensor-wide rescale) within the fp32 kernel's error"""
    rng = np.random.default_rng(B * 7 + C + N)
    if ints:
        x = rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64)
        dy = rng.integers(-2, 3, (B, 32, 32, N)).astype(np.float64)
        x[0] *= 4.0                      # different power-of-two scales per image
    else:
        mags = np.resize(np.array([1.0, 3e-3, 40.0, 1e-6]), B)[:, None, None, None]
        x = rng.standard_normal((B, 32, 32, C)) * mags
        dy = rng.standard_normal((B, 32, 32, N)) * mags[::-1]
    w = rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64)
    xd, dyd, wd = dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N), dev(w)
    xmax, dymax = ops.absmax_rows(xd), ops.absmax_rows(dyd)
    y, xs = ops.conv3x3_raw(xd, wd, xmax=xmax, planes=True)
    dx, dys = ops.conv3x3_dgrad_raw(dyd, wd, dymax=dymax, planes=True)
    dw = ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, B, C, N).cpu().double().numpy()
    xt = torch.tensor(x.astype(np.float32).astype(np.float64))
    wt = torch.tensor(w, requires_grad=True)
    tr.conv3x3(xt, {"kernel": wt}).backward(torch.tensor(dy.astype(np.float32).astype(np.float64)))
    ref = wt.grad.numpy()
    if ints:
        assert np.array_equal(dw, ref)
        assert np.array_equal(y.cpu().double().numpy().reshape(B, 32, 32, N), onp.conv3x3(x, w))
    else:
        d32 = ops.conv3x3_wgrad_raw(xd, dyd).cpu().double().numpy()      # the fp32-input 9-tap kernel
        scale = np.abs(ref).max()
        e_p, e_9 = np.abs(dw - ref).max() / scale, np.abs(d32 - ref).max() / scale
        assert e_p < 1e-5 and e_p < 2 * e_9 + 1e-6, (e_p, e_9)


@pytest.mark.parametrize("B,K1,K2,N", [(2, 128, 0, 128), (1, 128, 128, 128), (3, 256, 0, 256), (2, 128, 128, 256)])
def test_f16x3_linear_exact_on_integers_and_accuracy(ops, B, K1, K2, N):
    """[x1 | x2] @ w + bias + res through the f16x3 per-pixel dense kernel, and dy @ w^T split into two outputs"""
    rng = np.random.default_rng(B + K1 + K2 + N)
    K = K1 + K2
    for ints in (True, False):
        if ints:
            x = rng.integers(-3, 4, (B, 1024, K)).astype(np.float64)
            w = rng.integers(-2, 3, (K, N)).astype(np.float64)
            bias = rng.integers(-3, 4, N).astype(np.float64)
            res = rng.integers(-3, 4, (B, 1024, N)).astype(np.float64)
            dy = rng.integers(-2, 3, (B, 1024, N)).astype(np.float64)
        else:
            x = rng.standard_normal((B, 1024, K)) * np.array([1.0, 1e-4, 300.0])[:B, None, None]
            w = rng.standard_normal((K, N)) / math.sqrt(K)
            bias, res = rng.standard_normal(N), rng.standard_normal((B, 1024, N))
            dy = rng.standard_normal((B, 1024, N)) * np.array([1e-5, 1.0, 7.0])[:B, None, None]
        f32 = lambda a: a.astype(np.float32).astype(np.float64)
        x1d = dev(x[..., :K1])
        x2d = dev(x[..., K1:]) if K2 else None
        wd = dev(w)
        wp, wmax = ops.linear_pack(wd, False)
        y, _ = ops.linear_f16x3_raw(x1d, x2d, wp, wmax, N, 0, bias=dev(bias), res=None if K2 else dev(res))
        ref = f32(x) @ f32(w) + f32(bias) + (0 if K2 else f32(res))
        wpt, _ = ops.linear_pack(wd, True, wmax)
        dx1, dx2 = ops.linear_f16x3_raw(dev(dy), None, wpt, wmax, K1, K2)
        dxr = f32(dy) @ f32(w).T
        got_dx = dx1.cpu().double().numpy() if not K2 else np.concatenate([dx1.cpu().double().numpy(),
                                                                         dx2.cpu().double().numpy()], -1)
        if ints:
            assert np.array_equal(y.cpu().double().numpy(), ref)
            assert np.array_equal(got_dx, dxr)
        else:
            for b in range(B):   # per image: the scales are per image
                mag = np.abs(f32(x[b])) @ np.abs(f32(w)) + np.abs(f32(bias)) + (0 if K2 else np.abs(f32(res[b])))
                assert float((np.abs(y[b].cpu().double().numpy() - ref[b]) / mag).max()) < 2e-6, b
                magd = np.abs(f32(dy[b])) @ np.abs(f32(w)).T
                assert float((np.abs(got_dx[b] - dxr[b]) / magd).max()) < 2e-6, b


@pytest.mark.parametrize("B,K1,K2,N", [(3, 128, 128, 128), (2, 256, 0, 256), (5, 128, 0, 128)])
def test_f16x3_linear_row_tile_variants_are_bit_identical(ops, B, K1, K2, N):
    """64-row blocks (round 5: launches with fewer 128-row blocks than CUs -- a 16-image sampling batch) against 128-row
    blocks (dev switch tune[28] = 1): the same sum in the same order for every output element, the same plane by-product"""
    torch.manual_seed(B + K1 + N)
    x1 = torch.randn(B, 1024, K1, device="cuda") * 2
    x2 = torch.randn(B, 1024, K2, device="cuda") if K2 else None
    w = torch.randn(K1 + K2, N, device="cuda") / 16
    bias = torch.randn(N, device="cuda")
    res = torch.randn(B, 1024, N, device="cuda")
    wp, wmax = ops.linear_pack(w, False)
    outs = []
    try:
        for v in (1, 0):
            ops.call("mulan_set_tuning", 28, v)
            r = ops.linear_f16x3_raw(x1, x2, wp, wmax, N, 0, bias=bias, res=res, planes=True)
            outs.append([t.clone() for t in r if torch.is_tensor(t)])
    finally:
        ops.call("mulan_set_tuning", 28, 0)
    assert len(outs[0]) == len(outs[1]) >= 2
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


def test_f16x3_linear_autograd_matches_fp32_gemm(ops, monkeypatch):
    torch.manual_seed(1)
    B, K, N = 2, 128, 128
    x1 = torch.randn(B, 1024, K, device="cuda", requires_grad=True)
    x2 = torch.randn(B, 1024, K, device="cuda", requires_grad=True)
    w = (torch.randn(2 * K, N, device="cuda") * 0.05).requires_grad_(True)
    w1 = (torch.randn(K, N, device="cuda") * 0.05).requires_grad_(True)
    bias = torch.randn(N, device="cuda", requires_grad=True)
    g = torch.randn(B, 1024, N, device="cuda")
    outs = {}
    for mode in ("f32", "f16x3"):
        monkeypatch.setattr(ops, "CONV_MODE", mode)
        for t in (x1, x2, w, w1, bias):
            t.grad = None
        y = ops.linear2(x1, x2, w, bias) + ops.linear(x1, w1, bias, x2)
        y.backward(g)
        outs[mode] = [y.detach().clone()] + [t.grad.clone() for t in (x1, x2, w, w1, bias)]
    for a, b in zip(outs["f32"], outs["f16x3"]):
        assert float((a - b).abs().max() / a.abs().max()) < 1e-5


def test_conv_and_gn_backward_leave_maxima(ops):
    """by-products: the f16x3 convolution leaves the maxima of its output, GroupNorm backward those of its input
    gradients; both equal a pass over the tensor"""
    torch.manual_seed(5)
    B, C, N = 3, 128, 256
    x = torch.randn(B, 1024, C, device="cuda") * torch.tensor([1.0, 1e-3, 30.0], device="cuda").view(3, 1, 1)
    w = torch.randn(3, 3, C, N, device="cuda") * 0.05
    res = torch.randn(B, 1024, N, device="cuda")
    y = ops.conv3x3_raw(x, w, torch.randn(N, device="cuda"), None, res)
    got = y._absmax[0].cpu().numpy().view(np.float32).max(1)
    assert np.array_equal(got, y.abs().reshape(B, -1).amax(1).cpu().numpy())
    x1 = torch.randn(B, 1024, 128, device="cuda", requires_grad=True)
    x2 = torch.randn(B, 1024, 128, device="cuda", requires_grad=True)
    g, b = torch.randn(256, device="cuda", requires_grad=True), torch.randn(256, device="cuda", requires_grad=True)
    out = ops.group_norm(x1, x2, g, b, act=True, keep=0.9, seed=3, offset=0)
    seen = {}
    x1.register_hook(lambda t: seen.__setitem__("dx1", getattr(t, "_absmax", None)))
    x2.register_hook(lambda t: seen.__setitem__("dx2", getattr(t, "_absmax", None)))
    out.backward(torch.randn_like(out) * torch.tensor([1.0, 1e-4, 9.0], device="cuda").view(3, 1, 1))
    for name, t in (("dx1", x1.grad), ("dx2", x2.grad)):
        assert seen[name] is not None
        got = seen[name][0].cpu().numpy().view(np.float32).max(1)
        assert np.array_equal(got, t.abs().reshape(B, -1).amax(1).cpu().numpy()), name


def test_param_packer_matches_per_layer_packs(ops):
    """the once-per-step weight preparation (all leaves, two launches) is bit-identical to the per-layer maxima +
    pack calls, for both operands of a 3x3 kernel and a dense kernel, and is ignored once invalidated"""
    from mulan_amd.train_state import TrainState
    g = torch.Generator().manual_seed(0)
    tree = {"score_model": {"conv1": {"kernel": torch.randn(3, 3, 128, 128, generator=g) * 0.05,
                                      "bias": torch.randn(128, generator=g)},
                            "conv_in": {"kernel": torch.randn(3, 3, 16, 128, generator=g)},
                            "nin_shortcut": {"kernel": torch.randn(256, 128, generator=g) * 0.1},
                            "dense_big": {"kernel": torch.randn(1024, 1024, generator=g)}}}
    st = TrainState.create(apply_fn=None, variables={"params": tree}, device="cuda")
    packer = st.param_packer()
    assert packer is not None and packer.n == 3          # conv1, conv_in (forward operand only), nin_shortcut
    p = st.params["score_model"]
    wc, wi, wn = p["conv1"]["kernel"], p["conv_in"]["kernel"], p["nin_shortcut"]["kernel"]
    ref = {}
    for name, w, fn, args in (("c0", wc, ops._pack_weights, (128, 128, 0)), ("c1", wc, ops._pack_weights, (128, 128, 1)),
                              ("i0", wi, ops._pack_weights, (16, 128, 0)), ("n0", wn, ops.linear_pack, (False,)),
                              ("n1", wn, ops.linear_pack, (True,))):
        ref[name] = fn(w, *args)
    packer.refresh()
    for name, w, fn, args in (("c0", wc, ops._pack_weights, (128, 128, 0)), ("c1", wc, ops._pack_weights, (128, 128, 1)),
                              ("i0", wi, ops._pack_weights, (16, 128, 0)), ("n0", wn, ops.linear_pack, (False,)),
                              ("n1", wn, ops.linear_pack, (True,))):
        wp, wmax = fn(w, *args)
        assert wp.data_ptr() >= packer.packed.data_ptr() and wp.data_ptr() < packer.packed.data_ptr() + packer.packed.numel()
        assert torch.equal(wp, ref[name][0]), name
        assert int(wmax.max()) == int(ref[name][1].max()), name
    packer.invalidate()
    wp, _ = ops._pack_weights(wc, 128, 128, 0)
    assert not (packer.packed.data_ptr() <= wp.data_ptr() < packer.packed.data_ptr() + packer.packed.numel())


@pytest.mark.parametrize("B,N", [(3, 128), (2, 256), (5, 512)])
def test_tee_backward_leaves_the_bias_gradient_of_the_producer(ops, monkeypatch, B, N):
    """TeeFn.backward (sum of the two gradients of a block output with two consumers, mulan_add_absmax_rows_colsum) leaves 16
    partial column sums per image; the convolution that produced the tensor takes its bias gradient from them (one small
    launch) instead of a pass over the summed gradient: same bias gradient as the two-pass route to fp32 rounding, all
    other gradients identical, and the partials add up to the column sums in float64."""
    torch.manual_seed(N + B)
    x = torch.randn(B, 1024, 128, device="cuda", requires_grad=True)
    w = (torch.randn(3, 3, 128, N, device="cuda") * 0.05).requires_grad_(True)
    bias = torch.randn(N, device="cuda", requires_grad=True)
    res = torch.randn(B, 1024, N, device="cuda", requires_grad=True)
    g1, g2 = torch.randn(B, 1024, N, device="cuda"), torch.randn(B, 1024, N, device="cuda") * 3
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    monkeypatch.setattr(ops, "TEE_MAILBOX", False)      # (TeeFn: the form without a GroupNorm behind the tensor)

    def run(on):
        monkeypatch.setattr(ops, "TEE_COLSUM", on)
        for t in (x, w, bias, res):
            t.grad = None
        names.clear()
        y = ops.conv3x3(x, w, bias, None, res)
        a, b = ops.tee(y)
        ((a * g1).sum() + (b * g2).sum()).backward()
        return [t.grad.clone() for t in (x, w, bias, res)], list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    assert "mulan_add_absmax_rows_colsum" in got_names and "mulan_add_absmax_rows" in ref_names
    assert got_names.count("mulan_colsum") == 1 < ref_names.count("mulan_colsum")
    for i in (0, 1, 3):
        assert torch.equal(got[i], ref[i])
    want = (g1 + g2).double().sum((0, 1))
    scale = (g1 + g2).abs().double().sum((0, 1))
    assert float(((got[2].double() - want).abs() / scale).max()) < 2e-7
    assert float(((ref[2].double() - want).abs() / scale).max()) < 2e-7
    # the entry point itself: partials in float64 against the column sums of the sum, maxima as mulan_absmax_rows
    out = torch.empty_like(g1)
    m = torch.empty(B, 16, device="cuda", dtype=torch.int32)
    parts = torch.empty(B * 16, N, device="cuda")
    ops.call("mulan_add_absmax_rows_colsum", ops.ptr(g1), ops.ptr(g2), ops.ptr(out), ops.ptr(m), ops.ptr(parts), B, 1024 * N, N,
             ops.stream())
    assert torch.equal(out, g1 + g2)
    assert torch.equal(m.max(1).values, ops.absmax_rows(out).max(1).values)
    assert float(((parts.double().sum(0) - want).abs() / scale).max()) < 2e-7


def test_skip_connection_gradient_is_added_inside_the_groupnorm_backward(ops, monkeypatch):
    """A U-Net block output feeds the next block and a skip connection (ldm/model_vdm.py:351-372).  ops.tee / tee_take:
    the gradient that arrives through the skip connection waits in a box and the next block's GroupNorm backward kernel
    adds it (mulan_groupnorm_bwd_fused, add1b) -- no kernel of its own for the sum, and the maxima / bias-gradient
    by-products of that kernel serve the convolution in front.  Against TeeFn (mulan_add_absmax_rows): the same sum in the
    same order, so every gradient of a MuLAN train-mode pass is identical, bias gradients (summed in another order) to
    fp32 rounding."""
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg = M.VDMConfig(vocab_size=256, sample_softmax=False, antithetic_time_sampling=True, with_fourier_features=True,
                      with_attention=False, gamma_type='poly_fixedend', gamma_min=-13.3, gamma_max=5.0, sm_n_timesteps=0,
                      sm_n_embd=128, sm_n_layer=3, sm_pdrop=0.1, forward_n_layer=1, latent_size=50, latent_k=15,
                      encoder='unet', latent_type='topk', z_conditioning=True, reparam_type='true', unet_type='vdm',
                      condition='input')
    vdm = M.make_vdm("mulan_epsilon", cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(2)))
    for _, leaf in M.tree_leaves(params):
        leaf.normal_(0.0, 0.05)          # (zero-initialised layers would hide their inputs' gradients)
        leaf.requires_grad_(True)
    x = torch.randint(0, 256, (4, 32, 32, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).cuda()
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    def run(mailbox):
        monkeypatch.setattr(ops, "TEE_MAILBOX", mailbox)
        for _, leaf in M.tree_leaves(params):
            leaf.grad = None
        names.clear()
        out = vdm.apply(params, x, None, None, step=0, rngs={"sample": PRNGKey(5), "dropout": PRNGKey(6)}, deterministic=False)
        (out.loss_recon.mean() + out.loss_klz.mean() + out.loss_diff.mean()).backward()
        return {n: leaf.grad.clone() for n, leaf in M.tree_leaves(params)}, list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    n_tee = ref_names.count("mulan_add_absmax_rows_colsum") + ref_names.count("mulan_add_absmax_rows")
    assert n_tee == cfg.sm_n_layer + 1                               # one per skip connection
    assert not any(n.startswith("mulan_add_absmax_rows") for n in got_names)
    assert got_names.count("mulan_colsum") <= ref_names.count("mulan_colsum")   # (fewer in a train step: flat-buffer bias sinks)
    assert len(got_names) <= len(ref_names) - n_tee
    for n in ref:
        if n[-1] == "bias":
            assert float((got[n] - ref[n]).abs().max()) <= 2e-6 * float(ref[n].abs().max()) + 1e-12, n
        else:
            assert torch.equal(got[n], ref[n]), (n, float((got[n] - ref[n]).abs().max()))


def test_tee_mailbox_fails_loudly_when_its_order_invariant_breaks(ops, monkeypatch):
    """the tee() mailbox is only right if the skip-path gradient is deposited before the claiming consumer's
    backward takes it.  (1) the shipped order gives the two-consumer gradient; (2) the claiming consumer's backward first,
    the skip path second (torch.autograd.grad on sub-graphs): RuntimeError instead of a dropped gradient; (3) a backward
    pass through the skip path only leaves the gradient waiting: the next forward pass's tee() reports it."""
    monkeypatch.setattr(ops, "TEE_MAILBOX", True)
    torch.manual_seed(21)
    B, C = 2, 128
    g = torch.randn(C, device="cuda") * 0.2 + 1.0
    b = torch.randn(C, device="cuda") * 0.1
    w = torch.randn(3, 3, C, C, device="cuda") * 0.03

    def graph():
        x = torch.randn(B, 1024, C, device="cuda", requires_grad=True)
        h = x * 1.5                                             # a block output with two consumers
        a, skip = ops.tee(h)
        y1, s1, _ = ops.gn_conv3x3(a, None, g, b, w, act=True, skip=True)     # first consumer: claims the box
        y2 = ops.tee_take(skip) * 0.25                          # second consumer (the U-Net's up path)
        return x, h, y1, s1, y2

    # (1) the shipped order: one backward pass, the sum of both consumers' gradients reaches x
    x, h, y1, s1, y2 = graph()
    (y1.sum() + s1.sum() * 0.5 + y2.sum()).backward()
    got = x.grad.clone()
    monkeypatch.setattr(ops, "TEE_MAILBOX", False)
    # (the same graph without the mailbox, on the same x)
    x2 = x.detach().clone().requires_grad_(True)
    h2 = x2 * 1.5
    a2, skip2 = ops.tee(h2)
    z1, t1, _ = ops.gn_conv3x3(a2, None, g, b, w, act=True, skip=True)
    z2 = ops.tee_take(skip2) * 0.25
    (z1.sum() + t1.sum() * 0.5 + z2.sum()).backward()
    assert float((got - x2.grad).abs().max()) <= 1e-5 * float(x2.grad.abs().max())
    monkeypatch.setattr(ops, "TEE_MAILBOX", True)

    # (2) wrong order: the claiming consumer's backward runs before the skip-path gradient exists
    x, h, y1, s1, y2 = graph()
    torch.autograd.grad(y1.sum(), x, retain_graph=True)
    with pytest.raises(RuntimeError, match="mailbox"):
        torch.autograd.grad(y2.sum(), x, allow_unused=True)

    # (3) only the skip path is differentiated: its gradient waits in the box for a consumer that never runs
    x, h, y1, s1, y2 = graph()
    torch.autograd.grad(y2.sum(), x, allow_unused=True)
    with pytest.raises(RuntimeError, match="left waiting"):
        graph()
    graph()                                                     # (reported once; the next forward pass is clean)


def test_group_norm_skip_adds_the_skip_gradient_in_kernel(ops):
    """GroupNormSkipFn: gradients through the aliases s1 / s2 are added inside the backward kernel; the result, its
    maxima and its per-sample channel sums equal the two-step computation"""
    torch.manual_seed(9)
    B = 2
    for C2 in (0, 128):
        x1 = torch.randn(B, 1024, 128, device="cuda", requires_grad=True)
        x2 = torch.randn(B, 1024, C2, device="cuda", requires_grad=True) if C2 else None
        g = torch.randn(128 + C2, device="cuda", requires_grad=True)
        b = torch.randn(128 + C2, device="cuda", requires_grad=True)
        gy = torch.randn(B, 1024, 128 + C2, device="cuda")
        g1 = torch.randn(B, 1024, 128, device="cuda")
        g2 = torch.randn(B, 1024, 128, device="cuda") if C2 else None
        y = ops.group_norm(x1, x2, g, b, act=True, keep=0.9, seed=5, offset=64)
        loss = (y * gy).sum() + (x1 * g1).sum() + ((x2 * g2).sum() if C2 else 0)
        loss.backward()
        ref = [t.grad.clone() for t in (x1, x2, g, b) if t is not None]
        for t in (x1, x2, g, b):
            if t is not None:
                t.grad = None
        seen = {}
        x1.register_hook(lambda t: seen.__setitem__("dx1", t))
        y, s1, s2 = ops.group_norm_skip(x1, x2, g, b, act=True, keep=0.9, seed=5, offset=64)
        loss = (y * gy).sum() + (s1 * g1).sum() + ((s2 * g2).sum() if C2 else 0)
        loss.backward()
        got = [t.grad for t in (x1, x2, g, b) if t is not None]
        for a, r in zip(got, ref):
            assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max())
        dx1 = seen["dx1"]
        m = dx1._absmax[0].cpu().numpy().view(np.float32).max(1)
        assert np.array_equal(m, dx1.abs().reshape(B, -1).amax(1).cpu().numpy())
        if not C2:
            cs = dx1._colsum[0]
            refcs = dx1.double().sum(1)
            assert float((cs.double() - refcs).abs().max()) <= 2e-5 * float(dx1.abs().double().sum(1).max())


@pytest.mark.parametrize("B,C2", [(1, 0), (5, 0), (37, 128), (128, 128)])
def test_gn_backward_sums_over_samples_inside_the_launch(ops, monkeypatch, B, C2):
    """mulan_groupnorm_bwd_fused: dgamma, dbeta and the bias gradients of the convolution in front (and of the shortcut
    layer that shares its dy) are summed over the samples by the block that finishes last -- no column-sum launches.
    Equal to the separate-launch path within fp32 summation error, bit-stable across repeats (fixed summation order
    whichever block comes last), arrival counters back at zero."""
    torch.manual_seed(B + C2)
    C = 128

    def leaf(*shape, scale=1.0):
        t = (torch.randn(*shape, device="cuda") * scale).requires_grad_(True)
        t._gview = torch.zeros(*shape, device="cuda")           # the flat-gradient-buffer sink TrainState registers
        return t

    w, bias = leaf(3, 3, C, C, scale=0.05), leaf(C)
    wn, bn = leaf(C, C, scale=0.1), leaf(C)
    gamma, beta = leaf(C + C2), leaf(C + C2)
    x = torch.randn(B, 1024, C, device="cuda", requires_grad=True)
    skip = torch.randn(B, 1024, C2, device="cuda", requires_grad=True) if C2 else None
    gy = torch.randn(B, 1024, C + C2, device="cuda")
    leaves = (w, bias, wn, bn, gamma, beta)
    names = []
    real_call = ops.call
    monkeypatch.setattr(ops, "call", lambda name, *a: (names.append(name), real_call(name, *a))[1])

    def run(fused):
        monkeypatch.setattr(ops, "GN_FUSED_REDUCE", fused)
        for t in leaves + (x,):
            t.grad = None
        for t in leaves:
            t._gview.zero_()
        names.clear()
        h = ops.conv3x3(x, w, bias, None, ops.linear(x, wn, bn))
        y = ops.group_norm(h, skip, gamma, beta, act=True, keep=0.9, seed=11, offset=0)
        (y * gy).sum().backward()
        return [t.grad.clone() for t in leaves] + [x.grad.clone()], list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    assert "mulan_groupnorm_bwd_fused" in got_names and "mulan_colsum_pair" not in got_names
    assert sum(n == "mulan_colsum" for n in got_names) == 0 < sum(n == "mulan_colsum" for n in ref_names)
    for a, r, nm in zip(got, ref, ("w", "bias", "wn", "bn", "gamma", "beta", "x")):
        assert float((a - r).abs().max()) <= 2e-6 * B ** 0.5 * float(r.abs().max()) + 1e-30, nm
    assert torch.equal(got[1], got[3])                           # the shortcut bias received the very same sums
    assert bias.grad.data_ptr() == bias._gview.data_ptr()        # written in place, adopted by autograd without a copy
    for _ in range(3):
        again, _ = run(True)
        assert all(torch.equal(a, b) for a, b in zip(again, got))
    assert int(ops._gn_tickets(x.device).abs().sum()) == 0


@pytest.mark.parametrize("B,K1,K2,N", [(2, 128, 128, 128), (3, 128, 0, 128), (1, 256, 256, 256)])
def test_f16x3_dense_weight_gradient_from_planes(ops, B, K1, K2, N):
    """dw = [x1|x2]^T dy of a per-pixel dense layer from the planes handed on by its forward kernel and by the
    convolution that consumed the same dy: bit exact on integers, fp32-level error on random data"""
    rng = np.random.default_rng(B + K1 + K2 + N)
    K = K1 + K2
    for ints in (True, False):
        if ints:
            x = rng.integers(-3, 4, (B, 1024, K)).astype(np.float64)
            x[0] *= 8.0
            dy = rng.integers(-2, 3, (B, 1024, N)).astype(np.float64)
        else:
            x = rng.standard_normal((B, 1024, K)) * np.array([1.0, 1e-3, 50.0])[:B, None, None]
            dy = rng.standard_normal((B, 1024, N)) * np.array([2.0, 30.0, 1e-2])[:B, None, None]
        w = rng.integers(-2, 3, (K, N)).astype(np.float64)
        x1d = dev(x[..., :K1])
        x2d = dev(x[..., K1:]) if K2 else None
        wp, wmax = ops.linear_pack(dev(w), False)
        y, _, xs, xsmax = ops.linear_f16x3_raw(x1d, x2d, wp, wmax, N, 0, planes=True)
        dyd = dev(dy)
        dymax = ops.absmax_rows(dyd)
        wc = dev(rng.integers(-1, 2, (3, 3, N, N)).astype(np.float64))           # any 3x3 conv whose dgrad hands dy's planes on
        _, dys = ops.conv3x3_dgrad_raw(dyd, wc, dymax=dymax, planes=True)
        dw_dev = ops.linear_wgrad_planes_raw(xs, xsmax, dys, dymax, B, K, N)
        dw = dw_dev.cpu().double().numpy()
        # round 3: the same gradient with x read in fp32 and split in the kernel's staging path (the forward kernel then
        # writes no planes): the very same operand values, hence the same bits
        dw32 = ops.linear_wgrad_x32_raw(x1d, x2d, ops.absmax_rows(x1d), ops.absmax_rows(x2d) if K2 else None, dys, dymax, B, N)
        assert torch.equal(dw32, dw_dev)
        f32 = lambda a: a.astype(np.float32).astype(np.float64)
        ref = np.einsum("bpk,bpn->kn", f32(x), f32(dy))
        if ints:
            assert np.array_equal(dw, ref)
        else:
            mag = np.einsum("bpk,bpn->kn", np.abs(f32(x)), np.abs(f32(dy)))
            assert float((np.abs(dw - ref) / mag).max()) < 3e-6


# ------------------------------------------------------------------------------------------------ attention products
@pytest.mark.parametrize("B,K,N,transpose", [(3, 128, 1024, True), (2, 1024, 128, False), (2, 256, 1024, True)])
def test_f16x3_batched_linear_exact_on_integers(ops, B, K, N, transpose):
    """y[b] = x[b] @ W[b] with one packed operand and one scale per image (the attention products): bit-exact on
    integers, images of very different magnitude keep their own precision, the planes of x are handed on"""
    rng = np.random.default_rng(K + N)
    x = rng.integers(-4, 5, (B, 1024, K)).astype(np.float64)
    w = rng.integers(-3, 4, (B, K, N)).astype(np.float64)
    x[1] *= 2.0 ** 20                               # per-image scales: exactness must survive
    w[1] *= 2.0 ** -30
    ws = np.ascontiguousarray(w.transpose(0, 2, 1)) if transpose else w            # stored [B, N, K] when transposed
    xd, wd = dev(x), dev(ws)
    xm, wm = ops.absmax_rows(xd), ops.absmax_rows(wd)
    y, xs = ops.linear_batched_raw(xd, xm, ops._pack_batched(wd, transpose, wm), wm, N, planes=True)
    ref = np.einsum("brk,bkn->brn", x, w)
    assert np.array_equal(y.cpu().double().numpy(), ref)
    # out[b] = x[b]^T @ g[b] from the planes (dV = P^T dO / dK = dS^T q shapes need C, N multiples of 128)
    if K % 128 == 0:
        g = rng.integers(-3, 4, (B, 1024, 128)).astype(np.float64)
        gd = dev(g)
        gm = ops.absmax_rows(gd)
        ident = dev(np.broadcast_to(np.eye(128), (B, 128, 128)).copy())
        im = ops.absmax_rows(ident)
        _, gs = ops.linear_batched_raw(gd, gm, ops._pack_batched(ident, False, im), im, 128, planes=True)
        out = ops.bmm_tn_planes_raw(xs, xm, gs, gm, B, K, 128)
        assert np.array_equal(out.cpu().double().numpy(), np.einsum("brk,brn->bkn", x, g))


def test_scaled_softmax_and_gradient_maxima(ops):
    rng = np.random.default_rng(3)
    B, S = 2, 1024
    alpha = 1.0 / math.sqrt(128)
    s = rng.standard_normal((B * S, S)) * 30
    dp = rng.standard_normal((B * S, S))
    st = torch.tensor(s, requires_grad=True)
    pt = torch.softmax(st * alpha, dim=-1)
    pt.backward(torch.tensor(dp))
    p = torch.empty(B * S, S).cuda()
    ops.call("mulan_softmax_scaled_fwd", ops.ptr(dev(s)), ops.ptr(p), B * S, S, alpha, ops.stream())
    assert np.abs(p.cpu().double().numpy() - pt.detach().numpy()).max() < 1e-6
    g = torch.empty(B * S, S).cuda()
    rm = torch.empty(B * S).cuda()
    ops.call("mulan_softmax_scaled_bwd", ops.ptr(p), ops.ptr(dev(dp)), ops.ptr(g), B * S, S, alpha, ops.ptr(rm),
             ops.stream())
    ref = st.grad.numpy()
    assert np.abs(g.cpu().double().numpy() - ref).max() < 1e-5 * np.abs(ref).max()
    assert np.array_equal(rm.cpu().numpy(), np.abs(g.cpu().numpy()).max(axis=1))


@pytest.mark.parametrize("C", [128, 256])
def test_attention_on_split_kernels(ops, monkeypatch, C):
    """the f16x3 attention path against float64 autograd (same bar as the fp32 GEMM path, tests/test_gpu_kernels.py)
    and against that path; one image 2^12 times larger than the other"""
    rng = np.random.default_rng(C)
    B, S = 2, 1024
    q, k, v = (rng.standard_normal((B, S, C)) for _ in range(3))
    v[1] *= 4096.0
    do = rng.standard_normal((B, S, C))
    qt, kt, vt = (torch.tensor(a, requires_grad=True) for a in (q, k, v))
    o = torch.einsum("bqk,bkc->bqc", torch.softmax(torch.einsum("bqc,bkc->bqk", qt / math.sqrt(C), kt), -1), vt)
    o.backward(torch.tensor(do))
    rel = lambda a, r: float(np.abs(a - r).max() / np.abs(r).max())
    outs = {}
    for fast in (True, False):
        monkeypatch.setattr(ops, "ATTN_F16X3", fast)
        g = [dev(a).requires_grad_() for a in (q, k, v)]
        out = ops.attention(*g)
        out.backward(dev(do))
        outs[fast] = [out.detach().cpu().double().numpy()] + [a.grad.cpu().double().numpy() for a in g]
    refs = [o.detach().numpy(), qt.grad.numpy(), kt.grad.numpy(), vt.grad.numpy()]
    for b in range(B):                                   # per image: the small image is not drowned by the large one
        for got_fast, got_f32, r in zip(outs[True], outs[False], refs):
            assert rel(got_fast[b], r[b]) < 1e-5
            assert rel(got_fast[b], r[b]) < 2.0 * rel(got_f32[b], r[b]) + 2e-6
    # inference: no planes are produced, same output
    monkeypatch.setattr(ops, "ATTN_F16X3", True)
    with torch.no_grad():
        o2 = ops.attention(dev(q), dev(k), dev(v)).cpu().double().numpy()
    if C == 128:
        assert np.array_equal(o2, outs[True][0])
    else:       # C = 256 without a gradient: the fused forward kernel (round 3), another summation order
        for b in range(B):
            assert rel(o2[b], refs[0][b]) < 1e-5


def _heavy(rng, shape, outlier_axis0=True):
    """Student-t (nu = 2) values with one element per image (leading index) 1e4 x the bulk's scale"""
    x = rng.standard_t(2.0, size=shape)
    if outlier_axis0:
        flat = x.reshape(shape[0], -1)
        for b in range(shape[0]):
            flat[b, rng.integers(flat.shape[1])] = 1e4 * (1 if rng.random() < 0.5 else -1)
    return x


@pytest.mark.parametrize("B,C,N", [(128, 128, 128), (160, 128, 256), (256, 256, 128)])
def test_conv_v3_with_two_blocks_per_cu_is_exact_and_repeatable(ops, B, C, N):
    """The 2-blocks-per-CU convolution kernel at launch sizes where blocks really share a CU (more than 256 blocks, the
    second round runs with raised priority): equal to the one-block-per-CU kernel (same arithmetic, another schedule)
    to fp32 rounding of the epilogue, and bit-identical from launch to launch.  (Regression: accumulators of the last
    pixel tile read back before the last matrix instruction had retired -- only in blocks that ran without stalls.)"""
    torch.manual_seed(B + C + N)
    lib = ops.lib.load()
    x = torch.randn(B, 1024, C, device="cuda")
    w = torch.randn(3, 3, C, N, device="cuda") * 0.05
    bias, res = torch.randn(N, device="cuda"), torch.randn(B, 1024, N, device="cuda")
    try:
        lib.mulan_set_tuning(3, 2)
        ref, ref_planes = ops.conv3x3_raw(x, w, bias, None, res, planes=True)
    finally:
        lib.mulan_set_tuning(3, 0)
    first = None
    for _ in range(6):
        y, planes = ops.conv3x3_raw(x, w, bias, None, res, planes=True)
        assert torch.equal(planes, ref_planes)
        assert float((y - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
        first = y.clone() if first is None else first
        assert torch.equal(y, first)
        assert np.array_equal(y._absmax[0].cpu().numpy().view(np.float32).max(1),
                              y.abs().reshape(B, -1).amax(1).cpu().numpy())


@pytest.mark.parametrize("C,N", [(128, 128), (256, 128)])
def test_f16x3_heavy_tailed_operands(ops, monkeypatch, C, N):
    """Where the split scheme could break: the per-image / per-tensor power-of-two scale puts the absolute error floor
    at ~2^-38 of the operand's maximum, so what matters is how far the bulk sits below the maximum.  Activations with
    Student-t (nu = 2) tails plus a 1e4 x outlier per image, weights with one output channel 1e3 x the others:
    forward, input gradient and weight gradient, error per output element relative to sum |a||b| against float64, and
    not worse than 1.5 x the exact-fp32 MFMA kernel's on the same data."""
    rng = np.random.default_rng(C + N)
    B = 2
    x = _heavy(rng, (B, 32, 32, C))
    w = rng.standard_normal((3, 3, C, N)) / math.sqrt(9 * C)
    w[..., 5] *= 1e3
    dy = _heavy(rng, (B, 32, 32, N))
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    x, w, dy = r32(x), r32(w), r32(dy)
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    yt = tr.conv3x3(xt, {"kernel": wt})
    yt.backward(torch.tensor(dy))
    refs = dict(fwd=yt.detach().numpy(), dgrad=xt.grad.numpy(), wgrad=wt.grad.numpy())
    with torch.no_grad():
        ax, aw, ady = torch.tensor(np.abs(x), requires_grad=True), torch.tensor(np.abs(w), requires_grad=True), np.abs(dy)
    ya = tr.conv3x3(ax, {"kernel": aw})
    ya.backward(torch.tensor(ady))
    mags = dict(fwd=ya.detach().numpy(), dgrad=ax.grad.numpy(), wgrad=aw.grad.numpy())

    def run():
        xd, wd, dyd = dev(x).view(B, 1024, C), dev(w), dev(dy).view(B, 1024, N)
        return dict(fwd=ops.conv3x3_raw(xd, wd).cpu().double().numpy().reshape(B, 32, 32, N),
                    dgrad=ops.conv3x3_dgrad_raw(dyd, wd).cpu().double().numpy().reshape(B, 32, 32, C),
                    wgrad=ops.conv3x3_wgrad_raw(xd, dyd).cpu().double().numpy())
    got = run()
    monkeypatch.setattr(ops, "CONV_MODE", "f32")
    got32 = run()
    for k in ("fwd", "dgrad", "wgrad"):
        e16 = float((np.abs(got[k] - refs[k]) / mags[k]).max())
        e32 = float((np.abs(got32[k] - refs[k]) / mags[k]).max())
        # (with these tails the exact-fp32 MFMA kernel itself reaches ~4e-6 of sum |a||b|: long fp32 accumulation)
        assert e32 < 1e-5 and e16 < 1e-5 and e16 < 1.5 * e32 + 2e-7, (k, e16, e32)


def test_f16x3_plane_fed_wgrad_heavy_tails(ops):
    """the production weight-gradient path (planes written by the forward / input-gradient kernels, per-image scales,
    accumulator rescaling between images) on the same heavy-tailed data, images of very different magnitude"""
    rng = np.random.default_rng(77)
    B, C, N = 3, 128, 128
    x = _heavy(rng, (B, 32, 32, C)) * np.array([1.0, 1e-3, 50.0])[:, None, None, None]
    dy = _heavy(rng, (B, 32, 32, N)) * np.array([1e2, 1.0, 1e-2])[:, None, None, None]
    w = rng.standard_normal((3, 3, C, N)) / math.sqrt(9 * C)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    x, dy, w = r32(x), r32(dy), r32(w)
    wt = torch.tensor(w, requires_grad=True)
    tr.conv3x3(torch.tensor(x), {"kernel": wt}).backward(torch.tensor(dy))
    aw = torch.tensor(np.abs(w), requires_grad=True)
    tr.conv3x3(torch.tensor(np.abs(x)), {"kernel": aw}).backward(torch.tensor(np.abs(dy)))
    xd, dyd, wd = dev(x).view(B, 1024, C), dev(dy).view(B, 1024, N), dev(w)
    xmax, dymax = ops.absmax_rows(xd), ops.absmax_rows(dyd)
    _, xs = ops.conv3x3_raw(xd, wd, xmax=xmax, planes=True)
    _, dys = ops.conv3x3_dgrad_raw(dyd, wd, dymax=dymax, planes=True)
    dw = ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, B, C, N).cpu().double().numpy()
    err = float((np.abs(dw - wt.grad.numpy()) / aw.grad.numpy()).max())
    assert err < 1e-5, err


@pytest.mark.parametrize("C", [128, 256])
def test_fused_attention_forced_rescales_and_ranges(ops, monkeypatch, C):
    """The fused attention kernels (attention_f16x3.hip) where the online softmax could go wrong: keys that beat the
    running maximum late in the sweep (one query/key pair with a score far above the rest in tile 21, another in the
    last tile), a query whose scores are all very negative, images of very different magnitude; forward and all three
    gradients against float64 autograd, per image, and against the unfused split-operand path.  C = 256 (the ImageNet-32
    width, ldm/configs/imagenet32.py sm_n_embd): the backward kernels with their output channels split over blocks
    (dq, dk) and dv on its own."""
    rng = np.random.default_rng(9)
    B, S = 2, 1024
    q, k, v = (rng.standard_normal((B, S, C)) for _ in range(3))
    k[0, 700] = 6.0 * q[0, 5]                  # query 5: the maximum jumps by ~60 at row tile 21
    k[0, 1023] = 9.0 * q[0, 77]                # query 77: ... and in the very last tile
    q[1, 300] *= 25.0                          # a query with a wide score range
    k[1, 0] = -4.0 * q[1, 301]                 # one strongly negative score in the first tile
    v[1] *= 512.0
    do = rng.standard_normal((B, S, C))
    do[0] *= 1e-3
    qt, kt, vt = (torch.tensor(a, requires_grad=True) for a in (q, k, v))
    o = torch.einsum("bqk,bkc->bqc", torch.softmax(torch.einsum("bqc,bkc->bqk", qt / math.sqrt(C), kt), -1), vt)
    o.backward(torch.tensor(do))
    refs = [o.detach().numpy(), qt.grad.numpy(), kt.grad.numpy(), vt.grad.numpy()]
    rel = lambda a, r: float(np.abs(a - r).max() / np.abs(r).max())
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "ATTN_FUSED", fused)
        g = [dev(a).requires_grad_() for a in (q, k, v)]
        out = ops.attention(*g)
        assert (type(out.grad_fn).__name__ == "FusedAttentionFnBackward") == fused
        out.backward(dev(do))
        outs[fused] = [out.detach().cpu().double().numpy()] + [a.grad.cpu().double().numpy() for a in g]
    for b in range(B):
        for name, got, old, r in zip(("o", "dq", "dk", "dv"), outs[True], outs[False], refs):
            assert rel(got[b], r[b]) < 1e-5, (name, b, rel(got[b], r[b]))
            assert rel(got[b], r[b]) < 2.0 * rel(old[b], r[b]) + 2e-6, (name, b, rel(got[b], r[b]), rel(old[b], r[b]))
    # the rows that took the rescale branch, on their own (their dq is ~0: the softmax is saturated on one key)
    for (b, i) in ((0, 5), (0, 77), (1, 300)):
        assert rel(outs[True][0][b, i], refs[0][b, i]) < 1e-5


def test_attention_dual_pack_equals_the_generic_packs(ops):
    """mulan_attention_pack_f16x3 (both operand layouts in one pass) is bit-identical to the two calls of
    mulan_linear_pack_f16x3_batched it replaces"""
    torch.manual_seed(3)
    x = torch.randn(3, 1024, 128, device="cuda") * torch.tensor([1.0, 300.0, 1e-3], device="cuda")[:, None, None]
    m = ops.absmax_rows(x)
    xt, xn = ops._attn_packs(x, m)
    assert torch.equal(xt, ops._pack_batched(x, True, m)) and torch.equal(xn, ops._pack_batched(x, False, m))
    only_t, none = ops._attn_packs(x, m, True, False)
    assert none is None and torch.equal(only_t, xt)


@pytest.mark.parametrize("B,C1,C2,N,keep,with_res", [(3, 128, 0, 128, 1.0, False), (2, 128, 128, 128, 1.0, True),
                                                     (5, 128, 0, 256, 0.9, True), (2, 256, 256, 256, 0.9, False)])
def test_gn_conv_plane_hand_over_matches_the_two_op_path(ops, monkeypatch, B, C1, C2, N, keep, with_res):
    """ops.gn_conv3x3: GroupNorm -> 3x3 convolution as one node, the normalised tensor handed over as split fp16 planes
    under an a-priori bound (mulan_groupnorm_fwd_planes -> mulan_conv3x3_fwd_f16x3_planes_in -> plane-fed weight
    gradient) against the two-op path with an fp32 tensor in between (per-image scale from the true maximum): output
    and every gradient (inputs incl. the skip aliases, gamma, beta, kernel, bias, FiLM bias, residual) agree to the
    split's rounding; the bound really bounds; no plane store leaves the convolution."""
    torch.manual_seed(B + C1 + C2 + N)
    Ct = C1 + C2
    mk = lambda *s, scale=1.0: (torch.randn(*s, device="cuda") * scale).requires_grad_(True)
    x1 = mk(B, 1024, C1, scale=3.0)
    x2 = mk(B, 1024, C2) if C2 else None
    gamma, beta = mk(Ct), mk(Ct, scale=0.3)
    w, bias, cb = mk(3, 3, Ct, N, scale=0.03), mk(N), mk(B, N)
    res = mk(B, 1024, N) if with_res else None
    gy, g1 = torch.randn(B, 1024, N, device="cuda"), torch.randn(B, 1024, C1, device="cuda")
    g2 = torch.randn(B, 1024, C2, device="cuda") if C2 else None
    leaves = [t for t in (x1, x2, gamma, beta, w, bias, cb, res) if t is not None]
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    def run(planes):
        monkeypatch.setattr(ops, "GN_CONV_PLANES", planes)
        for t in leaves:
            t.grad = None
        names.clear()
        y, s1, s2 = ops.gn_conv3x3(x1, x2, gamma, beta, w, bias, cbias=cb, res=res, act=True, keep=keep, seed=7, offset=32,
                                   skip=True)
        loss = (y * gy).sum() + (s1 * g1).sum() + ((s2 * g2).sum() if C2 else 0)
        loss.backward()
        return [y.detach().clone()] + [t.grad.clone() for t in leaves], list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    assert "mulan_groupnorm_fwd_planes" in got_names and "mulan_conv3x3_fwd_f16x3_planes_in_stats" in got_names
    assert "mulan_groupnorm_fwd_dyn" not in got_names and "mulan_groupnorm_fwd_planes" not in ref_names
    assert got_names.count("mulan_conv3x3_fwd_f16x3_alone") == 1    # only the input-gradient launch is left on it
    labels = ["y"] + [n for n, t in zip(("x1", "x2", "gamma", "beta", "w", "bias", "cb", "res"),
                                         (x1, x2, gamma, beta, w, bias, cb, res)) if t is not None]
    for a, r, nm in zip(got, ref, labels):
        assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-30, (nm, float((a - r).abs().max()), float(r.abs().max()))


@pytest.mark.parametrize("B,C1,C2,N,act,cbdim,with_res", [(3, 128, 0, 128, True, 2, False), (2, 128, 128, 128, True, 2, True),
                                                          (5, 128, 0, 256, False, None, True), (2, 256, 256, 256, True, 2, False),
                                                          (2, 256, 0, 256, True, 3, False), (130, 128, 128, 128, True, 2, True)])
def test_gn_normalised_inside_the_convolution_is_bit_identical(ops, monkeypatch, B, C1, C2, N, act, cbdim, with_res):
    """mulan_groupnorm_stats + mulan_conv3x3_fwd_f16x3_gn_in (the GroupNorm normalised, activated and split inside the
    convolution's patch fill: forward-only paths) against mulan_groupnorm_fwd_planes + ..._planes_in: the same bits --
    output, output maxima, and the input gradient a likelihood evaluator takes through the node (kernel without
    gradient); with MULAN_GN_FILL_TRAIN also the train-step form (the convolution stores the planes for its weight
    gradient): every gradient identical.  Heavy-tailed input with an offset, so that the clamp, the zero padding of the
    NORMALISED tensor and (x - mean) are all exercised; B = 130 runs two blocks per CU."""
    torch.manual_seed(B + C1 + C2 + N)
    Ct = C1 + C2
    x1 = (torch.randn(B, 1024, C1, device="cuda") ** 3 + 0.7).requires_grad_(True)
    x2 = (torch.randn(B, 1024, C2, device="cuda") * 2 - 0.4).requires_grad_(True) if C2 else None
    gamma, beta = torch.randn(Ct, device="cuda").requires_grad_(True), (torch.randn(Ct, device="cuda") * 0.3).requires_grad_(True)
    w = (torch.randn(3, 3, Ct, N, device="cuda") * 0.03).requires_grad_(True)
    bias = torch.randn(N, device="cuda").requires_grad_(True)
    cb = None if cbdim is None else (torch.randn(B, N, device="cuda") if cbdim == 2 else torch.randn(B, 1024, N, device="cuda"))
    res = torch.randn(B, 1024, N, device="cuda") if with_res else None
    gy = torch.randn(B, 1024, N, device="cuda")
    leaves = [t for t in (x1, x2, gamma, beta, w, bias) if t is not None]
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    def run(fill, train):
        monkeypatch.setattr(ops, "GN_FILL", fill)
        monkeypatch.setattr(ops, "GN_FILL_MAX_N", 512)           # (the product path keeps N = 256 on the plane hand-over: speed only)
        monkeypatch.setattr(ops, "GN_FILL_STATS", False)         # (statistics from mulan_groupnorm_stats: the bit-identical form)
        monkeypatch.setattr(ops, "GN_FILL_TRAIN", fill and train)
        for t in leaves:
            t.grad = None
            t.requires_grad_(train or t is x1 or t is x2)
        names.clear()
        y = ops.gn_conv3x3(x1, x2, gamma, beta, w, bias, cbias=cb, res=res, act=act, keep=1.0)
        ymax = ops.cached_absmax(y).clone()
        (y * gy).sum().backward()
        return [y.detach().clone(), ymax] + [t.grad.clone() for t in leaves if t.grad is not None], list(names)

    for train in (False, True):
        ref, ref_names = run(False, train)
        got, got_names = run(True, train)
        assert "mulan_groupnorm_fwd_planes" in ref_names and "mulan_groupnorm_stats" not in ref_names
        assert "mulan_groupnorm_stats" in got_names and "mulan_conv3x3_fwd_f16x3_gn_in" in got_names
        assert "mulan_groupnorm_fwd_planes" not in got_names and "mulan_conv3x3_fwd_f16x3_planes_in_stats" not in got_names
        assert len(ref) == len(got) == (2 + len(leaves) if train else 2 + (2 if C2 else 1))
        for i, (a_, r_) in enumerate(zip(got, ref)):
            assert torch.equal(a_, r_), (train, i, float((a_.float() - r_.float()).abs().max()))


@pytest.mark.parametrize("B,E", [(3, 128), (130, 128), (2, 256)])
def test_gn_statistics_handed_from_convolution_to_convolution(ops, monkeypatch, B, E):
    """A forward-only chain conv -> GroupNorm -> conv -> ... (ResnetBlocks under the evaluators / the sampler): each
    GroupNorm-fed convolution leaves the partial sums of its output, the next one forms mean / rstd / bound from them in
    its prologue -- one mulan_groupnorm_stats launch for the chain's first tensor, none after it, concat inputs included.
    Same formulas in another summation order: every tensor of the chain, the saved statistics and the input gradient a
    likelihood evaluator takes through it agree with the statistics-kernel route to fp32 rounding."""
    torch.manual_seed(B + E)
    x0 = (torch.randn(B, 1024, E, device="cuda") * 2 + 0.3).requires_grad_(True)
    skip = torch.randn(B, 1024, E, device="cuda")
    mk = lambda *s_, sc=1.0: torch.randn(*s_, device="cuda") * sc
    layers = [(mk(E), mk(E, sc=0.3), mk(3, 3, E, E, sc=0.03), mk(E), mk(B, E)) for _ in range(3)]
    g2, b2, w2 = mk(2 * E), mk(2 * E, sc=0.3), mk(3, 3, 2 * E, E, sc=0.02)
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])
    monkeypatch.setattr(ops, "GN_FILL_MAX_N", 512)

    def run(hand_over):
        monkeypatch.setattr(ops, "GN_FILL_STATS", hand_over)
        names.clear()
        x0.grad = None
        outs = []
        h = x0
        for i, (g, b_, w, bias, cb) in enumerate(layers):
            h = ops.gn_conv3x3(h, None, g, b_, w, bias, cbias=cb if i % 2 == 0 else None, res=h if i % 2 else None)
            outs.append(h)
        a, b2_ = ops.tee(h)                                        # block output with two consumers
        sk, _ = ops.tee(outs[0])
        h = ops.gn_conv3x3(a, sk, g2, b2, w2)                      # up block: GroupNorm over the concat [h | skip]
        outs.append(h)
        h2 = ops.gn_conv3x3(b2_, skip, g2, b2, w2)                 # a skip tensor nobody left statistics on: falls back
        outs.append(h2)
        (h * 0.5 + h2).sum().backward()
        return [o.detach().clone() for o in outs] + [x0.grad.clone()], list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    assert ref_names.count("mulan_groupnorm_stats") == 5 and got_names.count("mulan_groupnorm_stats") == 2
    for i, (a_, r_) in enumerate(zip(got, ref)):
        assert float((a_ - r_).abs().max()) <= 3e-6 * float(r_.abs().max()), (i, float((a_ - r_).abs().max()), float(r_.abs().max()))


@pytest.mark.parametrize("vdm_type,unet_type", [("mulan_velocity", "vdm"), ("mulan_epsilon", "ldm")])
def test_forward_only_model_uses_the_fill_path_with_identical_losses(ops, monkeypatch, vdm_type, unet_type):
    """MuLAN forward under no_grad (evaluators / sampler): the ResnetBlock GroupNorms are normalised inside their
    convolutions (no mulan_groupnorm_fwd_planes launch is left for the dropout-free eval pass, no plane tensor written);
    the three losses are bit for bit those of the plane hand-over."""
    from mulan_amd import model as M
    from mulan_amd.rng import PRNGKey
    cfg = M.VDMConfig(vocab_size=256, sample_softmax=False, antithetic_time_sampling=True, with_fourier_features=True,
                      with_attention=False, gamma_type='poly_fixedend', gamma_min=-13.3, gamma_max=5.0, sm_n_timesteps=0,
                      sm_n_embd=128, sm_n_layer=2, sm_pdrop=0.1, forward_n_layer=1, latent_size=50, latent_k=15,
                      encoder='unet', latent_type='topk', z_conditioning=True, reparam_type='true', unet_type=unet_type,
                      condition='input')
    vdm = M.make_vdm(vdm_type, cfg)
    params = M.tree_map(lambda t: t.cuda(), vdm.init(PRNGKey(3)))
    for _, leaf in M.tree_leaves(params):
        leaf.normal_(0.0, 0.05)          # (zero-initialised layers would switch whole branches off)
    x = torch.randint(0, 256, (6, 32, 32, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    def run(fill, hand_over=False):
        monkeypatch.setattr(ops, "GN_FILL", fill)
        monkeypatch.setattr(ops, "GN_FILL_STATS", hand_over)
        monkeypatch.setattr(ops, "GN_FWD_STREAM", False)   # (the reference is the slab kernel: mulan_groupnorm_stats' summation order)
        names.clear()
        with torch.no_grad():
            out = vdm.apply(params, x, None, None, step=0, rngs={"sample": PRNGKey(5)}, deterministic=True)
        return [out.loss_recon.clone(), out.loss_klz.clone(), out.loss_diff.clone()], list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    # the shipped form: statistics handed from convolution to convolution (fewer launches still; losses to fp32 rounding)
    got2, got2_names = run(True, True)
    assert got2_names.count("mulan_groupnorm_stats") < got_names.count("mulan_groupnorm_stats") // 2
    for a_, r_ in zip(got2, ref):
        assert float((a_ - r_).abs().max()) <= 2e-6 * float(r_.abs().max())
    assert "mulan_conv3x3_fwd_f16x3_gn_in" not in ref_names and ref_names.count("mulan_groupnorm_fwd_planes") > 0
    assert got_names.count("mulan_conv3x3_fwd_f16x3_gn_in") == ref_names.count("mulan_groupnorm_fwd_planes")
    assert "mulan_groupnorm_fwd_planes" not in got_names
    for a_, r_ in zip(got, ref):
        assert torch.equal(a_, r_)


def test_gn_planes_bound_and_precision(ops):
    """mulan_groupnorm_fwd_planes: the planes decode (hi + lo) / scale to the fp32 output of the ordinary kernel within
    the split's 2^-22, the scale comes from a bound that really bounds (heavy-tailed input: |xhat| up to ~40 of the
    possible 64), and the plane-fed convolution is as close to float64 as the fp32-input one."""
    torch.manual_seed(4)
    B, C, N = 3, 128, 128
    x = torch.randn(B, 1024, C, device="cuda")
    x[0, 17, 5] = 400.0                                           # one element far out: a large |xhat| in its group
    x[1] *= 1e-3
    gamma, beta = torch.randn(C, device="cuda") * 2, torch.randn(C, device="cuda")
    y = ops.group_norm(x, None, gamma, beta, act=True).detach()
    ys = torch.empty(B * 1024 * C * 4, device="cuda", dtype=torch.uint8)
    bound = torch.empty(B, 16, device="cuda", dtype=torch.int32)
    mean, rstd = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
    ops.call("mulan_groupnorm_fwd_planes", ops.ptr(x), None, C, 0, ops.ptr(gamma), ops.ptr(beta), ops.ptr(ys), ops.ptr(mean),
             ops.ptr(rstd), B, 1024, 32, 1e-6, 1, 1.0, 0, 0, None, ops.ptr(bound), ops.stream())
    bnd = bound.cpu().numpy().view(np.float32).max(1)
    want = (np.sqrt(1024 * 4) * float(gamma.abs().max()) + float(beta.abs().max()))
    assert np.allclose(bnd, want, rtol=1e-6) and float(y.abs().max()) <= want
    e = int(np.frexp(bnd[0])[1]) - 1 + 127                        # biased exponent of the bound -> scale 2^(140 - e)
    planes = ys.view(torch.float16).view(B, C // 16, 1024, 2, 16).float()
    dec = (planes[:, :, :, 0] + planes[:, :, :, 1]).permute(0, 2, 1, 3).reshape(B, 1024, C) * 2.0 ** (e - 140)
    assert float((dec - y).abs().max()) <= 2.0 ** -21 * float(y.abs().max())
    w = torch.randn(3, 3, C, N, device="cuda") * 0.05
    out = torch.empty(B, 1024, N, device="cuda")
    wp, wmax = ops._pack_weights(w, C, N, 0)
    ops.call("mulan_conv3x3_fwd_f16x3_planes_in", ops.ptr(ys), ops.ptr(bound), ops.ptr(wp), ops.ptr(wmax), None, None, 0,
             None, ops.ptr(out), None, B, 32, 32, C, N, ops.stream())
    ref = torch.nn.functional.conv2d(y.double().view(B, 32, 32, C).permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1),
                                     padding=1).permute(0, 2, 3, 1).reshape(B, 1024, N)
    two_op = ops.conv3x3_raw(y, w)
    mag = float(ref.abs().max())
    e_planes, e_two = float((out.double() - ref).abs().max()) / mag, float((two_op.double() - ref).abs().max()) / mag
    assert e_planes <= 2.0 * e_two + 2e-7, (e_planes, e_two)


def test_gn_backward_planes_bound_and_precision(ops):
    """mulan_groupnorm_bwd_fused_planes (round 3): the GroupNorm backward writes dx as the split fp16 planes of the
    convolution in front instead of as fp32.  The planes decode (hi + lo) / scale to the fp32 dx of the ordinary kernel
    within the split's 2^-21 of the image's maximum, the scale comes from an a-priori bound that really bounds (one
    outlier pixel: |xhat| ~ 40 of the possible 64; an image 1000 x smaller; a heavy-tailed incoming gradient), dgamma /
    dbeta / the channel sums are those of the fp32 kernel bit for bit, and the plane-fed input-gradient convolution is as
    close to float64 as the fp32-input one."""
    torch.manual_seed(9)
    B, C, N = 3, 128, 128
    for keep in (1.0, 0.9):
        x = torch.randn(B, 1024, C, device="cuda")
        x[0, 17, 5] = 300.0
        x[1] *= 1e-3
        dy = torch.randn(B, 1024, C, device="cuda")
        dy[2] = dy[2] ** 3 * 1e-4                                      # heavy tails, small magnitude
        dy[0, 100, 7] = 50.0
        gamma, beta = torch.randn(C, device="cuda") * 2, torch.randn(C, device="cuda")
        mean, rstd = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
        y = torch.empty_like(x)
        ops.call("mulan_groupnorm_fwd_dyn", ops.ptr(x), None, C, 0, ops.ptr(gamma), ops.ptr(beta), ops.ptr(y), ops.ptr(mean),
                 ops.ptr(rstd), B, 1024, 32, 1e-6, 1, keep, 11, 64, None, None, ops.stream())
        dymax = ops.absmax_rows(dy)
        tick = torch.zeros(16, device="cuda", dtype=torch.int32)
        kb = torch.zeros(B * (C // 32) * 1024, device="cuda", dtype=torch.int32)

        def run(planes):
            dx = torch.empty_like(x)
            dxp = torch.empty(B * 1024 * C * 4, device="cuda", dtype=torch.uint8)
            m = torch.empty(B, 16, device="cuda", dtype=torch.int32)
            parts = torch.empty(2, B, C, device="cuda")
            csum = torch.empty(B, C, device="cuda")
            dg, db, sink = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
            if planes:
                ops.call("mulan_groupnorm_bwd_fused_planes", ops.ptr(dy), ops.ptr(dymax), ops.ptr(x), C, ops.ptr(gamma),
                         ops.ptr(beta), ops.ptr(mean), ops.ptr(rstd), ops.ptr(dxp), ops.ptr(parts[0]), ops.ptr(parts[1]), B,
                         1024, 32, 1, keep, 11, 64, None, ops.ptr(m), ops.ptr(csum), ops.ptr(dg), ops.ptr(db), ops.ptr(sink),
                         None, ops.ptr(tick), ops.ptr(kb) if planes == "kept" else None, ops.stream())
            else:
                ops.call("mulan_groupnorm_bwd_fused", ops.ptr(dy), ops.ptr(x), None, C, 0, ops.ptr(gamma), ops.ptr(beta),
                         ops.ptr(mean), ops.ptr(rstd), ops.ptr(dx), None, ops.ptr(parts[0]), ops.ptr(parts[1]), B, 1024, 32, 1,
                         keep, 11, 64, None, ops.ptr(m), None, None, None, None, ops.ptr(csum), ops.ptr(dg), ops.ptr(db),
                         ops.ptr(sink), None, ops.ptr(tick), ops.stream())
            return dx, dxp, m, csum, dg, db, sink

        dx, _, m_ref, csum_ref, dg_ref, db_ref, sink_ref = run(False)
        _, dxp, bound, csum, dg, db, sink = run(True)
        if keep < 1.0:
            # the keep-bits as the forward kernel stores them (mulan_groupnorm_fwd_planes_keepbits) instead of the re-draw:
            # the same bits, hence the same planes and sums; and the planes that kernel writes equal the plain entry point's
            ys0, ys1 = (torch.empty(B * 1024 * C * 4, device="cuda", dtype=torch.uint8) for _ in range(2))
            bd = torch.empty(B, 16, device="cuda", dtype=torch.int32)
            mm, rr = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
            ops.call("mulan_groupnorm_fwd_planes", ops.ptr(x), None, C, 0, ops.ptr(gamma), ops.ptr(beta), ops.ptr(ys0), ops.ptr(mm),
                     ops.ptr(rr), B, 1024, 32, 1e-6, 1, keep, 11, 64, None, ops.ptr(bd), ops.stream())
            ops.call("mulan_groupnorm_fwd_planes_keepbits", ops.ptr(x), None, C, 0, ops.ptr(gamma), ops.ptr(beta), ops.ptr(ys1),
                     ops.ptr(mm), ops.ptr(rr), B, 1024, 32, 1e-6, 1, keep, 11, 64, None, ops.ptr(bd), ops.ptr(kb), ops.stream())
            assert torch.equal(ys0, ys1)
            frac = float(sum(bin(int(v) & 0xffffffff).count("1") for v in kb[:4096].cpu().tolist())) / (4096 * 32)
            assert abs(frac - keep) < 0.01, frac
            _, dxp_k, bound_k, csum_k, dg_k, db_k, sink_k = run("kept")
            for a_, r_ in ((dxp_k, dxp), (bound_k, bound), (csum_k, csum), (dg_k, dg), (db_k, db), (sink_k, sink)):
                assert torch.equal(a_, r_)
        assert int(tick.abs().sum()) == 0
        for a, r in ((csum, csum_ref), (dg, dg_ref), (db, db_ref), (sink, sink_ref)):
            assert torch.equal(a, r)
        bnd = bound.cpu().numpy().view(np.float32)
        assert np.all(bnd[:, :4] == bnd[:, :1]) and np.all(bnd[:, 4:] == 0)       # the same bound from every slab's block
        true_max = dx.abs().amax(dim=(1, 2)).cpu().numpy()
        assert np.all(bnd[:, 0] >= true_max), (bnd[:, 0], true_max)
        print("GroupNorm backward planes: bound / true maximum per image =", bnd[:, 0] / true_max)
        assert np.all(bnd[:, 0] <= 2.0 ** 12 * true_max)                         # loose by a few hundred at most
        planes = dxp.view(torch.float16).view(B, C // 16, 1024, 2, 16).float()
        for b in range(B):
            e = int(np.frexp(bnd[b, 0])[1]) - 1 + 127                            # biased exponent -> scale 2^(140 - e)
            dec = (planes[b, :, :, 0] + planes[b, :, :, 1]).permute(1, 0, 2).reshape(1024, C) * 2.0 ** (e - 140)
            # two fp16 pieces of v * scale: relative 2^-22 of the element, absolute floor 2^-25 in scaled units
            tol = 2.0 ** -21 * dx[b].abs() + 2.0 ** (-25 + e - 140)
            assert bool(((dec - dx[b]).abs() <= tol).all()), (b, float((dec - dx[b]).abs().max()), float(dx[b].abs().max()))
        # the input-gradient convolution fed with these planes against the fp32-input launch, both against float64
        w = torch.randn(3, 3, N, C, device="cuda") * 0.05                        # conv N -> C; its dgrad maps C -> N
        got = ops.conv3x3_dgrad_planes_raw(dxp, bound, w)
        two = ops.conv3x3_dgrad_raw(dx, w)
        ref = torch.nn.functional.conv_transpose2d(dx.double().view(B, 32, 32, C).permute(0, 3, 1, 2),
                                                   w.double().permute(2, 3, 0, 1), padding=1).permute(0, 2, 3, 1).reshape(B, 1024, N)
        for b in range(B):
            mag = float(ref[b].abs().max())
            e_p, e_t = float((got[b].double() - ref[b]).abs().max()) / mag, float((two[b].double() - ref[b]).abs().max()) / mag
            assert e_p <= 2.0 * e_t + 2e-7, (keep, b, e_p, e_t)


@pytest.mark.parametrize("B,C,E,keep,shortcut", [(3, 128, 128, 0.9, False), (2, 256, 128, 0.9, True), (2, 256, 256, 1.0, False)])
def test_grad_planes_hand_over_matches_the_fp32_path(ops, monkeypatch, B, C, E, keep, shortcut):
    """A ResnetBlock-shaped pair of ops.gn_conv3x3 nodes (norm1 + swish -> conv1 + FiLM bias; norm2 + swish + dropout ->
    conv2 + residual): with x1_grad_planes the gradient norm2's backward hands to conv1 exists only as split planes
    (mulan_groupnorm_bwd_fused_planes -> plane-fed input-gradient convolution + plane-fed weight gradient); every
    gradient (input, both GroupNorms, both kernels and biases, FiLM bias, shortcut) agrees with the fp32 hand-over to
    the split's rounding, the stand-in tensor autograd carries is never read, and one fp32-input convolution launch per
    block is gone."""
    torch.manual_seed(B + C + E)
    mk = lambda *s, scale=1.0: (torch.randn(*s, device="cuda") * scale).requires_grad_(True)
    x = mk(B, 1024, C, scale=2.0)
    g1, b1, g2, b2 = mk(C), mk(C, scale=0.3), mk(E), mk(E, scale=0.3)
    w1, c1b, w2, c2b, cb = mk(3, 3, C, E, scale=0.03), mk(E), mk(3, 3, E, E, scale=0.03), mk(E), mk(B, E)
    wn = mk(C, E, scale=0.05) if shortcut else None
    gy = torch.randn(B, 1024, E, device="cuda")
    leaves = [t for t in (x, g1, b1, g2, b2, w1, c1b, w2, c2b, cb, wn) if t is not None]
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    def run(planes):
        monkeypatch.setattr(ops, "GRAD_PLANES", planes)
        for t in leaves:
            t.grad = None
        names.clear()
        h, s1, _ = ops.gn_conv3x3(x, None, g1, b1, w1, c1b, cbias=cb, act=True, skip=True)
        res = ops.linear(s1, wn, None) if shortcut else (s1 if C == E else None)
        y = ops.gn_conv3x3(h, None, g2, b2, w2, c2b, res=res, act=True, keep=keep, seed=5, offset=1 << 34, x1_grad_planes=True)
        (y * gy).sum().backward()
        return [y.detach().clone()] + [t.grad.clone() for t in leaves], list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    assert "mulan_groupnorm_bwd_fused_planes" in got_names and "mulan_groupnorm_bwd_fused_planes" not in ref_names
    assert got_names.count("mulan_conv3x3_fwd_f16x3_alone") == ref_names.count("mulan_conv3x3_fwd_f16x3_alone") - 1
    assert got_names.count("mulan_conv3x3_fwd_f16x3_planes_in_stats") == ref_names.count("mulan_conv3x3_fwd_f16x3_planes_in_stats") + 1
    assert torch.equal(got[0], ref[0])
    labels = ["y"] + [n for n, t in zip(("x", "g1", "b1", "g2", "b2", "w1", "c1b", "w2", "c2b", "cb", "wn"),
                                         (x, g1, b1, g2, b2, w1, c1b, w2, c2b, cb, wn)) if t is not None]
    for a, r, nm in zip(got, ref, labels):
        assert bool(torch.isfinite(a).all()), nm
        assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-30, (nm, float((a - r).abs().max()), float(r.abs().max()))


@pytest.mark.parametrize("B", [3, 1000])
def test_fused_attention_forward_at_the_imagenet32_width(ops, B):
    """C = 256 (ldm/configs/imagenet32.py:70): the fused forward kernel (no [B, 1024, 1024] tensor) against float64 and
    against the unfused path; ops.attention picks it wherever no gradient is taken (evaluators, sampler), also at the
    dense evaluator's batch of 1000 copies, where the unfused f16x3 products do not apply (B S^2 elements >= 2^31) and
    the score matrix alone would be 4 GiB."""
    torch.manual_seed(B)
    C = 256
    g = torch.Generator(device="cuda").manual_seed(B)
    q = torch.randn(B, 1024, C, device="cuda", generator=g) * 1.5
    k = torch.randn(B, 1024, C, device="cuda", generator=g) * 1.5
    v = torch.randn(B, 1024, C, device="cuda", generator=g)
    k[0, 700] = q[0, 3] * 2.5                        # one key far above the running maximum of its row, late in the sweep
    with torch.no_grad():
        base = torch.cuda.max_memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        o = ops.attention(q, k, v)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - torch.cuda.memory_allocated()
    assert o.shape == q.shape and bool(torch.isfinite(o).all())
    assert peak < 6 * q.numel() * 4 + (64 << 20), peak          # packs + o, no S / P matrices
    chk = range(B) if B <= 4 else (0, 1, B // 2, B - 1)
    for b in chk:
        s = (q[b].double() @ k[b].double().T) / np.sqrt(C)
        ref = torch.softmax(s, dim=-1) @ v[b].double()
        err = float((o[b].double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (b, err)
    if B <= 4:
        qq = q.clone().requires_grad_(True)          # with a gradient to take: the same kernels (round 4), the same values
        torch.cuda.reset_peak_memory_stats()
        oo = ops.attention(qq, k, v)
        assert type(oo.grad_fn).__name__ == "FusedAttentionFnBackward"
        assert torch.equal(oo.detach(), o)
        oo.backward(torch.randn_like(oo))
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - torch.cuda.memory_allocated()
        assert peak < 16 * q.numel() * 4 + (64 << 20), peak     # packs of q, k, v, do in both layouts, o, dq, dk, dv: no S / P
