"""Tile heights of the two-blocks-per-CU convolution kernel (round 5: 8, 4 or 2 image rows per block, chosen per launch so
that launches of fewer than 128 images still put two blocks on every CU -- conv3x3_f16x3_v3.hip, mulan_conv3x3_f16x3_tile_rows).
The contraction order of an output element does not depend on the tile, so every instantiation must give the SAME numbers
at every tile height: forward (fp32 input, plane-fed, GroupNorm-fed), input gradient, by-products (planes, maxima), and the
statistics hand-over must work between launches of different tile heights.  ldm/model_vdm.py:633-656 (the convolutions of
ResnetBlock) and their autodiff."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mulan_np as onp


@pytest.fixture()
def ops(monkeypatch):
    from mulan_amd import ops as _ops
    _ops.lib.load()
    monkeypatch.setattr(_ops, "CONV_MODE", "f16x3")
    yield _ops
    _ops.call("mulan_set_tuning", 23, 0)
    _ops.call("mulan_set_tuning", 27, 0)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).float().cuda()


def test_tile_rows_policy(ops):
    L = ops.lib.load()
    rows = lambda B, N, ymax=1: L.mulan_conv3x3_f16x3_tile_rows(B, 32, N, ymax)
    assert rows(128, 128) == 8 and rows(512, 128) == 8           # the headline batch keeps the 8-row tile
    assert rows(64, 128) == 8                                    # 64 images per GPU: one whole block per CU (measured best)
    assert rows(32, 128) == 4 and rows(16, 128) == 2 and rows(4, 128) == 2       # sampling batches
    assert rows(32, 256) == 8 and rows(16, 256) == 4             # two cout blocks double the grid
    assert rows(8, 256) == 4 and rows(8, 256, 0) == 2            # 16 row tiles x 2 cout blocks do not fit the 16 maxima entries


@pytest.mark.parametrize("B,C,N", [(2, 128, 128), (1, 256, 128), (1, 128, 256), (3, 32, 128), (2, 64, 128), (1, 192, 128)])
@pytest.mark.parametrize("rows,ks", [(4, 1), (2, 1), (8, 2), (4, 2), (2, 2)])
def test_short_tiles_exact_on_integers(ops, B, C, N, rows, ks):
    """fp32-input instantiation (+ bias, per-sample FiLM bias, residual) and its input gradient against the numpy oracle;
    ks = 2: as k-split blocks (two groups of four waves, each over half of the channels: tune[27] = 2)"""
    ops.call("mulan_set_tuning", 23, rows)
    ops.call("mulan_set_tuning", 27, 2 if ks == 2 else 1)
    rng = np.random.default_rng(B + C + N + rows)
    x = rng.integers(-3, 4, (B, 32, 32, C)).astype(np.float64)
    w = rng.integers(-2, 3, (3, 3, C, N)).astype(np.float64)
    bias, cb = rng.integers(-3, 4, N).astype(np.float64), rng.integers(-3, 4, (B, N)).astype(np.float64)
    res = rng.integers(-3, 4, (B, 32, 32, N)).astype(np.float64)
    ref = onp.conv3x3(x, w, bias) + cb[:, None, None, :] + res
    y = ops.conv3x3_raw(dev(x).view(B, 1024, C), dev(w), dev(bias), dev(cb), dev(res).view(B, 1024, N))
    assert np.array_equal(y.cpu().double().numpy().reshape(ref.shape), ref)
    if C % 128 == 0:
        dy = rng.integers(-3, 4, (B, 32, 32, N)).astype(np.float64)
        wt = np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2))        # flipped taps, channels swapped
        dx = ops.conv3x3_dgrad_raw(dev(dy).view(B, 1024, N), dev(w))
        assert np.array_equal(dx.cpu().double().numpy().reshape(B, 32, 32, C), onp.conv3x3(dy, wt, None))


@pytest.mark.parametrize("B,C,N", [(3, 128, 128), (2, 256, 128), (2, 128, 256)])
def test_every_tile_height_gives_the_same_numbers(ops, B, C, N):
    """random data: outputs, split planes and maxima of the fp32-input launch, and a training-mode GroupNorm -> convolution
    node (plane-fed forward + plane-fed input gradient + weight gradient) are bit-identical at 8, 4 and 2 rows per block"""
    torch.manual_seed(B * C + N)
    x = torch.randn(B, 1024, C, device="cuda") * 1.7
    w = torch.randn(3, 3, C, N, device="cuda") * 0.05
    bias, cb, res = torch.randn(N, device="cuda"), torch.randn(B, N, device="cuda"), torch.randn(B, 1024, N, device="cuda")
    gamma, beta = torch.randn(C, device="cuda") * 0.4 + 1, torch.randn(C, device="cuda") * 0.2
    L = ops.lib.load()

    def run(rows):
        ops.call("mulan_set_tuning", 23, rows)
        ops.call("mulan_set_tuning", 27, 1)              # (4-wave blocks at every height; the k-split blocks: below)
        y, xs = ops.conv3x3_raw(x, w, bias, cb, res, planes=True)
        ymax = y._absmax[0].view(B, 16).amax(1).clone()
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        z = ops.gn_conv3x3(xg, None, gamma, beta, wg, bias, cbias=cb, keep=0.9, seed=5, offset=64)
        (z * res).sum().backward()
        return y.clone(), xs.clone(), ymax, z.detach().clone(), xg.grad.clone(), wg.grad.clone()

    ref = run(8)
    for rows in (4, 2):
        got = run(rows)
        used = L.mulan_conv3x3_f16x3_tile_rows(B, 32, N, 1)
        assert used == (rows if (32 // rows) * (N // 128) <= 16 else 4), (rows, used)
        for i, (a, r) in enumerate(zip(got, ref)):
            assert torch.equal(a, r), (rows, i, float((a.float() - r.float()).abs().max()))
        assert torch.equal(got[2], ref[0].abs().amax((1, 2)).view(torch.int32))


def test_statistics_hand_over_between_tile_heights(ops, monkeypatch):
    """forward-only chain conv -> GroupNorm -> conv with the normalisation inside the convolutions' patch fill: the partial
    sums a launch with 8-row tiles leaves (4 per image) are consumed by a launch with 2-row tiles and the other way round
    (xstats_tiles); results equal the statistics-kernel route to fp32 rounding"""
    torch.manual_seed(1)
    B, E = 3, 128
    x0 = torch.randn(B, 1024, E, device="cuda") * 2 + 0.3
    mk = lambda *s_, sc=1.0: torch.randn(*s_, device="cuda") * sc
    layers = [(mk(E), mk(E, sc=0.3), mk(3, 3, E, E, sc=0.03), mk(E)) for _ in range(4)]
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    def run(hand_over, tiles):
        monkeypatch.setattr(ops, "GN_FILL_STATS", hand_over)
        names.clear()
        h, outs = x0, []
        with torch.no_grad():
            for (g, b_, w, bias), rows in zip(layers, tiles):
                real("mulan_set_tuning", 23, rows)
                h = ops.gn_conv3x3(h, None, g, b_, w, bias, res=h)
                outs.append(h.clone())
        return outs, names.count("mulan_groupnorm_stats")

    ref, n_ref = run(False, (8, 8, 8, 8))
    for tiles in ((8, 2, 4, 8), (2, 8, 2, 4), (4, 4, 4, 4)):
        got, n_got = run(True, tiles)
        assert n_ref == 4 and n_got == 1
        for i, (a, r) in enumerate(zip(got, ref)):
            assert float((a - r).abs().max()) <= 3e-6 * float(r.abs().max()), (tiles, i)


@pytest.mark.parametrize("B,C,N", [(3, 128, 128), (2, 256, 128), (2, 128, 256)])
@pytest.mark.parametrize("rows", [8, 4, 2])
def test_k_split_blocks_agree_with_four_wave_blocks(ops, monkeypatch, B, C, N, rows):
    """k-split blocks (round 5: launches of at most 256 blocks without a second stream on the chip run as two groups of
    four waves over half of the channels each, sums exchanged through LDS) against the 4-wave blocks on random data: the
    fp32-input launch (output to fp32 rounding of the two half sums, planes bit-identical, maxima = max |y| of its own
    output), a training-mode GroupNorm -> convolution node with the statistics by-product (plane-fed forward and input
    gradient), and a forward-only chain with the normalisation in the patch fill and the statistics handed over"""
    torch.manual_seed(B * C + N + rows)
    x = torch.randn(B, 1024, C, device="cuda") * 1.7
    w = torch.randn(3, 3, C, N, device="cuda") * 0.05
    bias, cb, res = torch.randn(N, device="cuda"), torch.randn(B, N, device="cuda"), torch.randn(B, 1024, N, device="cuda")
    gamma, beta = torch.randn(C, device="cuda") * 0.4 + 1, torch.randn(C, device="cuda") * 0.2
    g2, b2, w2 = torch.randn(N, device="cuda") * 0.4 + 1, torch.randn(N, device="cuda") * 0.2, torch.randn(3, 3, N, N, device="cuda") * 0.04
    monkeypatch.setattr(ops, "GN_FWD_STREAM_B", (1, 1 << 30))
    names = []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])
    L = ops.lib.load()

    def run(ks):
        real("mulan_set_tuning", 23, rows)
        real("mulan_set_tuning", 27, 2 if ks == 2 else 1)
        names.clear()
        y, xs = ops.conv3x3_raw(x, w, bias, cb, res, planes=True)
        ymax = y._absmax[0].view(B, 16).amax(1).clone()
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        z = ops.gn_conv3x3(xg, None, gamma, beta, wg, bias, cbias=cb, keep=0.9, seed=5, offset=64)
        st = z._gnstats[0].clone()
        (z * res).sum().backward()
        with torch.no_grad():                                    # GroupNorm-fed fill, statistics from the launch in front
            f1 = ops.gn_conv3x3(x, None, gamma, beta, w, bias, cbias=cb)
            f2 = ops.gn_conv3x3(f1, None, g2, b2, w2, None, res=f1) if N == 128 else f1
        return [y.clone(), z.detach().clone(), xg.grad.clone(), wg.grad.clone(), st, f1.clone(), f2.clone()], xs.clone(), ymax, list(names)

    ref, ref_xs, ref_max, _ = run(1)
    got, got_xs, got_max, got_names = run(2)
    used_rows = L.mulan_conv3x3_f16x3_tile_rows(B, 32, N, 1)
    assert used_rows == (rows if (32 // rows) * (N // 128) <= 16 else 4)
    assert "mulan_conv3x3_fwd_f16x3_planes_in_stats" in got_names
    assert ("mulan_conv3x3_fwd_f16x3_gn_in" in got_names) == (N == 128)      # (N = 256 keeps the plane hand-over: ops.GN_FILL_MAX_N)
    assert torch.equal(got_xs, ref_xs)
    assert torch.equal(got_max, got[0].abs().amax((1, 2)).view(torch.int32))
    for i, (a, r) in enumerate(zip(got, ref)):
        tol = (2e-6 if i in (0, 1, 4) else 2e-5) * float(r.abs().max())
        assert float((a - r).abs().max()) <= tol, (i, float((a - r).abs().max()), float(r.abs().max()))
    assert not torch.equal(got[0], ref[0])                        # (it did run the other summation order)
