"""Streaming GroupNorm kernels (round 5: mulan_groupnorm_fwd_stream / mulan_groupnorm_bwd_stream) against the register-slab
kernels they restate (same inputs; the statistics / group sums are handed in as the per-(image, 8-row tile, channel quad)
partial sums a producing convolution would leave, formed here in float64) and against the float64 oracle formulas of
nn.GroupNorm + swish + Dropout (ldm/model_vdm.py:622-623,632,643-644; oracle/torch_ref.py group_norm)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HW, G = 1024, 32


@pytest.fixture()
def ops(monkeypatch):
    from mulan_amd import ops as _ops
    _ops.lib.load()
    monkeypatch.setattr(_ops, "CONV_MODE", "f16x3")
    yield _ops
    for k in (20, 21, 22, 23):
        _ops.call("mulan_set_tuning", k, 0)


def partials(v):
    """[B, 1024, C] float64 -> [B, 4, C / 4] sums per image, 8-row tile and channel quad"""
    B, _, C = v.shape
    return v.view(B, 4, 256, C // 4, 4).sum((2, 4))


def xstats_of(x):
    xd = x.double()
    return torch.stack((partials(xd), partials(xd * xd)), -1).float().contiguous()


def decode_planes(planes, bound_bits, B, C):
    """split planes [B][C/16][1024][2][16] fp16 scaled by 2^(140 - e(bound)) -> [B, 1024, C] float64"""
    p = planes.view(torch.float16).view(B, C // 16, HW, 2, 16).double()
    bnd = bound_bits.cpu().numpy().view(np.float32)[:, 0]
    out = torch.empty(B, HW, C, dtype=torch.float64, device=planes.device)
    for b in range(B):
        e = min(max(int(np.frexp(bnd[b])[1]) - 1 + 127, 14), 254)
        out[b] = (p[b, :, :, 0] + p[b, :, :, 1]).permute(1, 0, 2).reshape(HW, C) * 2.0 ** (e - 140)
    return out


def make(B, C1, C2, seed):
    torch.manual_seed(seed)
    x1 = torch.randn(B, HW, C1, device="cuda") * 1.5 + 0.3
    x2 = torch.randn(B, HW, C2, device="cuda") * 0.7 - 0.2 if C2 else None
    if B > 1:
        x1[1] *= 1e-2
    Ct = C1 + C2
    gamma, beta = torch.randn(Ct, device="cuda") * 0.5 + 1.0, torch.randn(Ct, device="cuda") * 0.3
    return x1, x2, gamma, beta


@pytest.mark.parametrize("C1,C2", [(128, 0), (128, 128), (256, 0)])
@pytest.mark.parametrize("keep", [1.0, 0.9])
@pytest.mark.parametrize("nsp", [0, 2, 1])
def test_forward_stream_matches_slab_kernel(ops, C1, C2, keep, nsp):
    B, Ct = 3, C1 + C2
    ops.call("mulan_set_tuning", 20, nsp)
    x1, x2, gamma, beta = make(B, C1, C2, 3)
    p = ops.ptr
    mean, rstd, mean2, rstd2 = (torch.empty(B, G, device="cuda") for _ in range(4))
    bound, bound2 = (torch.empty(B, 16, device="cuda", dtype=torch.int32) for _ in range(2))
    ys, ys2 = (torch.empty(B * HW * Ct * 4, device="cuda", dtype=torch.uint8) for _ in range(2))
    kb, kb2 = (torch.zeros(B * (Ct // 32) * 1024, device="cuda", dtype=torch.int32) for _ in range(2))
    if keep < 1:
        ops.call("mulan_groupnorm_fwd_planes_keepbits", p(x1), p(x2), C1, C2, p(gamma), p(beta), p(ys), p(mean), p(rstd), B, HW, G,
                 1e-6, 1, keep, 77, 128, None, p(bound), p(kb), ops.stream())
    else:
        ops.call("mulan_groupnorm_fwd_planes", p(x1), p(x2), C1, C2, p(gamma), p(beta), p(ys), p(mean), p(rstd), B, HW, G, 1e-6, 1,
                 keep, 77, 128, None, p(bound), ops.stream())
    xs1, xs2 = xstats_of(x1), (xstats_of(x2) if C2 else None)
    ops.call("mulan_groupnorm_fwd_stream", p(x1), p(x2), C1, C2, p(gamma), p(beta), None, p(ys2), p(mean2), p(rstd2), p(xs1),
             p(xs2), 4, B, HW, G, 1e-6, 1, keep, 77, 128, None, p(bound2), p(kb2) if keep < 1 else None, ops.stream())
    torch.cuda.synchronize()
    # statistics from the partial sums: those of the slab kernel to fp32 rounding (another summation order)
    assert float((mean - mean2).abs().max()) <= 2e-6 * float(mean.abs().max() + 1)
    assert float(((rstd - rstd2) / rstd).abs().max()) <= 2e-6
    # the bound is the same number (row maximum of the maxima array), the keep-bits are the same bits
    assert torch.equal(bound.view(B, 16).amax(1), bound2.view(B, 16).amax(1))
    assert torch.equal(kb, kb2)
    a, r = decode_planes(ys2, bound2, B, Ct), decode_planes(ys, bound, B, Ct)
    for b in range(B):
        assert float((a[b] - r[b]).abs().max()) <= 4e-6 * float(r[b].abs().max()), (b, float((a[b] - r[b]).abs().max()))
    # ... and with mean / rstd GIVEN (no partial sums) the fp32 output is the slab kernel's to the last bit or two (the same
    # expressions; the compiler contracts multiply-adds differently in the two kernels), the dropped elements the same
    y, y2 = torch.empty(B, HW, Ct, device="cuda"), torch.empty(B, HW, Ct, device="cuda")
    m1, m2 = (torch.empty(B, 16, device="cuda", dtype=torch.int32) for _ in range(2))
    ops.call("mulan_groupnorm_fwd_dyn", p(x1), p(x2), C1, C2, p(gamma), p(beta), p(y), p(mean), p(rstd), B, HW, G, 1e-6, 1, keep,
             77, 128, None, p(m1), ops.stream())
    ops.call("mulan_groupnorm_fwd_stream", p(x1), p(x2), C1, C2, p(gamma), p(beta), p(y2), None, p(mean), p(rstd), None, None, 0, B,
             HW, G, 1e-6, 1, keep, 77, 128, None, p(m2), None, ops.stream())
    assert torch.equal(y == 0, y2 == 0)
    assert float((y - y2).abs().max()) <= 1e-6 * float(y.abs().max())
    assert torch.equal(m2.view(B, 16).amax(1), y2.abs().amax((1, 2)).view(torch.int32))


def reference_backward(dy, x1, x2, gamma, beta, mean, rstd, mask, keep, add1, add1b):
    """float64: dx (of the concat), dgamma, dbeta and the group sums' partial form the convolution epilogue would leave"""
    x = (x1 if x2 is None else torch.cat((x1, x2), -1)).double()
    B, _, C = x.shape
    cpg = C // G
    m = mean.double().repeat_interleave(cpg, 1)[:, None, :]
    r = rstd.double().repeat_interleave(cpg, 1)[:, None, :]
    xh = (x - m) * r
    u = xh * gamma.double() + beta.double()
    sg = torch.sigmoid(u)
    g = dy.double() * mask.double() / keep * (sg * (1 + u * (1 - sg)))
    da = g * gamma.double()
    gst = torch.stack((partials(da), partials(da * xh)), -1)
    s1 = da.view(B, HW, G, cpg).sum((1, 3)) / (HW * cpg)
    s2 = (da * xh).view(B, HW, G, cpg).sum((1, 3)) / (HW * cpg)
    dx = r * (da - s1.repeat_interleave(cpg, 1)[:, None, :] - xh * s2.repeat_interleave(cpg, 1)[:, None, :])
    C1 = x1.shape[-1]
    if add1 is not None:
        dx[..., :C1] += add1.double()
    if add1b is not None:
        dx[..., :C1] += add1b.double()
    return dx, (g * xh).sum((0, 1)), g.sum((0, 1)), gst.float().contiguous()


@pytest.mark.parametrize("C1,C2", [(128, 0), (128, 128)])
@pytest.mark.parametrize("keep", [1.0, 0.9])
@pytest.mark.parametrize("tune", ["", "21=2", "21=2,22=1", "21=1,22=2", "21=3"])
def test_backward_stream_matches_slab_kernel_and_float64(ops, C1, C2, keep, tune):
    B, Ct = 3, C1 + C2
    for kv in filter(None, tune.split(",")):
        k, v = kv.split("=")
        ops.call("mulan_set_tuning", int(k), int(v))
    thin = tune == "21=3"
    x1, x2, gamma, beta = make(B, C1, C2, 5)
    dy = torch.randn(B, HW, Ct, device="cuda") * 1e-2
    dy[0, 7, 3] = 1.0
    add1, add1b = torch.randn(B, HW, C1, device="cuda") * 1e-2, torch.randn(B, HW, C1, device="cuda") * 1e-2
    p = ops.ptr
    mean, rstd = torch.empty(B, G, device="cuda"), torch.empty(B, G, device="cuda")
    y = torch.empty(B, HW, Ct, device="cuda")
    ones, zeros = torch.ones(Ct, device="cuda"), torch.zeros(Ct, device="cuda")
    # the dropout mask as the kernels draw it: an activation-free pass with gamma = 1, beta = 0 is zero exactly where dropped
    ops.call("mulan_groupnorm_fwd_dyn", p(x1), p(x2), C1, C2, p(ones), p(zeros), p(y), p(mean), p(rstd), B, HW, G, 1e-6, 0, keep, 77,
             128, None, None, ops.stream())
    mask = (y != 0).float() if keep < 1 else torch.ones_like(y)
    ref_dx, ref_dg, ref_db, gst = reference_backward(dy, x1, x2, gamma, beta, mean, rstd, mask, keep, add1, add1b)
    tick = torch.zeros(16, device="cuda", dtype=torch.int32)

    def run(stream):
        dx1 = torch.empty(B, HW, C1, device="cuda")
        dx2 = torch.empty(B, HW, C2, device="cuda") if C2 else None
        parts = torch.zeros(3, 4 * B, Ct, device="cuda")
        dg, db, sink = (torch.zeros(n, device="cuda") for n in (Ct, Ct, C1))
        mx1 = torch.empty(B, 16, device="cuda", dtype=torch.int32)
        mx2 = torch.empty(B, 16, device="cuda", dtype=torch.int32) if C2 else None
        if stream:
            ops.call("mulan_groupnorm_bwd_stream", p(dy), None, p(x1), p(x2), C1, C2, p(gamma), p(beta), p(mean), p(rstd), p(gst),
                     p(dx1), p(dx2), None, p(parts[0]), p(parts[1]), B, HW, G, 1, keep, 77, 128, None, p(mx1), p(mx2), p(add1), None,
                     p(add1b), p(parts[2]), p(dg), p(db), p(sink), None, p(tick), None, ops.stream())
        else:
            ops.call("mulan_groupnorm_bwd_fused", p(dy), p(x1), p(x2), C1, C2, p(gamma), p(beta), p(mean), p(rstd), p(dx1), p(dx2),
                     p(parts[0]), p(parts[1]), B, HW, G, 1, keep, 77, 128, None, p(mx1), p(mx2), p(add1), None, p(add1b), p(parts[2]),
                     p(dg), p(db), p(sink), None, p(tick), ops.stream())
        dx = dx1 if dx2 is None else torch.cat((dx1, dx2), -1)
        return dx, dg, db, sink, mx1, mx2, parts

    dx_s, dg_s, db_s, sink_s, mx1_s, mx2_s, _ = run(False)
    dx_t, dg_t, db_t, sink_t, mx1_t, mx2_t, parts = run(True)
    torch.cuda.synchronize()
    assert int(tick.abs().sum()) == 0
    scale = float(ref_dx.abs().max())
    e_slab, e_stream = float((dx_s.double() - ref_dx).abs().max()) / scale, float((dx_t.double() - ref_dx).abs().max()) / scale
    assert e_stream <= 2.0 * e_slab + 1e-6, (e_stream, e_slab)          # as close to float64 as the slab kernel
    assert torch.equal(mx1_t.view(B, 16).amax(1), dx_t[..., :C1].abs().amax((1, 2)).view(torch.int32))
    if C2:
        assert torch.equal(mx2_t.view(B, 16).amax(1), dx_t[..., C1:].abs().amax((1, 2)).view(torch.int32))
    if thin:
        # the thin form leaves the per-block partial rows [B * 4][Ct]: their sums are the totals
        dg_t, db_t = parts[0].double().sum(0).float(), parts[1].double().sum(0).float()
        sink_t = parts[2].double().sum(0).float()[:C1]
    for got, slab, ref in ((dg_t, dg_s, ref_dg), (db_t, db_s, ref_db)):
        s = float(ref.abs().max())
        assert float((got.double() - ref).abs().max()) / s <= 2.0 * float((slab.double() - ref).abs().max()) / s + 2e-6
    ref_sink = ref_dx[..., :C1].sum((0, 1))
    assert float((sink_t.double() - ref_sink).abs().max()) <= 1e-5 * float(ref_dx.abs().sum((0, 1)).max())


@pytest.mark.parametrize("keep", [1.0, 0.9])
@pytest.mark.parametrize("tune", ["", "21=2,22=1", "21=3"])
def test_backward_stream_planes_output(ops, keep, tune):
    """dx as split planes (the gradient norm2 hands to conv1): the planes of mulan_groupnorm_bwd_fused_planes to rounding,
    the same bound, the stored keep-bits instead of the re-draw give the same planes bit for bit"""
    B, C = 3, 128
    for kv in filter(None, tune.split(",")):
        k, v = kv.split("=")
        ops.call("mulan_set_tuning", int(k), int(v))
    x, _, gamma, beta = make(B, C, 0, 11)
    dy = torch.randn(B, HW, C, device="cuda") * 1e-2
    p = ops.ptr
    mean, rstd = torch.empty(B, G, device="cuda"), torch.empty(B, G, device="cuda")
    ys = torch.empty(B * HW * C * 4, device="cuda", dtype=torch.uint8)
    bd = torch.empty(B, 16, device="cuda", dtype=torch.int32)
    kb = torch.zeros(B * (C // 32) * 1024, device="cuda", dtype=torch.int32)
    y = torch.empty(B, HW, C, device="cuda")
    ones, zeros = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    if keep < 1:
        ops.call("mulan_groupnorm_fwd_planes_keepbits", p(x), None, C, 0, p(gamma), p(beta), p(ys), p(mean), p(rstd), B, HW, G, 1e-6, 1,
                 keep, 5, 0, None, p(bd), p(kb), ops.stream())
    ops.call("mulan_groupnorm_fwd_dyn", p(x), None, C, 0, p(ones), p(zeros), p(y), p(mean), p(rstd), B, HW, G, 1e-6, 0, keep, 5, 0, None,
             None, ops.stream())
    mask = (y != 0).float() if keep < 1 else torch.ones_like(y)
    ref_dx, _, _, gst = reference_backward(dy, x, None, gamma, beta, mean, rstd, mask, keep, None, None)
    dymax = ops.absmax_rows(dy.view(B, -1))
    tick = torch.zeros(16, device="cuda", dtype=torch.int32)

    def run(stream, bits):
        dxp = torch.empty(B * HW * C * 4, device="cuda", dtype=torch.uint8)
        m = torch.empty(B, 16, device="cuda", dtype=torch.int32)
        parts = torch.zeros(3, 4 * B, C, device="cuda")
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        if stream:
            ops.call("mulan_groupnorm_bwd_stream", p(dy), p(dymax), p(x), None, C, 0, p(gamma), p(beta), p(mean), p(rstd), p(gst), None,
                     None, p(dxp), p(parts[0]), p(parts[1]), B, HW, G, 1, keep, 5, 0, None, p(m), None, None, None, None, p(parts[2]),
                     p(dg), p(db), None, None, p(tick), p(kb) if bits else None, ops.stream())
        else:
            ops.call("mulan_groupnorm_bwd_fused_planes", p(dy), p(dymax), p(x), C, p(gamma), p(beta), p(mean), p(rstd), p(dxp),
                     p(parts[0]), p(parts[1]), B, HW, G, 1, keep, 5, 0, None, p(m), p(parts[2]), p(dg), p(db), None, None, p(tick),
                     p(kb) if bits else None, ops.stream())
        return dxp, m

    slab, m_slab = run(False, False)
    got, m_got = run(True, False)
    assert torch.equal(m_slab.view(B, 16).amax(1), m_got.view(B, 16).amax(1))
    if keep < 1:
        got_k, _ = run(True, True)
        assert torch.equal(got, got_k)
    a, r = decode_planes(got, m_got, B, C), decode_planes(slab, m_slab, B, C)
    scale = float(ref_dx.abs().max())
    assert float((a - ref_dx).abs().max()) / scale <= 2.0 * float((r - ref_dx).abs().max()) / scale + 1e-6


@pytest.mark.parametrize("B,keep,concat", [(3, 0.9, False), (2, 1.0, True), (3, 0.9, True)])
@pytest.mark.parametrize("rows", [8, 2])
def test_training_chain_runs_the_streaming_forward_on_the_producers_statistics(ops, monkeypatch, B, keep, concat, rows):
    """the shipped use (ops.GnConv3x3Fn, training): a plane-fed convolution leaves the partial sums of its output
    (mulan_conv3x3_fwd_f16x3_planes_in_stats, at 8- and 2-row tiles), the GroupNorm behind it -- norm2 with dropout, or the
    up path's norm1 over the concat of two such outputs -- runs mulan_groupnorm_fwd_stream on them; outputs and every
    gradient against the slab-kernel path (MULAN_GN_FWD_STREAM=0) to the rounding of the statistics, the ystats against
    float64 sums, the same dropout bits"""
    ops.call("mulan_set_tuning", 23, rows)
    torch.manual_seed(B + int(keep * 10) + rows)
    E = 128
    mk = lambda *s, scale=1.0: (torch.randn(*s, device="cuda") * scale).requires_grad_(True)
    x, xb = mk(B, HW, E, scale=2.0), mk(B, HW, E)
    g0, b0, w0, c0 = mk(E), mk(E, scale=0.3), mk(3, 3, E, E, scale=0.03), mk(E)
    Ct = 2 * E if concat else E
    g1, b1, w1, c1 = mk(Ct), mk(Ct, scale=0.3), mk(3, 3, Ct, E, scale=0.03), mk(E)
    gy = torch.randn(B, HW, E, device="cuda")
    leaves = [x, xb, g0, b0, w0, c0, g1, b1, w1, c1]
    names, stats = [], []
    real = ops.call
    monkeypatch.setattr(ops, "call", lambda n, *a: (names.append(n), real(n, *a))[1])

    monkeypatch.setattr(ops, "GN_FWD_STREAM_B", (1, 1 << 30))    # (the product path: 32 ... 96 images per launch)

    def run(stream):
        monkeypatch.setattr(ops, "GN_FWD_STREAM", stream)
        for t in leaves:
            t.grad = None
        names.clear()
        h = ops.gn_conv3x3(x, None, g0, b0, w0, c0, act=True)                    # leaf input: the slab kernel
        h2 = ops.gn_conv3x3(xb, None, g0, b0, w0, c0, res=xb, act=True) if concat else None
        st = getattr(h, "_gnstats", None)
        stats.append(None if st is None else (st[0].clone(), h.detach().clone()))
        y = ops.gn_conv3x3(h, h2, g1, b1, w1, c1, act=True, keep=keep, seed=11, offset=1 << 20, x1_grad_planes=not concat)
        (y * gy).sum().backward()
        return [y.detach().clone()] + [t.grad.clone() for t in leaves if t.grad is not None], list(names)

    ref, ref_names = run(False)
    got, got_names = run(True)
    assert "mulan_groupnorm_fwd_stream" in got_names and "mulan_groupnorm_fwd_stream" not in ref_names
    assert got_names.count("mulan_groupnorm_fwd_stream") == 1      # the leaf-fed GroupNorms have no statistics: slab kernel
    assert stats[0] is None and stats[1] is not None
    st, h = stats[1]
    assert st.shape == (B, 32 // rows, E // 4, 2)
    hd = h.double().view(B, 32 // rows, rows * 32, E // 4, 4)
    want = torch.stack((hd.sum((2, 4)), (hd * hd).sum((2, 4))), -1)
    assert float((st.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    assert len(got) == len(ref)
    for i, (a, r) in enumerate(zip(got, ref)):
        assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-30, (i, float((a - r).abs().max()), float(r.abs().max()))
