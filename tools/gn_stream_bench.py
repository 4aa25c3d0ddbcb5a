#!/usr/bin/env python3
"""GroupNorm kernels alone: the register-slab kernels (mulan_groupnorm_fwd_planes / _bwd_fused / _bwd_fused_planes) against
the streaming forms of round 5 (mulan_groupnorm_fwd_stream / _bwd_stream), on rotating buffer sets (--sets, default 6:
1.2-2.4 GB, far beyond the 256 MB Infinity Cache: every launch streams from HBM) and on one set (cache-warm).
Also checks the streaming results against the slab kernels' (same inputs, statistics / group sums formed with torch).
Usage: python tools/gn_stream_bench.py [--batch 128] [--reps 30] [--sets 6]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402
from mulan_amd.lib import call, ptr, stream  # noqa: E402

HW = 1024


def timeit(fns, reps):
    """fns: one closure per buffer set, called round-robin"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        fns[i % len(fns)]()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def xstats_of(x):
    """[B, 1024, C] -> [B, 4, C / 4, 2]: sum and sum of squares per image, 8-row tile, channel quad"""
    B, _, C = x.shape
    v = x.view(B, 4, 256, C // 4, 4)
    return torch.stack((v.sum((2, 4)), (v * v).sum((2, 4))), -1).contiguous()


def gstats_of(dy, x1, x2, gamma, beta, mean, rstd, act, G):
    x = x1 if x2 is None else torch.cat((x1, x2), -1)
    B, _, C = x.shape
    cpg = C // G
    m = mean.repeat_interleave(cpg, 1)[:, None, :]
    r = rstd.repeat_interleave(cpg, 1)[:, None, :]
    xh = (x - m) * r
    u = xh * gamma + beta
    if act:
        sg = torch.sigmoid(u)
        g = dy * (sg * (1 + u * (1 - sg)))
    else:
        g = dy
    da = g * gamma
    v1 = da.view(B, 4, 256, C // 4, 4)
    v2 = (da * xh).view(B, 4, 256, C // 4, 4)
    return torch.stack((v1.sum((2, 4)), v2.sum((2, 4))), -1).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--tune", default="")
    a = ap.parse_args()
    ops.lib.load()
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        call("mulan_set_tuning", int(k), int(v))
    B, dev, G = a.batch, "cuda", 32
    for C1, C2 in ((128, 0), (128, 128), (256, 0)):
        Ct = C1 + C2
        gamma, beta = torch.randn(Ct, device=dev) * 0.3 + 1.0, torch.randn(Ct, device=dev) * 0.2
        sets = []
        for _ in range(a.sets):
            d = dict(x1=torch.randn(B, HW, C1, device=dev) * 1.5 + 0.3,
                     x2=(torch.randn(B, HW, C2, device=dev) if C2 else None),
                     dy=torch.randn(B, HW, Ct, device=dev) * 1e-3,
                     ys=torch.empty(B * HW * Ct * 4, device=dev, dtype=torch.uint8),
                     dx1=torch.empty(B, HW, C1, device=dev), dx2=(torch.empty(B, HW, C2, device=dev) if C2 else None),
                     add1=torch.randn(B, HW, C1, device=dev) * 1e-3,
                     dxp=torch.empty(B * HW * C1 * 4, device=dev, dtype=torch.uint8),
                     kb=torch.empty(B * (Ct // 32) * 1024, device=dev, dtype=torch.int32))
            sets.append(d)
        mean, rstd = torch.empty(B, G, device=dev), torch.empty(B, G, device=dev)
        mean2, rstd2 = torch.empty(B, G, device=dev), torch.empty(B, G, device=dev)
        bound = torch.empty(B, 16, device=dev, dtype=torch.int32)
        bound2 = torch.empty(B, 16, device=dev, dtype=torch.int32)
        parts = torch.empty(3, 4 * B, Ct, device=dev)
        dg, db, dg2, db2 = (torch.empty(Ct, device=dev) for _ in range(4))
        tick = torch.zeros(16, device=dev, dtype=torch.int32)
        m1, m2 = torch.empty(B, 16, device=dev, dtype=torch.int32), torch.empty(B, 16, device=dev, dtype=torch.int32)
        dymax = ops.absmax_rows(sets[0]["dy"].view(B, -1))
        s0 = sets[0]
        xs1 = [xstats_of(s["x1"]) for s in sets]
        xs2 = [xstats_of(s["x2"]) if C2 else None for s in sets]

        def fwd_slab(s, keep, kb=False):
            if kb:
                call("mulan_groupnorm_fwd_planes_keepbits", ptr(s["x1"]), ptr(s["x2"]), C1, C2, ptr(gamma), ptr(beta), ptr(s["ys"]),
                     ptr(mean), ptr(rstd), B, HW, G, 1e-6, 1, keep, 123, 0, None, ptr(bound), ptr(s["kb"]), stream())
            else:
                call("mulan_groupnorm_fwd_planes", ptr(s["x1"]), ptr(s["x2"]), C1, C2, ptr(gamma), ptr(beta), ptr(s["ys"]),
                     ptr(mean), ptr(rstd), B, HW, G, 1e-6, 1, keep, 123, 0, None, ptr(bound), stream())

        def fwd_stream(s, i, keep, kb=False, y=None):
            call("mulan_groupnorm_fwd_stream", ptr(s["x1"]), ptr(s["x2"]), C1, C2, ptr(gamma), ptr(beta), ptr(y),
                 None if y is not None else ptr(s["ys"]), ptr(mean2), ptr(rstd2), ptr(xs1[i]), ptr(xs2[i]), 4, B, HW, G, 1e-6, 1,
                 keep, 123, 0, None, ptr(bound2), ptr(s["kb"]) if kb else None, stream())

        # ---- correctness of the streaming forward against the slab kernel (statistics: another summation order)
        fwd_slab(s0, 1.0)
        ref = s0["ys"].clone()
        fwd_stream(s0, 0, 1.0)
        torch.cuda.synchronize()
        same = (ref == s0["ys"]).float().mean().item()
        print(f"C={C1}+{C2}: fwd planes bytes equal {same * 100:.3f} %  mean maxdiff {float((mean - mean2).abs().max()):.2e} "
              f"rstd rel {float(((rstd - rstd2) / rstd).abs().max()):.2e} bound equal {bool((bound == bound2).all())}")
        for keep in (1.0, 0.9):
            for label, nset in (("hbm ", a.sets), ("warm", 1)):
                t0 = timeit([(lambda s=s: fwd_slab(s, keep, keep < 1)) for s in sets[:nset]], a.reps)
                t1 = timeit([(lambda s=s, i=i: fwd_stream(s, i, keep, keep < 1)) for i, s in enumerate(sets[:nset])], a.reps)
                by = 2.0 * B * HW * Ct * 4
                print(f"  fwd C={C1}+{C2} keep={keep} {label}: slab {t0 * 1e6:7.1f} us ({by / t0 / 1e12:5.2f} TB/s)   "
                      f"stream {t1 * 1e6:7.1f} us ({by / t1 / 1e12:5.2f} TB/s)")

        # ---- backward
        fwd_slab(s0, 1.0)
        gst = [gstats_of(s["dy"], s["x1"], s["x2"], gamma, beta, mean, rstd, 1, G) for s in sets[:1]]
        gst = gst + [gst[0]] * (a.sets - 1)      # (timing only on the other sets)

        def bwd_slab(s, keep, adds):
            call("mulan_groupnorm_bwd_fused", ptr(s["dy"]), ptr(s["x1"]), ptr(s["x2"]), C1, C2, ptr(gamma), ptr(beta), ptr(mean),
                 ptr(rstd), ptr(s["dx1"]), ptr(s["dx2"]), ptr(parts[0]), ptr(parts[1]), B, HW, G, 1, keep, 123, 0, None,
                 ptr(m1), ptr(m2) if C2 else None, ptr(s["add1"]) if adds else None, None, None, ptr(parts[2]), ptr(dg),
                 ptr(db), None, None, ptr(tick), stream())

        def bwd_stream(s, i, keep, adds, planes=False):
            call("mulan_groupnorm_bwd_stream", ptr(s["dy"]), ptr(dymax), ptr(s["x1"]), ptr(s["x2"]), C1, C2, ptr(gamma), ptr(beta),
                 ptr(mean), ptr(rstd), ptr(gst[i]), None if planes else ptr(s["dx1"]), ptr(s["dx2"]),
                 ptr(s["dxp"]) if planes else None, ptr(parts[0]), ptr(parts[1]), B, HW, G, 1, keep, 123, 0, None, ptr(m1),
                 ptr(m2) if C2 else None, ptr(s["add1"]) if adds else None, None, None, ptr(parts[2]), ptr(dg2), ptr(db2),
                 None, None, ptr(tick), None, stream())

        def bwd_slab_planes(s, keep):
            call("mulan_groupnorm_bwd_fused_planes", ptr(s["dy"]), ptr(dymax), ptr(s["x1"]), C1, ptr(gamma), ptr(beta), ptr(mean),
                 ptr(rstd), ptr(s["dxp"]), ptr(parts[0]), ptr(parts[1]), B, HW, G, 1, keep, 123, 0, None, ptr(m1),
                 ptr(parts[2]), ptr(dg), ptr(db), None, None, ptr(tick), None, stream())

        bwd_slab(s0, 1.0, True)
        r1 = s0["dx1"].clone()
        r2 = s0["dx2"].clone() if C2 else None
        bwd_stream(s0, 0, 1.0, True)
        torch.cuda.synchronize()
        sc = float(r1.abs().max())
        print(f"C={C1}+{C2}: bwd dx1 maxdiff/scale {float((r1 - s0['dx1']).abs().max()) / sc:.2e}"
              + (f" dx2 {float((r2 - s0['dx2']).abs().max()) / sc:.2e}" if C2 else "")
              + f" dgamma rel {float((dg - dg2).abs().max() / dg.abs().max()):.2e} dbeta rel {float((db - db2).abs().max() / db.abs().max()):.2e}")
        for keep, adds in ((1.0, False), (0.9, False), (1.0, True)):
            for label, nset in (("hbm ", a.sets), ("warm", 1)):
                t0 = timeit([(lambda s=s: bwd_slab(s, keep, adds)) for s in sets[:nset]], a.reps)
                t1 = timeit([(lambda s=s, i=i: bwd_stream(s, i, keep, adds)) for i, s in enumerate(sets[:nset])], a.reps)
                by = (4.0 if adds else 3.0) * B * HW * Ct * 4
                print(f"  bwd C={C1}+{C2} keep={keep} adds={adds} {label}: slab {t0 * 1e6:7.1f} us ({by / t0 / 1e12:5.2f} TB/s)   "
                      f"stream {t1 * 1e6:7.1f} us ({by / t1 / 1e12:5.2f} TB/s)")
        if C2 == 0:
            for keep in (1.0, 0.9):
                t0 = timeit([(lambda s=s: bwd_slab_planes(s, keep)) for s in sets], a.reps)
                t1 = timeit([(lambda s=s, i=i: bwd_stream(s, i, keep, False, True)) for i, s in enumerate(sets)], a.reps)
                by = 3.0 * B * HW * Ct * 4
                print(f"  bwd->planes C={C1} keep={keep} hbm : slab {t0 * 1e6:7.1f} us ({by / t0 / 1e12:5.2f} TB/s)   "
                      f"stream {t1 * 1e6:7.1f} us ({by / t1 / 1e12:5.2f} TB/s)")
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
