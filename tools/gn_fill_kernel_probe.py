#!/usr/bin/env python3
"""Kernel-level costs behind tools/gn_fill_probe.py: each kernel alone (40 back-to-back launches on the same operands, HIP
events), B = 128: the plane-fed convolution, the fp32-input one (splits and stores planes), its ablation 16 (no split
arithmetic, no plane stores), the GroupNorm-fed one with / without SiLU, and the GroupNorm planes / statistics kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd import ops
from mulan_amd.lib import call, ptr, stream


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[2]


def main():
    L = ops.lib.load()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    for C, N in ((128, 128), (256, 256)):
        torch.manual_seed(0)
        x = torch.randn(B, 1024, C, device="cuda")
        w = torch.randn(3, 3, C, N, device="cuda") * 0.05
        g, b_ = torch.randn(C, device="cuda"), torch.randn(C, device="cuda") * 0.3
        bias, cb = torch.randn(N, device="cuda"), torch.randn(B, N, device="cuda")
        wmax = ops.absmax_rows(w.view(1, -1))
        wp, _ = ops._pack_weights(w, C, N, 0, wmax)
        xmax = ops.absmax_rows(x)
        ys = torch.empty(B * 1024 * C * 4, device="cuda", dtype=torch.uint8)
        bound = torch.empty(B, 16, device="cuda", dtype=torch.int32)
        mean, rstd = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
        y = torch.empty(B, 1024, N, device="cuda")
        ym = torch.empty(B, 16, device="cuda", dtype=torch.int32) if N // 128 * 4 <= 16 else None
        gn_planes = lambda: call("mulan_groupnorm_fwd_planes", ptr(x), None, C, 0, ptr(g), ptr(b_), ptr(ys), ptr(mean), ptr(rstd),
                                 B, 1024, 32, 1e-6, 1, 1.0, 0, 0, None, ptr(bound), stream())
        gn_stats = lambda: call("mulan_groupnorm_stats", ptr(x), None, C, 0, ptr(g), ptr(b_), ptr(mean), ptr(rstd), ptr(bound),
                                B, 1024, 32, 1e-6, stream())
        pin = lambda: call("mulan_conv3x3_fwd_f16x3_planes_in", ptr(ys), ptr(bound), ptr(wp), ptr(wmax), ptr(bias), ptr(cb), 1,
                           None, ptr(y), ptr(ym), B, 32, 32, C, N, stream())
        f32 = lambda: call("mulan_conv3x3_fwd_f16x3", ptr(x), ptr(xmax), ptr(wp), ptr(wmax), ptr(bias), ptr(cb), 1, None,
                           ptr(y), ptr(ys), ptr(ym), B, 32, 32, C, N, stream())
        gnin = lambda act, planes: call("mulan_conv3x3_fwd_f16x3_gn_in", ptr(x), None, C, 0, ptr(g), ptr(b_), ptr(mean),
                                        ptr(rstd), 32, act, ptr(bound), ptr(wp), ptr(wmax), ptr(bias), ptr(cb), 1, None,
                                        ptr(y), ptr(ym), ptr(ys) if planes else None, B, 32, 32, N, stream())
        gn_planes()
        res = [("GroupNorm -> planes", timed(gn_planes)), ("GroupNorm statistics", timed(gn_stats))]
        gn_planes()
        res.append(("conv, plane-fed", timed(pin)))
        res.append(("conv, fp32 input (splits, stores planes)", timed(f32)))
        call("mulan_set_tuning", 4, 16)
        res.append(("conv, fp32 input, ablation 16 (no split, no stores)", timed(f32)))
        call("mulan_set_tuning", 4, 32)
        res.append(("conv, fp32 input, ablation 32 (no split, stores)", timed(f32)))
        call("mulan_set_tuning", 4, 0)
        gn_stats()
        res.append(("conv, GroupNorm-fed, SiLU", timed(lambda: gnin(1, False))))
        res.append(("conv, GroupNorm-fed, no activation", timed(lambda: gnin(0, False))))
        res.append(("conv, GroupNorm-fed, SiLU, stores planes", timed(lambda: gnin(1, True))))
        for n, t in res:
            print(f"B={B} {C}->{N}  {n:55s} {t:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
