#!/usr/bin/env python3
"""A/B of the GroupNorm normalised inside the convolution's patch fill (mulan_groupnorm_stats +
mulan_conv3x3_fwd_f16x3_gn_in) against the plane hand-over (mulan_groupnorm_fwd_planes + ..._planes_in): a chain of
GroupNorm -> conv3x3 nodes as in the forward pass of the ResnetBlocks (each output feeds the next node, FiLM bias and
residual as in conv1 / conv2), HIP events around 24 nodes, forward only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd import ops


def chain(B, C1, C2, N, fill, n=24):
    ops.GN_FILL = fill
    torch.manual_seed(0)
    Ct = C1 + C2
    x = torch.randn(B, 1024, C1, device="cuda")
    skip = torch.randn(B, 1024, C2, device="cuda") if C2 else None
    gamma, beta = torch.randn(Ct, device="cuda"), torch.randn(Ct, device="cuda") * 0.3
    w = torch.randn(3, 3, Ct, N, device="cuda") * 0.02
    bias, cb = torch.randn(N, device="cuda"), torch.randn(B, N, device="cuda")
    assert N == C1
    with torch.no_grad():
        def run(k):
            h = x
            for i in range(k):
                h = ops.gn_conv3x3(h, skip, gamma, beta, w, bias, cbias=cb if i % 2 == 0 else None, res=h if i % 2 else None)
            return h
        run(4)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(n); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n)
        return sorted(ts)[2]


if __name__ == "__main__":
    ops.lib.load()
    for B, C1, C2, N in ((128, 128, 0, 128), (128, 128, 128, 128), (128, 256, 0, 256), (128, 256, 256, 256), (500, 256, 0, 256)):
        a, b = chain(B, C1, C2, N, False), chain(B, C1, C2, N, True)
        print(f"B={B} [{C1}|{C2}] -> {N}: planes hand-over {a:7.1f} us / node   normalised in the fill {b:7.1f} us / node   ({b / a:.3f}x)", flush=True)
