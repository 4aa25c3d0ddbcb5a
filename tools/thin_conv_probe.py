#!/usr/bin/env python3
"""Timing of the thin ends of the U-Nets at B = 128 (conv_out E -> 3 / 1: forward, input gradient, weight gradient):
vector-ALU kernels (default) against the padded MFMA kernels (mulan_set_tuning(8, 1))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd import ops


def timed(fn, reps=20):
    for _ in range(3):
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


ops.lib.load()
ops.CONV_MODE = "f16x3"
B = 128
for C, N in ((128, 3), (128, 1), (256, 3)):
    x, w = torch.randn(B, 1024, C, device="cuda"), torch.randn(3, 3, C, N, device="cuda") * 0.05
    bias, res, dy = torch.randn(N, device="cuda"), torch.randn(B, 1024, N, device="cuda"), torch.randn(B, 1024, N, device="cuda")
    for old in (0, 1):
        ops.call("mulan_set_tuning", 8, old)
        t = (timed(lambda: ops.conv3x3_raw(x, w, bias, None, res)), timed(lambda: ops.conv3x3_dgrad_raw(dy, w)),
             timed(lambda: ops.conv3x3_wgrad_raw(x, dy)))
        print(f"{C:3d} -> {N}  {'MFMA kernels (padded)' if old else 'vector-ALU kernels   '}: forward {t[0]:6.1f} us   "
              f"input gradient {t[1]:6.1f} us   weight gradient {t[2]:6.1f} us", flush=True)
    ops.call("mulan_set_tuning", 8, 0)
