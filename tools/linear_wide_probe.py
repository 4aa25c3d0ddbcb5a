#!/usr/bin/env python3
"""Per-pixel dense kernel (linear_f16x3): 256-column blocks (default where the layer has a multiple of 256 output
columns) against 128-column blocks (mulan_set_tuning(11, 1)); B images, cache-cold inputs (a ring of 3 tensors)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd import ops

ops.lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for K1, K2, N1, N2 in ((256, 256, 256, 0), (256, 0, 256, 256), (256, 0, 256, 0), (256, 0, 768, 0), (128, 0, 128, 128), (128, 0, 256, 0),
                       (128, 0, 384, 0), (128, 128, 128, 0)):
    xs1 = [torch.randn(B, 1024, K1, device="cuda") for _ in range(3)]
    xs2 = [torch.randn(B, 1024, K2, device="cuda") for _ in range(3)] if K2 else [None] * 3
    w = torch.randn(K1 + K2, N1 + N2, device="cuda") * 0.05
    wp, wmax = ops.linear_pack(w, False)
    for x in xs1 + [t for t in xs2 if t is not None]:
        ops.cached_absmax(x)
    res = {}
    for old in (0, 1, 0, 1):
        ops.call("mulan_set_tuning", 11, old)
        for i in range(3):
            ops.linear_f16x3_raw(xs1[i], xs2[i], wp, wmax, N1, N2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(18):
            ops.linear_f16x3_raw(xs1[i % 3], xs2[i % 3], wp, wmax, N1, N2)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(old, []).append(e0.elapsed_time(e1) * 1e3 / 18)
    ops.call("mulan_set_tuning", 11, 0)
    mb = B * 1024 * (K1 + K2 + N1 + N2) * 4 / 1e6
    a, b = min(res[0]), min(res[1])
    print(f"[{K1}|{K2}] -> [{N1}|{N2}]: default {a:7.1f} us ({mb / a:5.2f} TB/s of algorithmic bytes)   128-column blocks {b:7.1f} us ({mb / b:5.2f} TB/s)", flush=True)
