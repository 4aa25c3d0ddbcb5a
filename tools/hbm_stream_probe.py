#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this box at the GroupNorm kernels' read / write mixes and sizes (the ceiling
the 1.00x-traffic GroupNorm kernels are measured against): torch elementwise kernels on [128, 1024, C] fp32 tensors,
HIP events around 20 launches, cache-cold (a 1 GiB tensor is rewritten between timed launches)."""
import torch


def timed(fn, cold, n=12):
    ts = []
    for _ in range(n):
        cold.add_(1.0)                       # evict L2 / MALL
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    cold = torch.zeros(256 << 20, device="cuda")
    for C in (128, 256):
        x = torch.randn(128, 1024, C, device="cuda"); dy = torch.randn_like(x); out = torch.empty_like(x)
        mb = x.numel() * 4 / 1e6
        for name, fn, nbytes in (("1R+1W copy", lambda: out.copy_(x), 2 * mb), ("1R+1W mul", lambda: torch.mul(x, 1.5, out=out), 2 * mb),
                                 ("2R+1W add", lambda: torch.add(x, dy, out=out), 3 * mb), ("1R sum", lambda: x.sum(), mb),
                                 ("2R+1W addcmul", lambda: torch.addcmul(x, x, dy, out=out), 3 * mb)):
            us = timed(fn, cold)
            print(f"C={C} {name:16s} {us:7.1f} us  {nbytes / us:5.2f} TB/s  ({nbytes:.0f} MB)", flush=True)
        # warm (no eviction): back-to-back launches as in the step, where the producer's output is partly in MALL
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.add(x, dy, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"C={C} 2R+1W add back-to-back {us:7.1f} us  {3 * mb / us:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
