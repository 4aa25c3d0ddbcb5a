"""AdamW + EMA step alone (optim.hip adamw_ema_kernel) at the parameter count of the headline model: the dev variants of
tune[25] (loads in flight per thread, non-temporal access) and tune[26] (block cap); 36 B of HBM traffic per parameter.
    python tools/adamw_bench.py [--n 35600000]"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from mulan_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=35_600_000)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    ops.lib.load()
    n = a.n
    p, g, m, v, e = (torch.randn(n, device="cuda") * s for s in (0.05, 1e-3, 1e-3, 1e-6, 0.05))
    v = v.abs()
    ref = None
    for var, cap in ((1, 2048), (1, 0), (2, 2048), (3, 2048), (3, 0), (0, 2048), (0, 4096), (0, 0), (0, 16384)):
        ops.call("mulan_set_tuning", 25, var)
        ops.call("mulan_set_tuning", 26, cap)
        st = [t.clone() for t in (p, m, v, e)]
        ops.adamw_ema_step(st[0], g, st[1], st[2], st[3], n - 1000, 2e-4, 0.9, 0.99, 1e-8, 0.01, 3, 0.9999)
        if ref is None:
            ref = [t.clone() for t in st]
        same = all(torch.equal(x, r) for x, r in zip(st, ref))
        for _ in range(3):
            ops.adamw_ema_step(st[0], g, st[1], st[2], st[3], n - 1000, 2e-4, 0.9, 0.99, 1e-8, 0.01, 3, 0.9999)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.iters):
            ops.adamw_ema_step(st[0], g, st[1], st[2], st[3], n - 1000, 2e-4, 0.9, 0.99, 1e-8, 0.01, 3, 0.9999)
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1000 / a.iters
        print(f"tune25={var} cap={cap or 8192:5d}: {us:7.1f} us  {36.0 * n / us / 1e6:5.2f} TB/s  same bits as variant 1: {same}")
    ops.call("mulan_set_tuning", 25, 0)
    ops.call("mulan_set_tuning", 26, 0)


if __name__ == "__main__":
    main()
