#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on one MI355X (HIP events, random data): prints TFLOP/s or GB/s per kernel.
Usage: python tools/kbench.py [--batch 128] [--reps 20] [--only conv,wgrad,gn,gemm,misc] [--tune k=v,...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402
from mulan_amd.lib import call, ptr, stream  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="conv,wgrad,gn,gemm,misc")
    ap.add_argument("--tune", default="")
    ap.add_argument("--conv-mode", default=None)
    a = ap.parse_args()
    ops.lib.load()
    if a.conv_mode:
        ops.CONV_MODE = a.conv_mode
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        call("mulan_set_tuning", int(k), int(v))
    B, only = a.batch, set(a.only.split(","))
    dev = "cuda"
    r = lambda *s: torch.randn(*s, device=dev)
    if "conv" in only:
        for C, N in ((128, 128), (256, 128), (128, 256), (16, 128), (128, 3)):
            x, w = r(B, 1024, C), r(3, 3, C, N) * 0.05
            bias, cb, res = r(N), r(B, N), r(B, 1024, N)
            t = timeit(lambda: ops.conv3x3_raw(x, w, bias, cb, res), a.reps)
            fl = 2.0 * B * 1024 * 9 * C * N
            print(f"conv3x3_fwd  C={C:3d} N={N:3d}: {t*1e6:8.1f} us  {fl/t/1e12:7.2f} TFLOP/s  ({fl/t/1e12/157.3*100:5.1f}% of fp32 MFMA peak)")
    if "wgrad" in only:
        for C, N in ((128, 128), (256, 128), (16, 128), (128, 3)):
            x, dy = r(B, 1024, C), r(B, 1024, N)
            t = timeit(lambda: ops.conv3x3_wgrad_raw(x, dy), a.reps)
            fl = 2.0 * B * 1024 * 9 * C * N
            print(f"conv3x3_wgrad C={C:3d} N={N:3d}: {t*1e6:8.1f} us  {fl/t/1e12:7.2f} TFLOP/s  ({fl/t/1e12/157.3*100:5.1f}%)  [incl. slab reduce]")
    if "gn" in only:
        for C1, C2 in ((128, 0), (128, 128)):
            x1 = r(B, 1024, C1)
            x2 = r(B, 1024, C2) if C2 else None
            g, b_ = r(C1 + C2), r(C1 + C2)
            y = torch.empty(B, 1024, C1 + C2, device=dev)
            mean, rstd = torch.empty(B, 32, device=dev), torch.empty(B, 32, device=dev)
            f = lambda keep: call("mulan_groupnorm_fwd", ptr(x1), ptr(x2), C1, C2, ptr(g), ptr(b_), ptr(y), ptr(mean),
                                  ptr(rstd), B, 1024, 32, 1e-6, 1, keep, 123, 0, None, stream())
            for keep in (1.0, 0.9):
                t = timeit(lambda: f(keep), a.reps)
                by = 2.0 * B * 1024 * (C1 + C2) * 4
                print(f"groupnorm_fwd C={C1}+{C2} keep={keep}: {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s algorithmic")
            dy = r(B, 1024, C1 + C2)
            dx1 = torch.empty_like(x1)
            dx2 = torch.empty_like(x2) if C2 else None
            dgp, dbp = torch.empty(B, C1 + C2, device=dev), torch.empty(B, C1 + C2, device=dev)
            fb = lambda: call("mulan_groupnorm_bwd", ptr(dy), ptr(x1), ptr(x2), C1, C2, ptr(g), ptr(b_), ptr(mean),
                              ptr(rstd), ptr(dx1), ptr(dx2), ptr(dgp), ptr(dbp), B, 1024, 32, 1, 0.9, 123, 0, 0, None, None, None, None, None, stream())
            t = timeit(fb, a.reps)
            by = 3.0 * B * 1024 * (C1 + C2) * 4
            print(f"groupnorm_bwd C={C1}+{C2} keep=0.9: {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s algorithmic (x, dy -> dx)")
            # the variants the train step launches: 1-pass kernel with maxima, skip-gradient adds, channel sums
            m1 = torch.empty(B, 16, device=dev, dtype=torch.int32)
            m2 = torch.empty(B, 16, device=dev, dtype=torch.int32) if C2 else None
            a1, a2 = torch.randn_like(x1), (torch.randn_like(x2) if C2 else None)
            cs = torch.empty(B, C1 + C2, device=dev)
            for keep, adds in ((1.0, False), (0.9, False), (1.0, True)):
                fb2 = lambda: call("mulan_groupnorm_bwd", ptr(dy), ptr(x1), ptr(x2), C1, C2, ptr(g), ptr(b_), ptr(mean),
                                   ptr(rstd), ptr(dx1), ptr(dx2), ptr(dgp), ptr(dbp), B, 1024, 32, 1, keep, 123, 0, 0,
                                   ptr(m1), ptr(m2), ptr(a1) if adds else None, ptr(a2) if adds else None, ptr(cs), stream())
                t = timeit(fb2, a.reps)
                by = (4.0 if adds else 3.0) * B * 1024 * (C1 + C2) * 4
                print(f"groupnorm_bwd 1-pass C={C1}+{C2} keep={keep} skip-adds={adds}: {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s algorithmic")
    if "gemm" in only:
        M = B * 1024
        for (m, n, k, ta, tb, tag) in ((M, 128, 128, 0, 0, "nin/qkv fwd"), (M, 128, 128, 0, 1, "dx"),
                                       (128, 128, M, 1, 0, "dW split-K"), (B, 3072, 3072, 0, 0, "gamma MLP"),
                                       (B, 128, 512, 0, 0, "cond_proj")):
            A = r(k, m) if ta else r(m, k)
            Bm = r(n, k) if tb else r(k, n)
            t = timeit(lambda: ops.gemm_raw(A, Bm, m, n, k, transA=bool(ta), transB=bool(tb)), a.reps)
            fl = 2.0 * m * n * k
            print(f"gemm {tag:12s} M={m} N={n} K={k} ta={ta} tb={tb}: {t*1e6:8.1f} us  {fl/t/1e12:7.2f} TFLOP/s")
        S, C = 1024, 128
        q, k_, v = r(B, S, C), r(B, S, C), r(B, S, C)
        t = timeit(lambda: ops.AttentionFn.apply(q, k_, v), max(2, a.reps // 4))
        print(f"attention fwd B={B}: {t*1e6:8.1f} us  {4.0*B*S*S*C/t/1e12:7.2f} TFLOP/s")
    if "misc" in only:
        dy = r(B * 1024, 128)
        t = timeit(lambda: ops.colsum_raw(dy, B, 1024, 128), a.reps)
        print(f"colsum per-sample: {t*1e6:8.1f} us  {dy.numel()*4/t/1e9:8.1f} GB/s")
        t = timeit(lambda: ops.colsum_raw(dy, 1, B * 1024, 128), a.reps)
        print(f"colsum total     : {t*1e6:8.1f} us  {dy.numel()*4/t/1e9:8.1f} GB/s")
        n = 71_200_000
        p, g, m_, v_, e_ = (torch.randn(n, device=dev) for _ in range(5))
        v_.abs_()
        t = timeit(lambda: ops.adamw_ema_step(p, g, m_, v_, e_, n - 100000, 2e-4, 0.9, 0.99, 1e-8, 0.01, 5, 0.9999), a.reps)
        print(f"adamw_ema 71.2M  : {t*1e6:8.1f} us  {36.0*n/t/1e9:8.1f} GB/s algorithmic (36 B/param)")




def stamps(batch, C=128, N=128, tune="", mode=None):
    """dev: phase timing of block 0 of one conv3x3 launch from in-kernel s_memtime stamps (100 MHz ticks)."""
    ops.lib.load()
    if mode:
        ops.CONV_MODE = mode
    for kv in filter(None, tune.split(",")):
        k, v = kv.split("=")
        call("mulan_set_tuning", int(k), int(v))
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    x, w = torch.randn(batch, 1024, C, device="cuda"), torch.randn(3, 3, C, N, device="cuda") * 0.05
    bias, cb, res = torch.randn(N, device="cuda"), torch.randn(batch, N, device="cuda"), torch.randn(batch, 1024, N, device="cuda")
    for _ in range(30):
        ops.conv3x3_raw(x, w, bias, cb, res)
    call("mulan_set_debug_buffer", ptr(buf))
    ops.conv3x3_raw(x, w, bias, cb, res)
    torch.cuda.synchronize()
    call("mulan_set_debug_buffer", None)
    t = buf.cpu().tolist()
    nch = (C + 15) // 16
    rel = lambda i: (t[i] - t[0]) * 1.0       # shader cycles (s_memtime counts core clocks)
    print(f"stamps B={batch} C={C} N={N} tune={tune!r} mode={ops.CONV_MODE}: prologue {rel(1):.0f} cyc; chunks " +
          " ".join(f"{(t[3 + i] - t[2 + i]):.0f}" for i in range(nch - 1)) +
          f"; last chunk {(t[22] - t[1 + nch]):.0f}; epilogue {(t[23] - t[22]):.0f}; total {rel(23):.0f} cyc "
          f"(pure MFMA per chunk = 18432 cyc per wave); in-kernel clock {(t[23] - t[0]) / max(1, (t[31] - t[30])) * 0.1:.3f} GHz")


def timeline(batch=128, C=128, N=128):
    """dev: start / end of every block of one conv3x3 launch (s_memrealtime, 10 ns ticks)"""
    import numpy as np
    ops.lib.load()
    buf = torch.zeros(64 + 2048, dtype=torch.int64, device="cuda")
    x, w = torch.randn(batch, 1024, C, device="cuda"), torch.randn(3, 3, C, N, device="cuda") * 0.05
    bias, cb, res = torch.randn(N, device="cuda"), torch.randn(batch, N, device="cuda"), torch.randn(batch, 1024, N, device="cuda")
    for _ in range(30):
        ops.conv3x3_raw(x, w, bias, cb, res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call("mulan_set_debug_buffer", ptr(buf))
    e0.record()
    ops.conv3x3_raw(x, w, bias, cb, res)
    e1.record()
    torch.cuda.synchronize()
    call("mulan_set_debug_buffer", None)
    t = buf.cpu().numpy()[64:].reshape(-1, 2)
    t = t[t[:, 0] > 0]
    st, en = (t[:, 0] - t[:, 0].min()) * 0.01, (t[:, 1] - t[:, 0].min()) * 0.01      # us
    order = np.argsort(st)
    n = len(st)
    dur = en - st
    print(f"timeline B={batch} C={C} N={N}: {n} blocks, events {e0.elapsed_time(e1) * 1e3:.1f} us (incl. maxima/pack launches)")
    print(f"  starts: first gen (0..{n // 2 - 1}) min {st[order[0]]:.1f} p50 {st[order[n // 4]]:.1f} max {st[order[n // 2 - 1]]:.1f} us;"
          f" second gen min {st[order[n // 2]]:.1f} p50 {st[order[3 * n // 4]]:.1f} max {st[order[-1]]:.1f} us")
    print(f"  block duration: min {dur.min():.1f} p50 {np.median(dur):.1f} max {dur.max():.1f} us; first-gen p50 "
          f"{np.median(dur[order[:n // 2]]):.1f}, second-gen p50 {np.median(dur[order[n // 2:]]):.1f}")
    print(f"  ends: first {en.min():.1f} p50 {np.median(en):.1f} last {en.max():.1f} us")


def wstamps(batch=128, C=128, N=128):
    ops.lib.load()
    ops.CONV_MODE = "f16x3"
    x, dy = torch.randn(batch, 1024, C, device="cuda"), torch.randn(batch, 1024, N, device="cuda")
    w = torch.randn(3, 3, C, N, device="cuda") * 0.05
    xmax, dymax = ops.absmax_rows(x), ops.absmax_rows(dy)
    _, xs = ops.conv3x3_raw(x, w, xmax=xmax, planes=True)
    _, dys = ops.conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True)
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    for _ in range(30):
        ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, batch, C, N)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, batch, C, N)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    call("mulan_set_debug_buffer", ptr(buf))
    ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, batch, C, N)
    torch.cuda.synchronize()
    call("mulan_set_debug_buffer", None)
    t = buf.cpu().tolist()
    pairs = [t[3 + i] - t[2 + i] for i in range(19) if t[3 + i] > 0]
    print(f"wgrad(planes) B={batch} C={C} N={N}: {us:.1f} us incl. slab reduce; prologue {t[1] - t[0]} cyc; pairs {pairs}; "
          f"loop {t[22] - t[1]}; epilogue {t[23] - t[22]}; total {t[23] - t[0]} (pure MFMA per pair = 144 x 34.6 = 4982 "
          f"cyc); in-kernel clock {(t[23] - t[0]) / max(1, (t[31] - t[30])) * 0.1:.3f} GHz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "timeline":
        timeline()
        timeline(batch=64)
        timeline(C=256)
    elif len(sys.argv) > 1 and sys.argv[1] == "wstamps":
        wstamps()
        wstamps(C=256)
    elif len(sys.argv) > 1 and sys.argv[1] == "stamps":
        for md in ("bf16x6", "f16x3"):
            for bsz in (32, 128):
                stamps(bsz, mode=md)
            stamps(128, C=256, mode=md)
    else:
        main()
