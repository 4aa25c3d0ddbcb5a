#!/usr/bin/env python3
"""A/B of the f16x3 forward / input-gradient convolution kernels on one MI355X: interleaved rounds in one process
(variants = values of tuning knob 3: 0 = v3 (16x16x32, two blocks per CU), 2 = v2 (32x32x16, one block per CU)),
operands prepared once, the kernel alone between the events.  Prints the median and the minimum per variant and shape.
Usage: python tools/conv_ab.py [--batch 128] [--rounds 7] [--reps 40] [--variants 0,2]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402
from mulan_amd.lib import call, ptr, stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--variants", default="0,2")
    ap.add_argument("--ablate", default="", help="v3 only: comma list of ablation masks (tuning knob 4) timed as extra variants")
    ap.add_argument("--timeline", action="store_true", help="v3 only: per-block phase times from the debug stamps")
    ap.add_argument("--planes-in", action="store_true",
                    help="time the plane-fed entry point (mulan_conv3x3_fwd_f16x3_planes_in) on the planes the fp32 launch wrote")
    a = ap.parse_args()
    L = ops.lib.load()
    B = a.batch
    variants = [int(v) for v in a.variants.split(",")] + [100 + int(m) for m in a.ablate.split(",") if m]
    torch.manual_seed(0)
    # (C, N, bias, per-sample FiLM bias, residual, planes, ymax): the three launch shapes of a train step
    shapes = [("fwd 128->128 +res +planes", 128, 128, True, True, True, True),
              ("fwd 128->128 conv1 (FiLM) +planes", 128, 128, True, True, False, True),
              ("fwd 256->128 (up conv1) +planes", 256, 128, True, True, False, True),
              ("dgrad 128->256 (up conv1) +planes", 128, 256, False, False, False, True),
              ("dgrad 128->128 +planes", 128, 128, False, False, False, True)]
    for name, C, N, hb, hcb, hres, planes in shapes:
        x = torch.randn(B, 1024, C, device="cuda")
        w = torch.randn(3, 3, C, N, device="cuda") * 0.05
        bias = torch.randn(N, device="cuda") if hb else None
        cb = torch.randn(B, N, device="cuda") if hcb else None
        res = torch.randn(B, 1024, N, device="cuda") if hres else None
        xmax, wmax = ops.absmax_rows(x), ops.absmax_rows(w.view(1, -1))
        wp = torch.empty(L.mulan_conv3x3_pack_f16x3_bytes(C, N), device="cuda", dtype=torch.uint8)
        call("mulan_conv3x3_pack_f16x3", ptr(w), ptr(wp), ptr(wmax), C, N, 0, stream())
        y = torch.empty(B, 1024, N, device="cuda")
        xs = torch.empty(B * 1024 * C * 4, device="cuda", dtype=torch.uint8) if planes else None
        ymax = torch.empty(B, 16, device="cuda", dtype=torch.int32) if N // 128 * 4 <= 16 else None

        def launch():
            call("mulan_conv3x3_fwd_f16x3", ptr(x), ptr(xmax), ptr(wp), ptr(wmax), ptr(bias), ptr(cb), 1 if hcb else 0,
                 ptr(res), ptr(y), ptr(xs), ptr(ymax), B, 32, 32, C, N, stream())

        if a.planes_in and planes:
            launch()                 # the fp32 launch fills xs: from here on the plane-fed kernel is what is timed
            torch.cuda.synchronize()

            def launch():            # noqa: F811
                call("mulan_conv3x3_fwd_f16x3_planes_in", ptr(xs), ptr(xmax), ptr(wp), ptr(wmax), ptr(bias), ptr(cb),
                     1 if hcb else 0, ptr(res), ptr(y), ptr(ymax), B, 32, 32, C, N, stream())

        if a.timeline:
            import numpy as np
            call("mulan_set_tuning", 3, 0)
            for _ in range(20):
                launch()
            torch.cuda.synchronize()
            buf = torch.zeros(64 + 4 * 2048, dtype=torch.int64, device="cuda")
            call("mulan_set_debug_buffer", ptr(buf))
            launch()
            torch.cuda.synchronize()
            call("mulan_set_debug_buffer", None)
            rows = L.mulan_conv3x3_f16x3_tile_rows(B, 32, N, int(ymax is not None))    # tile height of this launch
            nblk = B * (32 // rows) * (N // 128)
            t = buf[64:64 + 4 * nblk].cpu().numpy().reshape(nblk, 4).astype(np.float64) * 0.01
            t -= t[:, 0].min()
            cyc = buf[:32].cpu().numpy().astype(np.float64)
            loop_us = t[:32, 2] - t[:32, 1]
            print(f"  {name}: in-kernel clock over the main loop (blocks 0-31): {np.median(cyc / loop_us) * 1e-3:.3f} GHz; "
                  f"loop cycles p50 {np.median(cyc):.0f} (MFMA issue cycles of one wave: {(C // 32) * 9 * 12 * rows * 16}, {rows} rows per block)")
            hi = ((np.arange(nblk) >> 8) & 1) == 1
            for nm, m in (("prio 1", hi), ("prio 0", ~hi)):
                if not m.any():
                    continue
                q = lambda v: f"{np.percentile(v, 10):6.1f} {np.median(v):6.1f} {np.percentile(v, 90):6.1f}"
                print(f"  {name} [{nm}, {int(m.sum())} blocks] us p10/p50/p90: start {q(t[m, 0])} | prologue {q(t[m, 1] - t[m, 0])} | "
                      f"loop {q(t[m, 2] - t[m, 1])} | epilogue {q(t[m, 3] - t[m, 2])} | end {q(t[m, 3])}")
            continue
        def select(v):
            call("mulan_set_tuning", 3, v if v < 100 else 0)
            call("mulan_set_tuning", 4, v - 100 if v >= 100 else 0)

        outs = {}
        times = {v: [] for v in variants}
        for v in variants:
            select(v)
            launch()
            torch.cuda.synchronize()
            outs[v] = y.clone()
        for _ in range(a.rounds):
            for v in variants:
                select(v)
                for _ in range(5):
                    launch()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.reps):
                    launch()
                e.record()
                torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e) * 1e3 / a.reps)
        select(0)
        fl = 2.0 * B * 1024 * 9 * C * N
        ref = outs[variants[0]]
        line = f"{name:36s} B={B}:"
        for v in variants:
            med, mn = statistics.median(times[v]), min(times[v])
            d = float((outs[v] - ref).abs().max() / ref.abs().max())
            line += f"  [v{v}] med {med:7.1f} us min {mn:7.1f} us {fl / med / 1e6:6.1f} TF/s (x3 = {3 * fl / med / 1e6 / 2500 * 100:4.1f}% of 2.5 PF) d={d:.1e}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
