#!/usr/bin/env python3
"""A/B timing of the plane-fed f16x3 weight-gradient kernel (+ its slab reduction) on one MI355X.
Usage: python tools/wgrad_ab.py [--batch 128] [--tunes "", "1=255,6=1", ...]   (tune k=v lists as for MULAN_TUNE)"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--tunes", nargs="*", default=[""])
    a = ap.parse_args()
    lib = ops.lib.load()
    B = a.batch
    torch.manual_seed(0)
    for C, N in ((128, 128), (256, 128), (128, 256), (256, 256)):
        x, dy = torch.randn(B, 1024, C, device="cuda"), torch.randn(B, 1024, N, device="cuda")
        w = torch.randn(3, 3, C, N, device="cuda") * 0.05
        xmax, dymax = ops.absmax_rows(x), ops.absmax_rows(dy)
        _, xs = ops.conv3x3_raw(x, w, None, None, None, xmax=xmax, planes=True)
        _, dys = ops.conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True)
        ref = None
        for tune in a.tunes:
            for k in (1, 6, 7, 10, 19):
                lib.mulan_set_tuning(k, 0)
            for kv in filter(None, tune.split(",")):
                k, v = kv.split("=")
                lib.mulan_set_tuning(int(k), int(v))
            dw = ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, B, C, N)
            ref = dw.clone() if ref is None else ref
            d = float((dw - ref).abs().max() / ref.abs().max())
            med, mn = timeit(lambda: ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, B, C, N))
            fl = 2.0 * B * 1024 * 9 * C * N
            print(f"wgrad {C:3d}->{N:3d} B={B} tune[{tune:12s}]: med {med:7.1f} us  min {mn:7.1f} us  "
                  f"{fl / med / 1e6:6.1f} TF/s (x3 = {3 * fl / med / 1e6 / 2500 * 100:4.1f}% of 2.5 PF)  d={d:.1e}")


if __name__ == "__main__":
    main()
