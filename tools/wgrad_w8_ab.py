#!/usr/bin/env python3
"""A/B of the plane-fed 3x3 weight-gradient kernels (+ slab reduction) on one MI355X: bursts of back-to-back launches on
rotating operand sets (every byte from HBM, no per-launch host gap), the four-wave block (tune 29=1) against the
eight-wave block (default), alone (share 0) and with the split count of the train step's side stream (share 1).
Usage: python tools/wgrad_w8_ab.py [--batch 128] [--sets 4] [--burst 40] [--tunes 29=1 "" ...] [--shapes 128x128 ...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402
from mulan_amd.lib import call, ptr, stream  # noqa: E402

KEYS = (1, 6, 7, 29, 30)


def set_tunes(lib, tune):
    for k in KEYS:
        lib.mulan_set_tuning(k, 0)
    for kv in filter(None, tune.split(",")):
        k, v = kv.split("=")
        lib.mulan_set_tuning(int(k), int(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--burst", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--tunes", nargs="*", default=["29=1", ""])
    ap.add_argument("--shapes", nargs="*", default=["128x128", "256x128", "256x256"])
    ap.add_argument("--shares", nargs="*", type=int, default=[0, 1])
    a = ap.parse_args()
    lib = ops.lib.load()
    B = a.batch
    torch.manual_seed(0)
    for shape in a.shapes:
        C, N = (int(v) for v in shape.split("x"))
        sets = []
        for _ in range(a.sets):
            x, dy = torch.randn(B, 1024, C, device="cuda"), torch.randn(B, 1024, N, device="cuda")
            w = torch.randn(3, 3, C, N, device="cuda") * 0.05
            xmax, dymax = ops.absmax_rows(x), ops.absmax_rows(dy)
            _, xs = ops.conv3x3_raw(x, w, None, None, None, xmax=xmax, planes=True)
            _, dys = ops.conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True)
            sets.append((xs, xmax, dys, dymax))
            del x, dy, w
        dw = torch.empty(3, 3, C, N, device="cuda")
        for share in a.shares:
            ref = None
            for tune in a.tunes:
                set_tunes(lib, tune)
                nbytes = lib.mulan_conv3x3_wgrad_f16x3_planes_workspace(B, 32, 32, C, N, share)
                ws = torch.empty(nbytes // 4, device="cuda")

                def launch(i):
                    xs, xmax, dys, dymax = sets[i % len(sets)]
                    call("mulan_conv3x3_wgrad_f16x3_planes", ptr(xs), ptr(xmax), ptr(dys), ptr(dymax), ptr(dw), ptr(ws), B,
                         32, 32, C, N, 0, share, stream())

                launch(0)
                torch.cuda.synchronize()
                got = dw.clone()
                ref = got if ref is None else ref
                d = float((got - ref).abs().max() / ref.abs().max())
                for i in range(8):
                    launch(i)
                ts = []
                for _ in range(a.rounds):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    s.record()
                    for i in range(a.burst):
                        launch(i)
                    e.record()
                    torch.cuda.synchronize()
                    ts.append(s.elapsed_time(e) * 1e3 / a.burst)
                ts.sort()
                med = ts[len(ts) // 2]
                fl = 2.0 * B * 1024 * 9 * C * N
                print(f"wgrad {C:3d}->{N:3d} B={B} share={share} tune[{tune:10s}]: {med:7.1f} us/launch incl. reduce "
                      f"(min {ts[0]:6.1f})  {fl / med / 1e6:6.1f} TF/s = {fl / med / 1e6 / 833.3:5.3f} of the f16x3 peak  "
                      f"d_vs_first={d:.1e}", flush=True)
        del sets
        torch.cuda.empty_cache()
    set_tunes(lib, "")


if __name__ == "__main__":
    main()
