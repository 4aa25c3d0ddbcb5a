#!/usr/bin/env python3
"""Timing of the per-pixel dense kernels (linear_f16x3.hip and the one-tap weight-gradient kernel) on the shapes of a
train step, with the bytes each launch really moves: nin_shortcut forward on concat[h, skip] (K = 2E -> N = E) with and
without the plane by-product, its input gradient (K = E -> N = E | E), the attention projections (E -> E) and the dense
weight gradient from planes.  Usage: python tools/linear_probe.py [--batch 128] [--width 128]"""
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
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--tune", default="", help="k=v,... developer switches (mulan_set_tuning)")
    a = ap.parse_args()
    L = ops.lib.load()
    for kv in filter(None, a.tune.split(",")):
        k, v = kv.split("=")
        L.mulan_set_tuning(int(k), int(v))
    B, E = a.batch, a.width
    M = B * 1024
    torch.manual_seed(0)
    h, skip = torch.randn(B, 1024, E, device="cuda"), torch.randn(B, 1024, E, device="cuda")
    dy = torch.randn(B, 1024, E, device="cuda")
    w = torch.randn(2 * E, E, device="cuda") * 0.05
    wq = torch.randn(E, E, device="cuda") * 0.05
    for t in (h, skip, dy):
        ops.cached_absmax(t)
    wp, wmax = ops.linear_pack(w, False)
    wpt, wmaxt = ops.linear_pack(w, True)
    wpq, wmaxq = ops.linear_pack(wq, False)
    MB = 1e-6

    def report(name, us, nbytes, flops):
        print(f"{name:58s} {us:7.1f} us  {nbytes * MB:7.1f} MB  {nbytes / us * 1e-6:5.2f} TB/s  {flops / us * 1e-6:6.1f} TFLOP/s")

    us = timeit(lambda: ops.linear_f16x3_raw(h, skip, wp, wmax, E, 0))
    report(f"nin_shortcut fwd [h|skip] {2 * E}->{E}, no planes", us, M * 3 * E * 4, 2.0 * M * 2 * E * E)
    us = timeit(lambda: ops.linear_f16x3_raw(h, skip, wp, wmax, E, 0, planes=True))
    report(f"nin_shortcut fwd [h|skip] {2 * E}->{E}, + planes of the input", us, M * 5 * E * 4, 2.0 * M * 2 * E * E)
    us = timeit(lambda: ops.linear_f16x3_raw(dy, None, wpt, wmaxt, E, E))
    report(f"nin_shortcut input gradient {E}->{E}|{E}", us, M * 3 * E * 4, 2.0 * M * 2 * E * E)
    us = timeit(lambda: ops.linear_f16x3_raw(h, None, wpq, wmaxq, E, 0))
    report(f"attention projection {E}->{E}", us, M * 2 * E * 4, 2.0 * M * E * E)
    _, _, xs, xmax = ops.linear_f16x3_raw(h, skip, wp, wmax, E, 0, planes=True)
    wc = torch.randn(3, 3, E, E, device="cuda") * 0.05
    _, dys = ops.conv3x3_dgrad_raw(dy, wc, dymax=ops.cached_absmax(dy), planes=True)
    dymax = ops.cached_absmax(dy)
    us = timeit(lambda: ops.linear_wgrad_planes_raw(xs, xmax, dys, dymax, B, 2 * E, E))
    report(f"dense weight gradient from planes {2 * E}x{E} (+ slab reduce)", us, M * 3 * E * 4, 2.0 * M * 2 * E * E)
    # the same with operands that are not in the 256 MB Infinity Cache (three rotating operand sets, as in a train step)
    sets = [(xs.clone(), dys.clone()) for _ in range(3)]
    state = {"i": 0}

    def cold():
        a_, b_ = sets[state["i"] % 3]
        state["i"] += 1
        ops.linear_wgrad_planes_raw(a_, xmax, b_, dymax, B, 2 * E, E)
    us = timeit(cold)
    report(f"  ... cache-cold operands (3 rotating sets)", us, M * 3 * E * 4, 2.0 * M * 2 * E * E)
    y0 = [torch.empty(B, 1024, E, device="cuda") for _ in range(3)]
    hs = [(torch.randn(B, 1024, E, device="cuda"), torch.randn(B, 1024, E, device="cuda")) for _ in range(3)]
    for t_ in hs:
        ops.cached_absmax(t_[0]); ops.cached_absmax(t_[1])

    def cold_fwd():
        a_, b_ = hs[state["i"] % 3]
        state["i"] += 1
        ops.linear_f16x3_raw(a_, b_, wp, wmax, E, 0, planes=True)
    us = timeit(cold_fwd)
    report(f"nin_shortcut fwd + planes, cache-cold operands", us, M * 5 * E * 4, 2.0 * M * 2 * E * E)
    x1 = torch.randn(B, 1024, E, device="cuda")
    gn = lambda: ops.group_norm(x1, None, torch.ones(E, device="cuda"), torch.zeros(E, device="cuda"), act=True)
    us = timeit(gn)
    report(f"(for scale) GroupNorm forward C={E}", us, M * 2 * E * 4, 0.0)


if __name__ == "__main__":
    main()
