#!/usr/bin/env python3
"""How do the HBM-bound kernels of the backward pass scale with the number of CUs they may use, and can a GroupNorm
backward on a CU-masked stream run beside a weight gradient that leaves those CUs alone?  (Feasibility probe for
profiles/DESIGN_r04.md 7.1: hipExtStreamCreateWithCUMask streams wrapped as torch ExternalStreams.)
Usage: python tools/cumask_probe.py [--batch 128]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402
from mulan_amd.lib import call, ptr, stream  # noqa: E402


def masked_stream(bits):
    """a HIP stream restricted to the CUs whose bit is set (256-bit mask as a list of 8 uint32)"""
    hip = ctypes.CDLL("libamdhip64.so")
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * 8)(*bits)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def mask_of(n, pattern):
    """n of 256 CUs: 'low' = bits 0 .. n-1, 'spread' = every (256 / n)-th bit"""
    bits = [0] * 8
    idx = range(n) if pattern == "low" else [int(i * 256 / n) for i in range(n)]
    for i in idx:
        bits[i >> 5] |= 1 << (i & 31)
    return bits


def timeit(fn, st, reps=20):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    ops.lib.load()
    B, C = a.batch, 128
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn(B, 1024, C, device=dev)
    dy = torch.randn(B, 1024, C, device=dev)
    g, b_ = torch.randn(C, device=dev), torch.randn(C, device=dev)
    mean, rstd = torch.empty(B, 32, device=dev), torch.empty(B, 32, device=dev)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    parts = torch.empty(2, B, C, device=dev)
    m1 = torch.empty(B, 16, device=dev, dtype=torch.int32)
    cs = torch.empty(B, C, device=dev)
    call("mulan_groupnorm_fwd", ptr(x), None, C, 0, ptr(g), ptr(b_), ptr(y), ptr(mean), ptr(rstd), B, 1024, 32, 1e-6, 1, 1.0, 1, 0,
         None, stream())
    torch.cuda.synchronize()

    def gn_bwd():
        call("mulan_groupnorm_bwd", ptr(dy), ptr(x), None, C, 0, ptr(g), ptr(b_), ptr(mean), ptr(rstd), ptr(dx), None,
             ptr(parts[0]), ptr(parts[1]), B, 1024, 32, 1, 1.0, 1, 0, 0, ptr(m1), None, None, None, ptr(cs),
             torch.cuda.current_stream().cuda_stream)

    def gn_fwd():
        call("mulan_groupnorm_fwd", ptr(x), None, C, 0, ptr(g), ptr(b_), ptr(y), ptr(mean), ptr(rstd), B, 1024, 32, 1e-6, 1, 1.0,
             1, 0, None, torch.cuda.current_stream().cuda_stream)

    print(f"B = {B}, C = {C}: GroupNorm forward / backward alone on a CU-masked stream")
    for pattern in ("low", "spread"):
        for n in (256, 128, 96, 64, 32):
            st = masked_stream(mask_of(n, pattern))
            print(f"  {pattern:6s} {n:3d} CUs: fwd {timeit(gn_fwd, st):6.1f} us   bwd {timeit(gn_bwd, st):6.1f} us")

    # One "layer" of the backward pass in miniature: a 3x3 weight gradient beside two GroupNorm backward launches, ten
    # layers in a row, captured as a HIP graph (no host effects).  Variants: one stream; two unmasked streams (the
    # shipped scheme: 120 weight-gradient blocks, the dispatcher places everything); two streams with disjoint CU
    # masks (weight gradient on the upper 256 - n CUs with as many blocks, GroupNorm on the lower n).
    w = torch.randn(3, 3, C, C, device=dev) * 0.05
    xmax, dymax = ops.absmax_rows(x), ops.absmax_rows(dy)
    _, xs = ops.conv3x3_raw(x, w, None, None, None, xmax=xmax, planes=True)
    _, dys = ops.conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True)
    L = ops.lib.load()
    wg = lambda: ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, B, C, C)

    def graph_time(build, reps=10):
        build()
        torch.cuda.synchronize()
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            build()
        torch.cuda.synchronize()
        for _ in range(3):
            g_.replay()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(reps):
            g_.replay()
        e_.record()
        torch.cuda.synchronize()
        return s_.elapsed_time(e_) * 1e3 / reps / 10        # per layer

    def one_stream():
        for _ in range(10):
            wg(); gn_bwd(); gn_bwd()

    def two_streams(st_w, st_g):
        cur = torch.cuda.current_stream()
        if st_w is not None:
            st_w.wait_stream(cur)
        st_g.wait_stream(cur)
        for _ in range(10):
            if st_w is None:
                wg()
            else:
                with torch.cuda.stream(st_w):
                    wg()
            with torch.cuda.stream(st_g):
                gn_bwd(); gn_bwd()
        if st_w is not None:
            cur.wait_stream(st_w)
        cur.wait_stream(st_g)

    L.mulan_set_tuning(1, 240)
    print(f"per layer (wgrad + 2 GroupNorm backward), graph replay: one stream, 240 blocks: {graph_time(one_stream):6.1f} us")
    plain = torch.cuda.Stream()
    for blocks in (240, 160, 120):
        L.mulan_set_tuning(1, blocks)
        print(f"  two unmasked streams, {blocks} weight-gradient blocks: {graph_time(lambda: two_streams(None, plain)):6.1f} us")
    for n in (64, 96, 128):
        lo = mask_of(n, "low")
        hi = [(~v) & 0xffffffff for v in lo]
        st_g, st_w = masked_stream(lo), masked_stream(hi)
        for blocks in (256 - n, 240):
            L.mulan_set_tuning(1, blocks)
            t = graph_time(lambda: two_streams(st_w, st_g))
            print(f"  disjoint masks: GroupNorm on {n} CUs, weight gradient on {256 - n} CUs with {blocks} blocks: {t:6.1f} us")
    L.mulan_set_tuning(1, 0)


if __name__ == "__main__":
    main()
