#!/usr/bin/env python3
"""Feasibility probe for profiles/DESIGN_r04.md 7.6 (round-2 numbering) (two half-batches in flight): does running the forward chain of ResnetBlock ops
(GroupNorm -> planes -> 3x3 convolution, ops.gn_conv3x3) as TWO independent half-batch chains on two HIP streams beat
one full-batch chain?  The HBM-bound GroupNorm of one half could run beside the MFMA-bound convolution of the other.
Both variants are captured into HIP graphs (no host launch limits) and replayed.  Also the backward-shaped chain
(GroupNorm backward -> input-gradient convolution) the same way.
Usage: python tools/halfbatch_probe.py [--batch 128] [--layers 24] [--width 128]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--width", type=int, default=128)
    a = ap.parse_args()
    ops.lib.load()
    B, E, L = a.batch, a.width, a.layers
    torch.manual_seed(0)
    dev = "cuda"
    gamma, beta = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    ws = [torch.randn(3, 3, E, E, device=dev) * 0.03 for _ in range(L)]
    bias = torch.zeros(E, device=dev)
    x = torch.randn(B, 1024, E, device=dev)

    def chain(h, cb):
        for w in ws:
            h = ops.gn_conv3x3(h, None, gamma, beta, w, bias, cbias=cb, act=True)
        return h

    def timed_graph(build, reps=20):
        g = torch.cuda.CUDAGraph()
        build()                                   # eager warm-up (kernel attributes, allocator)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            build()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / reps

    cbf = torch.zeros(B, E, device=dev)
    xa, xb = x[:B // 2].contiguous(), x[B // 2:].contiguous()
    cba, cbb = cbf[:B // 2].contiguous(), cbf[B // 2:].contiguous()
    side = torch.cuda.Stream()

    def one():
        with torch.no_grad():
            chain(x, cbf)

    def halves_serial():
        with torch.no_grad():
            chain(xa, cba)
            chain(xb, cbb)

    def halves_two_streams():
        cur = torch.cuda.current_stream()
        with torch.no_grad():
            side.wait_stream(cur)
            ha, hb = xa, xb
            for i, w in enumerate(ws):            # interleaved issue order: A_i on the current stream, B_i on the side
                ha = ops.gn_conv3x3(ha, None, gamma, beta, w, bias, cbias=cba, act=True)
                with torch.cuda.stream(side):
                    hb = ops.gn_conv3x3(hb, None, gamma, beta, w, bias, cbias=cbb, act=True)
            cur.wait_stream(side)

    t1 = timed_graph(one)
    t2 = timed_graph(halves_serial)
    t3 = timed_graph(halves_two_streams)
    print(f"forward chain, {L} x (GroupNorm -> planes -> conv {E}->{E}), batch {B}:")
    print(f"  one chain of {B}:                       {t1:8.1f} us  ({t1 / L:6.1f} us per layer)")
    print(f"  two chains of {B // 2}, one stream:          {t2:8.1f} us  ({t2 / t1:5.3f} x)")
    print(f"  two chains of {B // 2}, two streams:         {t3:8.1f} us  ({t3 / t1:5.3f} x)")


if __name__ == "__main__":
    main()
