"""Per-block phase times of the GroupNorm-fed convolution (forward-only chain: statistics handed over, normalisation in the
patch fill) from the kernel's debug stamps, next to the kernel's duration between events.
    python tools/gnf_timeline.py [--batch 16]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulan_amd import ops  # noqa: E402
from mulan_amd.lib import call, ptr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    L = ops.lib.load()
    B, E = a.batch, 128
    torch.manual_seed(0)
    mk = lambda *s, sc=1.0: torch.randn(*s, device="cuda") * sc
    for name, concat, film in (("128->128 +res", False, False), ("128->128 FiLM", False, True), ("256->128 FiLM (concat)", True, True)):
        Ct = 2 * E if concat else E
        g0, b0, w0, c0 = mk(E), mk(E, sc=0.3), mk(3, 3, E, E, sc=0.03), mk(E)
        g1, b1, w1, c1 = mk(Ct), mk(Ct, sc=0.3), mk(3, 3, Ct, E, sc=0.03), mk(E)
        cb = mk(B, E) if film else None
        x = mk(B, 1024, E, sc=2.0)
        with torch.no_grad():
            h = ops.gn_conv3x3(x, None, g0, b0, w0, c0, res=x)
            h2 = ops.gn_conv3x3(h, None, g0, b0, w0, c0, res=h) if concat else None
            run = lambda: ops.gn_conv3x3(h, h2, g1, b1, w1, c1, cbias=cb, res=None if film else h)
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                run()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / 50
            buf = torch.zeros(64 + 4 * 2048, dtype=torch.int64, device="cuda")
            call("mulan_set_debug_buffer", ptr(buf))
            run()
            torch.cuda.synchronize()
            call("mulan_set_debug_buffer", None)
        rows = L.mulan_conv3x3_f16x3_tile_rows(B, 32, E, 1)
        nblk = min(B * (32 // rows), 2048)
        t = buf[64:64 + 4 * nblk].cpu().numpy().reshape(nblk, 4).astype(np.float64) * 0.01
        t -= t[:, 0].min()
        q = lambda v: f"{np.percentile(v, 10):6.1f} {np.median(v):6.1f} {np.percentile(v, 90):6.1f}"
        print(f"B={B} {name} ({rows} rows per block, {nblk} blocks): {us:6.1f} us per call | us p10/p50/p90: start {q(t[:, 0])} | "
              f"prologue {q(t[:, 1] - t[:, 0])} | loop {q(t[:, 2] - t[:, 1])} | epilogue {q(t[:, 3] - t[:, 2])} | end {q(t[:, 3])}", flush=True)


if __name__ == "__main__":
    main()
