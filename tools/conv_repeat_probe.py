#!/usr/bin/env python3
"""Repeatability probe for the 2-blocks-per-CU convolution kernel: the same launch several times at a batch size where
blocks share a CU (DBG_B, default 128), each result compared with the one-block-per-CU kernel; prints which blocks /
waves / pixel rows / couts differ.  (This is how the early accumulator read-back of round 2 was located: only blocks of
the second dispatch round, last pixel tile, first registers of the last accumulator.)
Usage: [DBG_B=256] [DBG_ABL=0,..] python tools/dbg_race.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd import ops
lib = ops.lib.load()
torch.manual_seed(1)
B = int(os.environ.get("DBG_B", 128))
C = N = 128
x = torch.randn(B, 1024, C, device="cuda")
w = torch.randn(3, 3, C, N, device="cuda") * 0.05
bias = torch.randn(N, device="cuda")
res = torch.randn(B, 1024, N, device="cuda")
lib.mulan_set_tuning(3, 2)
ref = ops.conv3x3_raw(x, w, bias, None, res).clone()
lib.mulan_set_tuning(3, 0)
for abl in [int(a) for a in os.environ.get("DBG_ABL", "0").split(",")]:
    lib.mulan_set_tuning(4, abl)
    for rep in range(4):
        y = ops.conv3x3_raw(x, w, bias, None, res)
        bad = ((y - ref).abs() > 1e-3).view(B, 4, 8, 32, 4, 32)      # b, row tile, row, col, wave, cout
        n = int(bad.sum())
        blocks = bad.sum((2, 3, 5)).nonzero().tolist()                # (b, tile, wave)
        info = []
        for b, t, wv in blocks[:6]:
            sub = bad[b, t, :, :, wv, :]
            rows = sub.sum((1, 2)).nonzero().flatten().tolist()
            cols = sub.sum((0, 2)).nonzero().flatten().tolist()
            couts = sub.sum((0, 1)).nonzero().flatten().tolist()
            info.append((b * 4 + t, wv, int(sub.sum()), rows, cols[:4] + cols[-2:], couts[:3] + couts[-2:]))
        print("abl", abl, "rep", rep, "bad", n, "nblocks", len(blocks))
        for i in info:
            print("    blk %d wave %d nbad %d rows %s cols %s couts %s" % i)
