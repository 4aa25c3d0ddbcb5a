#!/usr/bin/env python3
"""Runs the dominant kernel (3x3 conv, B=128, 128->128 channels, f16x3 mode with plane hand-over: the shape of 2/3 of
the launches of a train step) a few times, for rocprofv3 --pmc passes:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f --output-format csv -- python3 tools/pmc_conv.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE ...        (separate passes: TCC has 4 slots, FETCH_SIZE takes 3)
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mulan_amd import ops  # noqa: E402

ops.lib.load()
B, C, N = 128, 128, 128
torch.manual_seed(0)
x, w = torch.randn(B, 1024, C, device="cuda"), torch.randn(3, 3, C, N, device="cuda") * 0.05
bias, cb, res = torch.randn(N, device="cuda"), torch.randn(B, N, device="cuda"), torch.randn(B, 1024, N, device="cuda")
xmax = ops.absmax_rows(x)
wmax = ops.absmax_rows(w.view(1, -1))
for _ in range(6):
    y, xs = ops.conv3x3_raw(x, w, bias, cb, res, xmax=xmax, planes=True, wmax=wmax)
torch.cuda.synchronize()
print("done", float(y[0, 0, 0]))
