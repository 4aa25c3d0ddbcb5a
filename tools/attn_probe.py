"""dev: attention core fwd + bwd on the split-operand kernels vs the fp32 GEMM path (run under rocprofv3 --stats)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from mulan_amd import ops

ops.lib.load()
B, S, C = 128, 1024, 128
q, k, v, do = (torch.randn(B, S, C, device="cuda") for _ in range(4))
for fast in ((True, False) if len(sys.argv) < 2 else (sys.argv[1] == "fast",)):
    ops.ATTN_F16X3 = fast
    for it in range(4):
        g = [t.clone().requires_grad_() for t in (q, k, v)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = ops.attention(*g); torch.cuda.synchronize(); t1 = time.perf_counter()
        o.backward(do); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("f16x3" if fast else "fp32", "fwd %.0f us bwd %.0f us" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6))
