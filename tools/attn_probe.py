"""dev: attention core fwd + bwd at B = 128 (argv[1]), S = 1024, C = 128 (argv[2]: 128 or 256): the fused kernels (attention_f16x3.hip) vs the unfused
split-operand path (S / P in HBM) vs the fp32 GEMM path; HIP-event times of whole fwd / bwd calls incl. packs."""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mulan_amd import ops

ops.lib.load()
B, S, C = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), 1024, (int(sys.argv[2]) if len(sys.argv) > 2 else 128)
q, k, v, do = (torch.randn(B, S, C, device="cuda") for _ in range(4))


def ev():
    return torch.cuda.Event(enable_timing=True)


for name, fused, fast in (("fused f16x3", True, True), ("unfused f16x3", False, True), ("fp32 GEMM", False, False)):
    ops.ATTN_FUSED, ops.ATTN_F16X3 = fused, fast
    tf, tb = [], []
    for it in range(6):
        g = [t.clone().requires_grad_() for t in (q, k, v)]
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        o = ops.attention(*g)
        e1.record()
        o.backward(do)
        e2.record()
        torch.cuda.synchronize()
        tf.append(e0.elapsed_time(e1) * 1e3)
        tb.append(e1.elapsed_time(e2) * 1e3)
    print(f"{name:14s} B={B} C={C}: fwd {min(tf):7.0f} us  bwd {min(tb):7.0f} us  total {min(tf) + min(tb):7.0f} us  "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
    torch.cuda.reset_peak_memory_stats()
