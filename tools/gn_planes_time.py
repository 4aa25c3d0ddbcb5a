#!/usr/bin/env python3
"""GroupNorm forward at B = 128: planes output (mulan_groupnorm_fwd_planes) vs fp32 output, three variants."""
import torch, sys
sys.path.insert(0, ".")
from mulan_amd import ops
from mulan_amd.lib import call, ptr, stream
ops.lib.load()
B=128
for C1,C2,keep in ((128,0,1.0),(128,0,0.9),(128,128,1.0)):
    C=C1+C2
    x1=torch.randn(B,1024,C1,device="cuda"); x2=torch.randn(B,1024,C2,device="cuda") if C2 else None
    g,b_=torch.randn(C,device="cuda"),torch.randn(C,device="cuda")
    ys=torch.empty(B*1024*C*4,device="cuda",dtype=torch.uint8); y=torch.empty(B,1024,C,device="cuda")
    bound=torch.empty(B,16,device="cuda",dtype=torch.int32); mean=torch.empty(B,32,device="cuda"); rstd=torch.empty(B,32,device="cuda")
    def t(fn):
        for _ in range(5): fn()
        torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/30
    tp=t(lambda: call("mulan_groupnorm_fwd_planes", ptr(x1), ptr(x2), C1, C2, ptr(g), ptr(b_), ptr(ys), ptr(mean), ptr(rstd), B, 1024, 32, 1e-6, 1, keep, 1, 0, None, ptr(bound), stream()))
    tf=t(lambda: call("mulan_groupnorm_fwd_dyn", ptr(x1), ptr(x2), C1, C2, ptr(g), ptr(b_), ptr(y), ptr(mean), ptr(rstd), B, 1024, 32, 1e-6, 1, keep, 1, 0, None, ptr(bound), stream()))
    print(C1,C2,keep,"planes %.1f us  fp32 %.1f us"%(tp,tf))
