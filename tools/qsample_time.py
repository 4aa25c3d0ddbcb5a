import torch, numpy as np, sys
sys.path.insert(0, ".")
from mulan_amd import ops
ops.lib.load()
B=128
g=torch.Generator().manual_seed(0)
x=torch.randint(0,256,(B,3072),dtype=torch.uint8,generator=g).cuda()
g0=torch.full((B,),-13.3).cuda(); g1=torch.full((B,),5.0).cuda(); gt=torch.rand(B).cuda()*16-12
e0=torch.randn(B,3072).cuda(); e=torch.randn(B,3072).cuda()
for ab in (1,0):
    ops.call("mulan_set_tuning",24,ab)
    for _ in range(3): ops.qsample(x,g0,g1,gt,e0,e)
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.qsample(x,g0,g1,gt,e0,e)
    b.record(); torch.cuda.synchronize()
    print("all_bins",ab,"us per call",a.elapsed_time(b)*1000/20)
ops.call("mulan_set_tuning",24,0)
