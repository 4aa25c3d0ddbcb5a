import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd import ops
from mulan_amd.lib import call, ptr, stream
ops.lib.load()
def timed(fn, reps=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)*1e3/reps)
    return sorted(ts)[2]
for B,C,N in ((128,128,256),(128,256,256),(500,256,256),(1000,256,256),(1000,512,256)):
    x=torch.randn(B,1024,C,device="cuda"); w=torch.randn(3,3,C,N,device="cuda")*0.05
    g,b_=torch.randn(C,device="cuda"),torch.randn(C,device="cuda")*0.3
    bias,cb=torch.randn(N,device="cuda"),torch.randn(B,N,device="cuda")
    wmax=ops.absmax_rows(w.view(1,-1)); wp,_=ops._pack_weights(w,C,N,0,wmax)
    ys=torch.empty(B*1024*C*4,device="cuda",dtype=torch.uint8); bound=torch.empty(B,16,device="cuda",dtype=torch.int32)
    mean,rstd=torch.empty(B,32,device="cuda"),torch.empty(B,32,device="cuda")
    y=torch.empty(B,1024,N,device="cuda"); ym=torch.empty(B,16,device="cuda",dtype=torch.int32)
    call("mulan_groupnorm_fwd_planes",ptr(x),None,C,0,ptr(g),ptr(b_),ptr(ys),ptr(mean),ptr(rstd),B,1024,32,1e-6,1,1.0,0,0,None,ptr(bound),stream())
    pin=lambda: call("mulan_conv3x3_fwd_f16x3_planes_in",ptr(ys),ptr(bound),ptr(wp),ptr(wmax),ptr(bias),ptr(cb),1,None,ptr(y),ptr(ym),B,32,32,C,N,stream())
    r={}
    for old in (0,1,0,1):
        call("mulan_set_tuning",13,old); r.setdefault(old,[]).append(timed(pin))
    call("mulan_set_tuning",13,0)
    print(f"B={B} {C}->{N}: paired {min(r[0]):8.1f} us   plain 2-D order {min(r[1]):8.1f} us", flush=True)
