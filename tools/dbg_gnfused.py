import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd import ops
ops.lib.load()
torch.manual_seed(256)
B, C, C2 = 128, 128, 128

def leaf(*shape, scale=1.0):
    t = (torch.randn(*shape, device="cuda") * scale).requires_grad_(True)
    t._gview = torch.zeros(*shape, device="cuda")
    return t

w, bias = leaf(3, 3, C, C, scale=0.05), leaf(C)
wn, bn = leaf(C, C, scale=0.1), leaf(C)
gamma, beta = leaf(C + C2), leaf(C + C2)
x = torch.randn(B, 1024, C, device="cuda", requires_grad=True)
skip = torch.randn(B, 1024, C2, device="cuda", requires_grad=True)
gy = torch.randn(B, 1024, C + C2, device="cuda")
leaves = (w, bias, wn, bn, gamma, beta)
seen = {}

def run(fused):
    ops.GN_FUSED_REDUCE = fused
    for t in leaves + (x, skip):
        t.grad = None
    for t in leaves:
        t._gview.zero_()
    h = ops.conv3x3(x, w, bias, None, ops.linear(x, wn, bn))
    h.register_hook(lambda t: seen.__setitem__("dh", (t.clone(), t._absmax[0].clone())))
    y = ops.group_norm(h, skip, gamma, beta, act=True, keep=0.9, seed=11, offset=0)
    (y * gy).sum().backward()
    return [t.grad.clone() for t in leaves] + [x.grad.clone(), skip.grad.clone(), seen["dh"][0], seen["dh"][1].float()]

names = ("w", "bias", "wn", "bn", "gamma", "beta", "x", "skip", "dh", "dhmax")
runs = [run(f) for f in (False, False, True, True, False, True)]
for i in range(1, len(runs)):
    print(i, {n: float((a - b).abs().max()) for n, a, b in zip(names, runs[i], runs[0])})
