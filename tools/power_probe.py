#!/usr/bin/env python3
"""dev (round 6): socket power and shader clock of each kernel family of the train step, run back to back for ~1.5 s on
rotating operand sets (B = 128): W, MHz, us per launch and JOULES per launch.  The step sits at 92 % of the 1400 W cap on
average (bench.py `chip`), so a kernel's share of the step's ENERGY says more about what it costs than its share of time.
Usage: python tools/power_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from mulan_amd import ops  # noqa: E402
from mulan_amd.lib import call, ptr, stream  # noqa: E402

B = 128
ops.lib.load()
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda")


def measure(name, fn, seconds=1.5, flops=0.0, bytes_=0.0):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    n = 0
    tele = bench.Telemetry(0)
    with tele:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for i in range(50):
                fn(n + i)
            n += 50
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    s = tele.summary()
    w = (s["socket_power_w"] or {}).get("mean")
    us = dt / n * 1e6
    j = None if w is None else w * dt / n
    extra = (f"  {flops / (dt / n) / 1e12:6.1f} TFLOP/s" if flops else "") + (f"  {bytes_ / (dt / n) / 1e12:5.2f} TB/s" if bytes_ else "")
    print(f"{name:44s} {us:8.1f} us  {w} W  sclk {(s['sclk_mhz'] or {}).get('mean')} MHz  {None if j is None else round(j * 1e3, 2)} mJ per launch{extra}", flush=True)


def idle(i):
    time.sleep(0.0002)


measure("idle (host sleeps, nothing launched)", idle)
sets = []
for _ in range(4):
    x, dy = r(B, 1024, 128), r(B, 1024, 128)
    w = r(3, 3, 128, 128) * 0.05
    xmax, dymax = ops.absmax_rows(x), ops.absmax_rows(dy)
    _, xs = ops.conv3x3_raw(x, w, None, None, None, xmax=xmax, planes=True)
    _, dys = ops.conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True)
    sets.append((x, dy, w, xmax, dymax, xs, dys))
fl = 2.0 * B * 1024 * 9 * 128 * 128
dw = torch.empty(3, 3, 128, 128, device="cuda")
L = ops.lib.load()
for share in (0, 1):
    nb = L.mulan_conv3x3_wgrad_f16x3_planes_workspace(B, 32, 32, 128, 128, share)
    ws = torch.empty(nb // 4, device="cuda")

    def wg(i, share=share, ws=ws):
        x, dy, w, xmax, dymax, xs, dys = sets[i % 4]
        call("mulan_conv3x3_wgrad_f16x3_planes", ptr(xs), ptr(xmax), ptr(dys), ptr(dymax), ptr(dw), ptr(ws), B, 32, 32, 128, 128, 0, share, stream())
    measure(f"3x3 weight gradient + reduce 128->128, share {share}", wg, flops=fl)


def conv_pl(i):
    x, dy, w, xmax, dymax, xs, dys = sets[i % 4]
    ops.conv3x3_dgrad_planes_raw(dys, dymax, w)


def conv_f32(i):
    x, dy, w, xmax, dymax, xs, dys = sets[i % 4]
    ops.conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True)


measure("3x3 conv, plane-fed input gradient 128->128", conv_pl, flops=fl)
measure("3x3 conv, fp32 input + plane by-product", conv_f32, flops=fl)
g, b_ = r(128), r(128)
y = torch.empty(B, 1024, 128, device="cuda")
mean, rstd = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
ym = torch.empty(B, 16, device="cuda", dtype=torch.int32)


def gnf(i):
    x = sets[i % 4][0]
    call("mulan_groupnorm_fwd_dyn", ptr(x), None, 128, 0, ptr(g), ptr(b_), ptr(y), ptr(mean), ptr(rstd), B, 1024, 32, 1e-6, 1, 0.9,
         123, 0, None, ptr(ym), stream())


measure("GroupNorm forward 128 ch + dropout", gnf, bytes_=2.0 * B * 1024 * 128 * 4)
dx = torch.empty(B, 1024, 128, device="cuda")
parts, cs = torch.empty(2, B, 128, device="cuda"), torch.empty(B, 128, device="cuda")
dg, db, sink = torch.empty(128, device="cuda"), torch.empty(128, device="cuda"), torch.empty(128, device="cuda")
m1 = torch.empty(B, 16, device="cuda", dtype=torch.int32)
tick = torch.zeros(16, device="cuda", dtype=torch.int32)


def gnb(i):
    x, dy = sets[i % 4][0], sets[(i + 1) % 4][1]
    call("mulan_groupnorm_bwd_fused", ptr(dy), ptr(x), None, 128, 0, ptr(g), ptr(b_), ptr(mean), ptr(rstd), ptr(dx), None,
         ptr(parts[0]), ptr(parts[1]), B, 1024, 32, 1, 0.9, 123, 0, None, ptr(m1), None, None, None, None, ptr(cs), ptr(dg), ptr(db),
         ptr(sink), None, ptr(tick), stream())


measure("GroupNorm backward 128 ch + dropout", gnb, bytes_=3.0 * B * 1024 * 128 * 4)
cp_src = [r(B, 1024, 128) for _ in range(4)]
cp_dst = torch.empty(B, 1024, 128, device="cuda")
measure("torch copy 67 MB", lambda i: cp_dst.copy_(cp_src[i % 4]), bytes_=2.0 * B * 1024 * 128 * 4)
