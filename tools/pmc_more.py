#!/usr/bin/env python3
"""The kernels behind the dominant one -- plane-fed 3x3 weight gradient (two shapes) and GroupNorm forward / backward --
a few launches each at B = 128, for rocprofv3 --pmc passes (one counter group per pass, as tools/pmc_conv.py):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o FETCH_SIZE --output-format csv -- python3 tools/pmc_more.py
  ... WRITE_SIZE ... ; ... SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...
  python3 tools/pmc_more.py --parse out  > profiles/r02_pmc_wgrad_groupnorm.json

Launch counts identify the shapes inside one kernel symbol (the first launch of each group is dropped as cold)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = 128
# (kernel substring, label, launches, algorithmic bytes per launch, MFMA 32x32x16 instructions per launch)
PX = B * 1024
GROUPS = [
    # (round 6: the 3x3 weight gradient runs on conv3x3_wgrad_f16x3_w16_kernel -- eight waves, 16x16x32 MFMAs)
    ("conv3x3_wgrad_f16x3_w16_kernel", "wgrad_128_128", 6, PX * 128 * 4 * 2, 3.0 * PX * 9 * 128 * 128 / (16 * 16 * 32)),
    ("conv3x3_wgrad_f16x3_w16_kernel", "wgrad_256_128", 5, PX * (256 + 128) * 4, 3.0 * PX * 9 * 256 * 128 / (16 * 16 * 32)),
    ("gn_fwd_kernel", "groupnorm_fwd_128_dropout", 6, PX * 128 * 4 * 2, 0),
    ("gn_fwd_kernel", "groupnorm_fwd_256_concat", 5, PX * 256 * 4 * 2, 0),
    ("gn_bwd_kernel_1pass", "groupnorm_bwd_128_dropout", 6, PX * 128 * 4 * 3, 0),
    ("gn_bwd_kernel_1pass", "groupnorm_bwd_128_skip_add", 5, PX * 128 * 4 * 4, 0),
]


def run():
    import torch
    from mulan_amd import ops
    from mulan_amd.lib import call, ptr, stream
    ops.lib.load()
    torch.manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda")
    for C, N, n in ((128, 128, 6), (256, 128, 5)):
        x, dy, w = r(B, 1024, C), r(B, 1024, N), r(3, 3, C, N) * 0.05
        xmax, dymax = ops.absmax_rows(x), ops.absmax_rows(dy)
        _, xs = ops.conv3x3_raw(x, w, None, None, None, xmax=xmax, planes=True)
        _, dys = ops.conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True)
        for _ in range(n):
            dw = ops.conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, B, C, N)
        torch.cuda.synchronize()
    for C1, C2, n, keep in ((128, 0, 6, 0.9), (128, 128, 5, 1.0)):
        x1, x2 = r(B, 1024, C1), (r(B, 1024, C2) if C2 else None)
        g, b_ = r(C1 + C2), r(C1 + C2)
        y = torch.empty(B, 1024, C1 + C2, device="cuda")
        mean, rstd = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
        ym = torch.empty(B, 16, device="cuda", dtype=torch.int32)
        for _ in range(n):
            call("mulan_groupnorm_fwd_dyn", ptr(x1), ptr(x2), C1, C2, ptr(g), ptr(b_), ptr(y), ptr(mean), ptr(rstd), B, 1024,
                 32, 1e-6, 1, keep, 123, 0, None, ptr(ym), stream())
        torch.cuda.synchronize()
    x1, g, b_ = r(B, 1024, 128), r(128), r(128)
    y = torch.empty_like(x1)
    mean, rstd = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
    call("mulan_groupnorm_fwd", ptr(x1), None, 128, 0, ptr(g), ptr(b_), ptr(y), ptr(mean), ptr(rstd), B, 1024, 32, 1e-6, 1,
         0.9, 123, 0, None, stream())
    dy, dx, add = r(B, 1024, 128), torch.empty_like(x1), r(B, 1024, 128)
    parts, cs = torch.empty(2, B, 128, device="cuda"), torch.empty(B, 128, device="cuda")
    dg, db, sink = torch.empty(128, device="cuda"), torch.empty(128, device="cuda"), torch.empty(128, device="cuda")
    m1 = torch.empty(B, 16, device="cuda", dtype=torch.int32)
    tick = torch.zeros(16, device="cuda", dtype=torch.int32)
    for n, keep, a1 in ((6, 0.9, None), (5, 1.0, add)):
        for _ in range(n):
            call("mulan_groupnorm_bwd_fused", ptr(dy), ptr(x1), None, 128, 0, ptr(g), ptr(b_), ptr(mean), ptr(rstd), ptr(dx),
                 None, ptr(parts[0]), ptr(parts[1]), B, 1024, 32, 1, keep, 123, 0, None, ptr(m1), None, ptr(a1), None, None,
                 ptr(cs), ptr(dg), ptr(db), ptr(sink), None, ptr(tick), stream())
        torch.cuda.synchronize()
    print("done", float(dw[0, 0, 0, 0]), float(dx[0, 0, 0]))


def parse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    out = {"batch": B, "corrections": "FETCH_SIZE x 1024 x 2 (gfx950: 16-B-per-lane loads), WRITE_SIZE x 1024; one "
                                      "counter group per rocprofv3 pass; first launch of each group dropped", "kernels": {}}
    for sym in sorted({g[0] for g in GROUPS}):
        mine = [r for r in rows if sym in r["Kernel_Name"]]
        by = {}
        for r in mine:
            by.setdefault(r["Counter_Name"], []).append(
                (int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        start = 0
        for s2, label, n, alg, mfmas in GROUPS:
            if s2 != sym:
                continue
            ent = {"launches_profiled": n, "algorithmic_bytes_per_launch": alg}
            for counter, vals in by.items():
                vals = sorted(vals)[start:start + n][1:]
                if not vals:
                    continue
                ent[counter] = sum(v for _, v, _ in vals) / len(vals)
                ent["avg_duration_us"] = sum(t for _, _, t in vals) / len(vals) / 1e3
            if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
                ent["hbm_bytes_per_launch"] = ent["FETCH_SIZE"] * 2048 + ent["WRITE_SIZE"] * 1024
                ent["traffic_over_algorithmic"] = ent["hbm_bytes_per_launch"] / alg
                ent["hbm_TBps"] = ent["hbm_bytes_per_launch"] / (ent["avg_duration_us"] * 1e-6) / 1e12
            if mfmas and "SQ_VALU_MFMA_BUSY_CYCLES" in ent and "GRBM_GUI_ACTIVE" in ent:
                ent["mfma_instructions"] = mfmas
                ent["clock_GHz_from_GRBM"] = ent["GRBM_GUI_ACTIVE"] / 8 / (ent["avg_duration_us"] * 1e3)
                ent["mfma_util"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (ent["GRBM_GUI_ACTIVE"] / 8 * 1024)
            out["kernels"][label] = ent
            start += n
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
