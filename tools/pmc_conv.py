#!/usr/bin/env python3
"""Runs the dominant kernel (f16x3 3x3 convolution, forward and input-gradient launches of a train step) in its three
launch shapes at B = 128, a few launches each, for rocprofv3 --pmc passes (one counter group per pass):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o fetch --output-format csv -- python3 tools/pmc_conv.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE ...        (separate passes: TCC has 4 slots, FETCH_SIZE takes 3)
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...

Shapes (launch order and counts tell them apart inside a kernel symbol): the fp32-input kernel (input gradients of the
train step; forward where no GroupNorm is in front) 128->128 with residual, 256->128, 128->256 and 128->128 input
gradients; the plane-fed forward kernel 128->128 + residual, 128->128 + FiLM bias, 256->128 + FiLM bias.
`python3 tools/pmc_conv.py --parse <dir>` turns the three counter_collection CSVs found under <dir> into the JSON that
bench.py reads (the newest profiles/rNN_pmc_conv3x3_f16x3.json)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # name, kernel (fp32 input / plane-fed), C, N, residual, launches (distinct counts identify the shape in the trace)
    ("fwd_128_128_res", "fp32", 128, 128, True, 6),
    ("fwd_256_128", "fp32", 256, 128, False, 5),
    ("dgrad_128_256", "fp32", 128, 256, False, 4),
    ("dgrad_128_128", "fp32", 128, 128, False, 7),
    ("pin_fwd_128_128_res", "pin", 128, 128, True, 6),
    ("pin_fwd_128_128_film", "pin", 128, 128, False, 5),
    ("pin_fwd_256_128_film", "pin", 256, 128, False, 4),
    # round 3: the input-gradient launches of conv1 (its dy arrives as planes from the GroupNorm backward behind it):
    # no bias / FiLM / residual, output maxima written
    ("pin_dgrad_128_128", "pin_plain", 128, 128, False, 7),
    ("pin_dgrad_128_256", "pin_plain", 128, 256, False, 3),
]
SYMBOL = {"fp32": "conv3x3_f16x3_v3_kernel<0, false, 0", "pin": "conv3x3_f16x3_v3_kernel<0, true, 0"}   # (prefixes: round 4 added a template parameter)
KIND_SYMBOL = {"fp32": "fp32", "pin": "pin", "pin_plain": "pin"}
B = 128


def algorithmic_bytes(kind, C, N, res):
    """per launch: x (fp32, or its planes: same bytes) with the vertical halo of the 8-row tiles (10/8), y, the planes
    of x written as a by-product (fp32-input kernel only), residual"""
    px = B * 1024
    return {"x_with_halo": px * C * 4 * 10 / 8 * (N // 128), "y": px * N * 4, "planes": px * C * 4 if kind == "fp32" else 0,
            "residual": px * N * 4 if res else 0}


def run():
    import torch
    from mulan_amd import ops
    from mulan_amd.lib import call, ptr, stream
    ops.lib.load()
    torch.manual_seed(0)
    for name, kind, C, N, has_res, n in SHAPES:
        x, w = torch.randn(B, 1024, C, device="cuda"), torch.randn(3, 3, C, N, device="cuda") * 0.05
        bias = torch.randn(N, device="cuda") if (has_res or kind == "pin") else None
        cb = torch.randn(B, N, device="cuda") if (has_res or kind == "pin") else None
        mode = 1 if cb is not None else 0
        res = torch.randn(B, 1024, N, device="cuda") if has_res else None
        wmax = ops.absmax_rows(w.view(1, -1))
        if kind == "fp32":
            xmax = ops.absmax_rows(x)
            for _ in range(n):
                y, xs = ops.conv3x3_raw(x, w, bias, cb, res, xmax=xmax, planes=True, wmax=wmax)
        else:           # the planes come from the GroupNorm kernel, as in the train step
            g, b_ = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
            ys = torch.empty(B * 1024 * C * 4, device="cuda", dtype=torch.uint8)
            bound = torch.empty(B, 16, device="cuda", dtype=torch.int32)
            mean, rstd = torch.empty(B, 32, device="cuda"), torch.empty(B, 32, device="cuda")
            call("mulan_groupnorm_fwd_planes", ptr(x), None, C, 0, ptr(g), ptr(b_), ptr(ys), ptr(mean), ptr(rstd), B, 1024,
                 32, 1e-6, 1, 1.0, 0, 0, None, ptr(bound), stream())
            wp, _ = ops._pack_weights(w, C, N, 0, wmax)
            y = torch.empty(B, 1024, N, device="cuda")
            ym = torch.empty(B, 16, device="cuda", dtype=torch.int32)
            for _ in range(n):
                call("mulan_conv3x3_fwd_f16x3_planes_in", ptr(ys), ptr(bound), ptr(wp), ptr(wmax), ptr(bias), ptr(cb), mode,
                     ptr(res), ptr(y), ptr(ym), B, 32, 32, C, N, stream())
        torch.cuda.synchronize()
        print("done", name, float(y[0, 0, 0]))


def parse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if "conv3x3_f16x3_v3_kernel" in r["Kernel_Name"]]
    if not rows:
        raise SystemExit(f"no conv3x3_f16x3_v3_kernel rows under {d}")
    out = {"kernel": "conv3x3_f16x3_v3_kernel (<0, false, 0>: fp32 input, splits and stores planes; <0, true, 0>: plane-fed)",
           "batch": B,
           "corrections": "FETCH_SIZE x 1024 x 2 (gfx950 counts the 128-B requests of 16-B-per-lane loads as 64 B); "
                          "WRITE_SIZE x 1024 exact for 16-B-per-lane stores; one counter group per rocprofv3 pass; "
                          "dispatches of one kernel symbol in launch order, the first of each shape dropped as cold",
           "shapes": {}}
    for kind, sym in SYMBOL.items():
        by = {}
        for r in rows:
            if sym in r["Kernel_Name"]:
                by.setdefault(r["Counter_Name"], []).append(
                    (int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        start = 0
        for name, k2, C, N, has_res, n in SHAPES:
            if KIND_SYMBOL[k2] != kind:
                continue
            ent = {"kernel": sym, "C": C, "N": N, "residual": has_res, "launches_profiled": n}
            for counter, vals in by.items():
                vals = sorted(vals)[start:start + n][1:]
                if not vals:
                    continue
                ent[counter] = sum(v for _, v, _ in vals) / len(vals)
                ent["avg_duration_us"] = sum(t for _, _, t in vals) / len(vals) / 1e3
            start += n
            if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
                ent["hbm_read_bytes_per_launch"] = ent["FETCH_SIZE"] * 1024 * 2
                ent["hbm_write_bytes_per_launch"] = ent["WRITE_SIZE"] * 1024
                ent["hbm_bytes_per_launch"] = ent["hbm_read_bytes_per_launch"] + ent["hbm_write_bytes_per_launch"]
                ent["algorithmic_bytes_per_launch"] = algorithmic_bytes(kind, C, N, has_res)
                ent["traffic_over_algorithmic"] = ent["hbm_bytes_per_launch"] / sum(ent["algorithmic_bytes_per_launch"].values())
            if "SQ_VALU_MFMA_BUSY_CYCLES" in ent and "GRBM_GUI_ACTIVE" in ent:
                mfmas = 3.0 * B * 1024 * N * 9 * C / (16 * 16 * 32)              # v_mfma_f32_16x16x32_f16 instructions
                ent["mfma_instructions"] = mfmas
                ent["clock_GHz_from_GRBM"] = ent["GRBM_GUI_ACTIVE"] / 8 / (ent["avg_duration_us"] * 1e3)
                # SQ_VALU_MFMA_BUSY_CYCLES counts 16 cycles per 16x16x32 MFMA; 1024 SIMDs
                ent["mfma_util"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (ent["GRBM_GUI_ACTIVE"] / 8 * 1024)
            out["shapes"][name] = ent
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
