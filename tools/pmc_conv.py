#!/usr/bin/env python3
"""Runs the dominant kernel (f16x3 3x3 convolution, forward and input-gradient launches of a train step) in its three
launch shapes at B = 128, a few launches each, for rocprofv3 --pmc passes (one counter group per pass):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o fetch --output-format csv -- python3 tools/pmc_conv.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE ...        (separate passes: TCC has 4 slots, FETCH_SIZE takes 3)
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...

Shapes (grid sizes tell them apart in the trace): 128->128 with residual (grid 512 x 1, 6 launches), 256->128 up-block
conv1 (512 x 1 with C = 256: 5 launches), 128->256 input gradient of the same layer (512 x 2: 4 launches).
`python3 tools/pmc_conv.py --parse <dir>` turns the three counter_collection CSVs found under <dir> into the JSON that
bench.py reads (profiles/r02_pmc_conv3x3_f16x3.json)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # name, C, N, residual, launches (distinct counts identify the shape in the trace)
    ("fwd_128_128_res", 128, 128, True, 6),
    ("fwd_256_128", 256, 128, False, 5),
    ("dgrad_128_256", 128, 256, False, 4),
]
B = 128


def algorithmic_bytes(C, N, res):
    """per launch: x with the vertical halo of the 8-row tiles (10/8), y, planes of x, residual"""
    px = B * 1024
    return {"x_with_halo": px * C * 4 * 10 / 8 * (N // 128), "y": px * N * 4, "planes": px * C * 4,
            "residual": px * N * 4 if res else 0}


def run():
    import torch
    from mulan_amd import ops
    ops.lib.load()
    torch.manual_seed(0)
    for name, C, N, has_res, n in SHAPES:
        x, w = torch.randn(B, 1024, C, device="cuda"), torch.randn(3, 3, C, N, device="cuda") * 0.05
        bias = torch.randn(N, device="cuda") if has_res else None
        cb = torch.randn(B, N, device="cuda") if has_res else None
        res = torch.randn(B, 1024, N, device="cuda") if has_res else None
        xmax, wmax = ops.absmax_rows(x), ops.absmax_rows(w.view(1, -1))
        for _ in range(n):
            y, xs = ops.conv3x3_raw(x, w, bias, cb, res, xmax=xmax, planes=True, wmax=wmax)
        torch.cuda.synchronize()
        print("done", name, float(y[0, 0, 0]))


def parse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if "conv3x3_f16x3_v3_kernel" in r["Kernel_Name"]]
    if not rows:
        raise SystemExit(f"no conv3x3_f16x3_v3_kernel rows under {d}")
    # group dispatches by (C inferred from launch count and grid): key = (Grid_Size, Counter) -> values per dispatch
    by = {}
    for r in rows:
        by.setdefault((int(r["Grid_Size"]), r["Counter_Name"]), []).append(
            (int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {"kernel": "conv3x3_f16x3_v3_kernel", "batch": B,
           "corrections": "FETCH_SIZE x 1024 x 2 (gfx950 counts the 128-B requests of 16-B-per-lane loads as 64 B); "
                          "WRITE_SIZE x 1024 exact for 16-B-per-lane stores; one counter group per rocprofv3 pass",
           "shapes": {}}
    # 128->128 and 256->128 share grid 512 x 256 threads = 131072; split them by launch order inside each pass
    for name, C, N, has_res, n in SHAPES:
        grid = B * 4 * (N // 128) * 256
        ent = {"C": C, "N": N, "residual": has_res, "launches_profiled": n}
        for counter in sorted({k[1] for k in by if k[0] == grid}):
            vals = sorted(by[(grid, counter)])
            if grid == B * 4 * 256:                 # two shapes on this grid: the first 6 dispatches are 128->128
                vals = vals[:6] if C == 128 else vals[6:11]
            if not vals:
                continue
            ent[counter] = sum(v for _, v, _ in vals[1:]) / max(1, len(vals) - 1)         # skip the first (cold) launch
            ent["avg_duration_us"] = sum(t for _, _, t in vals[1:]) / max(1, len(vals) - 1) / 1e3
        if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
            ent["hbm_read_bytes_per_launch"] = ent["FETCH_SIZE"] * 1024 * 2
            ent["hbm_write_bytes_per_launch"] = ent["WRITE_SIZE"] * 1024
            ent["hbm_bytes_per_launch"] = ent["hbm_read_bytes_per_launch"] + ent["hbm_write_bytes_per_launch"]
            ent["algorithmic_bytes_per_launch"] = algorithmic_bytes(C, N, has_res)
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
