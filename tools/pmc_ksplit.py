#!/usr/bin/env python3
"""MFMA utilisation of the GroupNorm-fed convolution of a forward-only chain at a small batch, as 4-wave blocks (dev switch
tune[27] = 1) and as k-split blocks of eight waves (the launcher's choice for <= 256 short-tile blocks), for one
rocprofv3 counter pass:

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d out -o pmc --output-format csv -- python3 tools/pmc_ksplit.py
  python3 tools/pmc_ksplit.py --parse out        (-> JSON: per kernel symbol duration, clock, MFMA busy share)

The two forms are different kernel symbols (template argument KS), so the trace tells them apart."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B, E, REPS = 16, 128, 8


def run():
    import torch
    from mulan_amd import ops
    from mulan_amd.lib import call
    ops.lib.load()
    torch.manual_seed(0)
    mk = lambda *s, sc=1.0: torch.randn(*s, device="cuda") * sc
    x = mk(B, 1024, E, sc=2.0)
    g0, b0, w0, c0 = mk(E), mk(E, sc=0.3), mk(3, 3, E, E, sc=0.03), mk(E)
    cb = mk(B, E)
    with torch.no_grad():
        h = ops.gn_conv3x3(x, None, g0, b0, w0, c0, res=x)          # leaves the statistics of h on it
        for ks_off in (1, 0):
            call("mulan_set_tuning", 27, ks_off)
            for _ in range(REPS):
                ops.gn_conv3x3(h, None, g0, b0, w0, c0, cbias=cb)
            torch.cuda.synchronize()
    call("mulan_set_tuning", 27, 0)


def parse(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    by = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "conv3x3_f16x3_v3_kernel<0, false, 1, false" not in n:
                continue
            targs = re.search(r"conv3x3_f16x3_v3_kernel<([^>]*)>", n).group(1)
            key = "k-split (KS = 2)" if targs.replace(" ", "").endswith(",2") else "4-wave blocks (KS = 1)"
            e = by.setdefault(key, {}).setdefault(int(r["Dispatch_Id"]), {})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
            e["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e["symbol"] = "conv3x3_f16x3_v3_kernel<" + targs + ">"
    out = {"workload": f"GroupNorm-fed convolution 128 -> 128 + FiLM bias, statistics handed over, B = {B} (2-row tiles, 256 blocks)",
           "forms": {}}
    mfmas = 3.0 * B * 1024 * E * 9 * E / (16 * 16 * 32)
    for key, disp in by.items():
        v = [e for _, e in sorted(disp.items())][2:]                 # (first launches: cold caches, clock ramp)
        n = len(v)
        busy = sum(e["SQ_VALU_MFMA_BUSY_CYCLES"] for e in v) / n
        gui = sum(e["GRBM_GUI_ACTIVE"] for e in v) / n
        us = sum(e["ns"] for e in v) / n / 1e3
        out["forms"][key] = {"symbol": v[0]["symbol"], "launches": n, "avg_duration_us": round(us, 2),
                             "clock_GHz_from_GRBM": round(gui / 8 / (us * 1e3), 3),
                             "mfma_util": round(busy / (gui / 8 * 1024), 4),
                             "mfma_instructions": mfmas}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
