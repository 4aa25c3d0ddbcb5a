#!/usr/bin/env python3
"""dev: in a rocprofv3 --kernel-trace CSV, which kernels run right before / after the launches whose name contains a
given substring (where do stray small kernels sit in the step)?   python tools/trace_neighbours.py <kernel_trace.csv> <substr>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sub = sys.argv[2]
short = lambda n: n.split("(")[0][-70:]
ctx = collections.Counter()
for i, r in enumerate(rows):
    if sub in r["Kernel_Name"]:
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        ctx[(prev, nxt, r.get("Queue_Id", ""), r.get("Grid_Size", "") + "/" + r.get("Workgroup_Size", ""))] += 1
print(len(rows), "kernels;", sum(ctx.values()), "matching")
for (p, n, q, g), c in ctx.most_common(40):
    print(f"{c:5d}  q{q} grid {g}  after {p}   before {n}")
