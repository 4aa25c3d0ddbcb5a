#!/usr/bin/env python3
"""dev: the tail of a rocprofv3 --kernel-trace CSV as the short table tools/step_timeline.py reads (name, queue, start, end):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o kt -- python3 bench.py --steps 6 --warmup 3 ...
    python tools/kt_shorten.py gpurun_out/kt/kt_kernel_trace.csv gpurun_out/kt_tail.csv [rows = 6000]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-keep:]
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["name", "queue", "start", "end"])
    for r in rows:
        name = r["Kernel_Name"]
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        w.writerow([name[:80], r.get("Queue_Id", "0"), r["Start_Timestamp"], r["End_Timestamp"]])
print("wrote", len(rows), "rows to", dst)
