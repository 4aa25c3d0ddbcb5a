#!/usr/bin/env python3
"""dev: one replayed train step out of a shortened kernel trace (name, queue, start, end in ns; written by the recipe in
profiles/README.md): time per queue, time with >= 1 / >= 2 kernels running, gaps, and the head / tail of the two
backward streams.   python tools/step_timeline.py gpurun_out/kt_tail.csv [--back N] [--list ms0 ms1]
(bench.py ends with two eager steps of conv_roofline -- as run with event timers, then serial: --back 3 is a replayed step)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["start"], r["end"] = int(r["start"]), int(r["end"])
ad = [i for i, r in enumerate(rows) if "adamw" in r["name"]]
back = int(sys.argv[sys.argv.index("--back") + 1]) if "--back" in sys.argv else 0   # 0: the last step of the trace
a, b = ad[-2 - back], ad[-1 - back]
step = rows[a + 1:b + 1]
T0, T1 = step[0]["start"], step[-1]["end"]
print(f"step span {(T1 - T0) / 1e6:.2f} ms, {len(step)} kernels")
byq = collections.defaultdict(list)
for r in step:
    byq[r["queue"]].append(r)
for q, rs in byq.items():
    busy = sum(r["end"] - r["start"] for r in rs)
    gaps = [rs[i + 1]["start"] - rs[i]["end"] for i in range(len(rs) - 1)]
    pos = sorted(g for g in gaps if g > 0)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms, from {(rs[0]['start'] - T0) / 1e6:.2f} to {(rs[-1]['end'] - T0) / 1e6:.2f} ms; "
          f"gaps: sum {sum(pos) / 1e6:.2f} ms, median {pos[len(pos) // 2] / 1e3 if pos else 0:.1f} us, > 5 us: {sum(g > 5000 for g in pos)}, > 20 us: {sum(g > 20000 for g in pos)}")
ev = []
for r in step:
    ev += [(r["start"], 1), (r["end"], -1)]
ev.sort()
cur, last, t_any, t_two = 0, None, 0, 0
for t, d in ev:
    if last is not None:
        t_any += (t - last) * (cur >= 1)
        t_two += (t - last) * (cur >= 2)
    cur += d
    last = t
print(f">= 1 kernel running {t_any / 1e6:.2f} ms, >= 2 running {t_two / 1e6:.2f} ms, nothing running {(T1 - T0 - t_any) / 1e6:.2f} ms")
if "--list" in sys.argv:
    lo, hi = float(sys.argv[sys.argv.index("--list") + 1]), float(sys.argv[sys.argv.index("--list") + 2])
    for r in step:
        if lo <= (r["start"] - T0) / 1e6 <= hi:
            print(f"  q{r['queue']} {(r['start'] - T0) / 1e6:8.3f} +{(r['end'] - r['start']) / 1e3:7.1f} us  {r['name']}")
