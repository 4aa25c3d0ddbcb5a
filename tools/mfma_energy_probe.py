#!/usr/bin/env python3
"""dev (round 6): runs tools/bin/mfma_energy_probe (bare fp16 MFMA loops, operands in registers) and samples the socket power
of the busiest GPU beside it: W, TFLOP/s and pJ per executed FLOP of the matrix cores alone.  python tools/mfma_energy_probe.py"""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        best = 0.0
        for f in files:
            try:
                best = max(best, float(open(f).read()) * 1e-6)
            except Exception:      # noqa: BLE001
                pass
        samples.append((time.time(), best))
        stop.wait(0.02)


th = threading.Thread(target=sampler, daemon=True)
th.start()
p = subprocess.Popen([os.path.join(ROOT, "tools", "bin", "mfma_energy_probe"), "3.0"], stdout=subprocess.PIPE, text=True)
marks = [time.time()]
for line in p.stdout:
    if line.startswith("RESULT"):
        marks.append(time.time())
        kv = dict(x.split("=") for x in line.split()[2:])
        t1 = marks[-1]
        t0 = t1 - float(kv["seconds"])
        w = [v for (t, v) in samples if t0 + 0.5 <= t <= t1 - 0.1]
        watts = sum(w) / max(1, len(w))
        tf = float(kv["tflops"])
        print(f"{line.split()[1]:9s} {kv['waves_per_simd']} wave(s) per SIMD: {tf:7.1f} TFLOP/s executed  {watts:6.0f} W  "
              f"{watts / (tf * 1e12) * 1e12:5.2f} pJ per FLOP  ({tf / 2500:.3f} of 2.5 PF)", flush=True)
p.wait()
stop.set()
