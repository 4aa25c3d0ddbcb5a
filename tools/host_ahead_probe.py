#!/usr/bin/env python3
"""dev (round 4): is the host ahead of the GPU in the replayed train step?  Host time spent in GraphedStep._fill and in
graph.replay() per step, against the step time: if one of them takes ~ a whole step, the host waits for the previous
replay there and the GPU idles for the launch latency at every step boundary.   python tools/host_ahead_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from mulan_amd import experiment as E
    from mulan_amd.config import load_config_file
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.data.dataset = "synthetic"
    B = 128
    config.training.batch_size_train = B
    config.training.batch_size_eval = B
    config.training.substeps = 1
    exp = E.Experiment_VDM(config)
    g = torch.Generator().manual_seed(0)
    batch = {"images": torch.randint(0, 256, (B, 32, 32, 3), generator=g, dtype=torch.uint8).cuda(),
             "labels": torch.zeros(B, dtype=torch.int32).cuda(), "conditioning": torch.zeros(B, dtype=torch.uint8).cuda()}
    state = exp.state
    for _ in range(6):
        state, _m = exp.train_step(exp._train_rng, state, batch)
    torch.cuda.synchronize()
    gs = exp._graphed
    assert gs is not None
    t = {"fill": [], "replay": [], "sync": []}
    fill0, replay0, sync0 = gs._fill, gs.graph.replay, gs.copied.synchronize

    def timed(name, fn):
        def f(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            t[name].append((time.perf_counter() - t0) * 1e3)
            return r
        return f
    ev = []      # (before the replay, after it): GPU time between the end of one replay and the start of the next = the
                 # staging copies and noise kernels GraphedStep._fill queues between them

    def replay_with_events():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        replay0()
        b.record()
        ev.append((a, b))
    gs._fill = timed("fill", fill0)
    gs.graph.replay = timed("replay", replay_with_events)
    gs.copied.synchronize = timed("sync", sync0)
    n = 12
    t0 = time.perf_counter()
    marks = []
    for _ in range(n):
        state, _m = exp.train_step(exp._train_rng, state, batch)
        marks.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    print(f"{n} steps in {total:.1f} ms = {total / n:.2f} ms per step; host returned from step i at (ms): "
          + " ".join(f"{m:.1f}" for m in marks))
    print("GPU ms inside the replayed graph : " + " ".join(f"{a.elapsed_time(b):6.2f}" for a, b in ev))
    print("GPU ms between two replays       : " + " ".join(f"{ev[i][1].elapsed_time(ev[i + 1][0]):6.2f}" for i in range(len(ev) - 1)))
    for k, v in t.items():
        print(f"host ms in {k:7s}: " + " ".join(f"{x:6.2f}" for x in v))


if __name__ == "__main__":
    main()
