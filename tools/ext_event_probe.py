#!/usr/bin/env python3
"""Dev probe (round 4): can work OUTSIDE a replayed HIP graph wait for a point INSIDE it?  The multi-rank train step wants
the gradient all-reduce of bucket k (a collective, outside the graph) to start as soon as the captured backward pass has
produced that bucket, while the rest of the graph still runs.  mulan_event_record_external on the capturing stream plants an event-record NODE in the
graph being captured (hipStreamGetCaptureInfo_v2 + hipGraphAddEventRecordNode; torch.cuda.Event(external=True) raises
"External events are disallowed in rocm" in torch 2.10, and hipEventRecordWithFlags(..., hipEventRecordExternal) returns
hipErrorInvalidValue in the runtime it ships) instead of an internal fork/join edge.  Checked here:
  (1) a side stream that waits for the event after graph.replay() sees the value written BEFORE the node (every replay:
      the value changes from replay to replay, so a stale wait shows), and
  (2) it gets going long before the graph ends (timestamps).
    python tools/ext_event_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    dev = torch.device("cuda")
    a = torch.randn(4096, 4096, device=dev)        # 64 MB: one elementwise pass ~ 35 us
    x = torch.zeros(1 << 20, device=dev)
    y = torch.zeros_like(x)
    main_s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    from mulan_amd import lib as L
    lib = L.load()
    evp = ctypes.c_void_p()
    assert lib.mulan_event_create(ctypes.byref(evp)) == 0
    ev = evp.value

    def chain(n):
        b = a
        for _ in range(n):
            b = b * 1.0001 + 0.5
        return b

    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main_s):
        chain(2)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main_s):
            chain(20)                  # ~ 1 ms in front
            x.add_(1.0)
            rc = lib.mulan_event_record_external(ev, main_s.cuda_stream, None)            # an event-record node
            print("mulan_event_record_external inside the capture ->", rc, flush=True)
            out = chain(600)           # the long tail (~ 20 ms) the side stream should NOT wait for
    torch.cuda.synchronize()
    ok = True
    for it in range(1, 6):
        t0 = torch.cuda.Event(enable_timing=True); t_side = torch.cuda.Event(enable_timing=True)
        t_end = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main_s):
            t0.record()
            g.replay()
            t_end.record()
        assert lib.mulan_stream_wait_event(side.cuda_stream, ev) == 0
        with torch.cuda.stream(side):
            y.copy_(x)
            t_side.record()
        torch.cuda.synchronize()
        val = float(y[0])
        ms_side, ms_end = t0.elapsed_time(t_side), t0.elapsed_time(t_end)
        good = val == float(it) and ms_side < 0.5 * ms_end
        ok &= good
        print(f"replay {it}: side stream saw x = {val:.0f} (want {it}) after {ms_side:.2f} ms; graph ended after {ms_end:.2f} ms"
              f"  {'OK' if good else 'NOT OK'}", flush=True)
    print("external event nodes usable for overlap:", ok)


if __name__ == "__main__":
    main()
