#!/usr/bin/env python3
"""Dev probe (round 4): what does the overlapped all-reduce of the REPLAYED train step buy, on the one GPU these boxes have?
The full bench workload (MuLAN-eps, CIFAR config, 128 images) is run as if it were one of N ranks, with a STAND-IN for the
collective: a kernel that occupies the collective stream for the time a ring all-reduce of that bucket would take over xGMI
(bytes * 2 (N - 1) / N / link rate; one block that sleeps: it takes no bandwidth and one wave of one CU, like RCCL's
kernels take little).  Everything else is the product path: GradReducer's buckets, hooks, event-record nodes and streams,
GraphedStep, the optimizer launch behind the collectives.  Three variants, alternating:
   replay + signal hand-off (shipped)  MULAN_GRAPH_OVERLAP=1: bucket k's collective waits for the word a kernel node of
                                      the graph sets behind bucket k (MULAN_OVERLAP_SIGNAL=0: for its event-record node)
   replay, collectives behind it    MULAN_GRAPH_OVERLAP=0: round 3's replayed form
   eager step                       MULAN_HIP_GRAPH=0: hooks launch the collectives from the host during backward
    python tools/overlap_timing_probe.py [--ranks 8] [--gbps 90] [--steps 20]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--gbps", type=float, default=90.0, help="ring rate per direction and link the stand-in assumes (GB/s)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rccl", action="store_true",
                    help="no stand-in: the process group is RCCL (backend nccl) with this one rank, so the capture, the "
                         "signal waits and the optimizer run beside ProcessGroupNCCL's own threads and streams (a "
                         "one-rank all-reduce moves nothing: a smoke run of the plumbing, not a timing)")
    ap.add_argument("--footprint", type=int, default=0,
                    help="K > 0: the stand-in is a kernel with RCCL's footprint -- K workgroups of 512 threads that stream 2 x the "
                         "bucket through HBM, paced over the ring's duration (tools/footprint_kernel.hip) -- instead of one "
                         "sleeping wave")
    ap.add_argument("--footprint-lds", type=int, default=0, help="bytes of LDS each footprint workgroup holds (e.g. 98304: "
                                                                 "no convolution block can share its CU)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl" if a.rccl else "gloo", rank=0, world_size=1)
    from mulan_amd import experiment as E, parallel
    from mulan_amd.config import load_config_file

    # cycles of torch.cuda._sleep per microsecond on this box
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)

    class Done:
        def wait(self):
            return True

    calls = {"n": 0, "us": 0.0}

    fp_lib, fp_dst = None, {}
    if a.footprint > 0:
        import ctypes
        import subprocess
        so = "/tmp/libfootprint.so"
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-w", "-shared", "-fPIC", "-o", so,
                        os.path.join(ROOT, "tools", "footprint_kernel.hip")], check=True)
        fp_lib = ctypes.CDLL(so)
        fp_lib.footprint_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_ulonglong,
                                          ctypes.c_int, ctypes.c_void_p]

    def standin_all_reduce(t, op=None, group=None, async_op=False):
        nbytes = t.numel() * t.element_size()
        if nbytes >= (1 << 20):                     # a gradient bucket (the scalar metrics cost nothing)
            us = nbytes * 2.0 * (a.ranks - 1) / a.ranks / (a.gbps * 1e3)
            if fp_lib is not None:                  # K workgroups stream the bucket twice over the ring's duration
                dst = fp_dst.get(nbytes)
                if dst is None:
                    dst = fp_dst[nbytes] = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
                rc = fp_lib.footprint_copy(t.data_ptr(), dst.data_ptr(), nbytes, a.footprint, int(us * 1e3), a.footprint_lds,
                                           torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            else:
                torch.cuda._sleep(int(us * cyc_per_us))  # on the current (= collective) stream
            calls["n"] += 1
            calls["us"] += us
        return Done() if async_op else None

    real_world = parallel.world_size
    parallel.world_size = lambda: a.ranks             # GradReducer / the scalar mean believe in N ranks
    real_all_reduce = dist.all_reduce

    def counted_all_reduce(t, *args, **kw):
        if t.numel() * t.element_size() >= (1 << 20):
            calls["n"] += 1
        return real_all_reduce(t, *args, **kw)

    dist.all_reduce = counted_all_reduce if a.rccl else standin_all_reduce

    def run(hip_graph, overlap, collectives=False):
        config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
        config.data.dataset = "synthetic"
        config.training.batch_size_train = a.batch
        config.training.batch_size_eval = a.batch
        config.training.substeps = 1
        config.training.hip_graph = hip_graph
        exp = E.Experiment_VDM(config)
        exp.world = a.ranks
        exp.graph_overlap = bool(overlap)             # (opt-in since round 5)
        exp.graph_collectives = bool(collectives)     # (--rccl only: the all-reduces captured into the graph)                           # the step applies 1 / N and keeps the optimizer outside the graph
        g = torch.Generator().manual_seed(0)
        batch = {"images": torch.randint(0, 256, (a.batch, 32, 32, 3), generator=g, dtype=torch.uint8).cuda(),
                 "labels": torch.zeros(a.batch, dtype=torch.int32).cuda(),
                 "conditioning": torch.zeros(a.batch, dtype=torch.uint8).cuda()}
        state = exp.state
        for _ in range(5):
            state, _m = exp.train_step(exp._train_rng, state, batch)
        torch.cuda.synchronize()
        calls["n"], calls["us"] = 0, 0.0
        t0 = time.perf_counter()
        for _ in range(a.steps):
            state, _m = exp.train_step(exp._train_rng, state, batch)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        cap = exp.reducer.capture or {}
        info = (exp._graphed is not None, len(exp.reducer.buckets),
                dict(marked=list(cap.get("order", [])), trial_leads_ms=getattr(exp.reducer, "calibration", None),
                     bucket_leads_ms=getattr(exp.reducer, "bucket_leads", None),
                     signal_kernel_leads_ms=getattr(exp.reducer, "signal_leads", None), hooks_at_mark=cap.get("hooks_at_mark"),
                     leaves_per_bucket=[b[2] for b in exp.reducer.buckets]),
                calls["n"] / a.steps, calls["us"] / a.steps / 1e3)
        exp._graphed = None
        del exp, state
        torch.cuda.empty_cache()
        return ms, info

    print(f"stand-in collective: ring all-reduce over {a.ranks} ranks at {a.gbps:.0f} GB/s per link; "
          f"{cyc_per_us:.0f} sleep cycles per us", flush=True)
    for rep in range(1):
        variants = [("replay + signal hand-off (opt-in)", True, True, False), ("replay, collectives behind the graph", True, False, False),
                    ("eager step, hooks launch the collectives", False, True, False)]
        if a.rccl:
            variants.append(("replay, collectives captured into the graph", True, False, True))
        for name, hg, ov, cc in variants:
            ms, (graphed, nb, marked, ncalls, coll_ms) = run(hg, ov, cc)
            print(f"{name:42s}: {ms:7.2f} ms per step   (graph {graphed}, {nb} buckets, {marked}, "
                  f"{ncalls:.0f} collectives = {coll_ms:.2f} ms of stand-in per step)", flush=True)
    parallel.world_size = real_world


if __name__ == "__main__":
    main()
