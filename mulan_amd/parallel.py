"""Data-parallel plumbing: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

Replaces jax.pmap + lax.pmean of the reference (ldm/experiment.py:89-95,341,347,365): parameters are
replicated, the batch is sharded by rank, and the only data-path collective is the all-reduce of the
flat gradient buffer.  TrainState lays the buffer out in the order gradients become ready during backward
(train_state.grad_ready_rank), so cutting it into contiguous buckets by offset gives buckets that complete one
after the other; each bucket is all-reduced on a side HIP stream as soon as its last gradient has landed, so the
exchange overlaps the rest of the backward pass (`ready_order` records the order the buckets actually fired in).  The 1/world_size factor is folded into the
optimizer kernel (grad_scale).  Works unchanged with backend "gloo" on CPU tensors (tests).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialises torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "MULAN_FORCE_DEVICE" in os.environ:      # test hook: several ranks share one GPU (needs MULAN_DIST_BACKEND=gloo)
        local = int(os.environ["MULAN_FORCE_DEVICE"])
    backend = backend or os.environ.get("MULAN_DIST_BACKEND")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


# the hand-off between the replayed graph and the collectives: signal words set by kernel nodes (default) or, with
# MULAN_OVERLAP_SIGNAL=0, the event-record nodes alone (which fire late in a multi-branch graph on this runtime)
USE_SIGNALS = os.environ.get("MULAN_OVERLAP_SIGNAL", "1") == "1"


class GradReducer:
    """Bucketed, backward-overlapped sum all-reduce of a flat gradient buffer.

    `leaves` is a list of (tensor, offset, numel) views of `flat_grad` (the tensors whose .grad are
    those views).  Buckets are contiguous [lo, hi) ranges of the flat buffer; a bucket is launched
    when all of its leaves have reported ready through their post-accumulate hook.
    """

    def __init__(self, flat_grad, leaves, bucket_bytes=None, process_group=None):
        if bucket_bytes is None:             # 64 MB: a few large all-reduces per step (xGMI ring: per-link bound)
            bucket_bytes = int(os.environ.get("MULAN_BUCKET_MB", "64")) << 20
        self.flat = flat_grad
        self.pg = process_group
        self.world = world_size()
        self.enabled = self.world > 1
        self.cuda = flat_grad.is_cuda
        self.side = torch.cuda.Stream() if (self.cuda and self.enabled) else None
        # RCCL: a collective issued as a *synchronous* op is launched on the stream that is current (torch >= 2.8; before
        # that on the process group's own stream, which the current one is then made to wait for) -- i.e. on `side`, the
        # stream calibrate_stream() chose, instead of a pool stream whose hardware queue nobody measured.  Neither form
        # blocks the host.  Other backends (gloo in the tests) keep the asynchronous op + wait()
        self.sync_ops = bool(self.side is not None and dist.get_backend(process_group) == "nccl")
        order = sorted(leaves, key=lambda l: l[1])
        cap = max(1, bucket_bytes // 4)
        self.buckets = []          # [lo, hi, n_leaves]
        self.leaf_bucket = {}
        lo = hi = None
        cnt = 0
        for t, off, n in order:
            if lo is None:
                lo, hi, cnt = off, off + n, 0
            if off + n - lo > cap and cnt > 0:
                self.buckets.append([lo, hi, cnt])
                lo, hi, cnt = off, off + n, 0
            hi = max(hi, off + n)
            cnt += 1
            self.leaf_bucket[id(t)] = len(self.buckets)
        if lo is not None:
            self.buckets.append([lo, hi, cnt])
        # pad bucket ranges so they tile the whole flat buffer (alignment padding between leaves)
        for i, b in enumerate(self.buckets):
            b[1] = self.buckets[i + 1][0] if i + 1 < len(self.buckets) else flat_grad.numel()
        if self.buckets:
            self.buckets[0][0] = 0
        # measurement only (bench.py `allreduce_exposed_ms`): the step with its collectives left out -- every bucket is
        # accounted for as usual, nothing is sent.  The replicas drift apart from the first such step on.
        self.skip_collectives = False
        self.paused = False         # True while a train step is being captured into a HIP graph: hooks do not launch
        self.tick, self.tick_dev = 0, None   # replay counter: the value the graph's signal kernels store (before_replay)
        self.capture = None         # while / after a capture: {"order": [bucket, ...], "events": {bucket: handle}}
        self.pending = [0] * len(self.buckets)
        self.ready_order = []       # bucket indices in the order they were launched in the last backward
        self.eager_order = []       # ... in the order an eager (hook-driven) step launched them: allreduce_now follows it
        self.works = []
        self.launched = [False] * len(self.buckets)
        if self.enabled:
            for t, off, n in order:
                t.register_post_accumulate_grad_hook(self._hook)

    def prepare(self):
        """Call before each backward."""
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.ready_order = []
        self.works = []

    def _launch(self, bi):
        lo, hi, _ = self.buckets[bi]
        view = self.flat[lo:hi]
        self.launched[bi] = True
        self.ready_order.append(bi)
        if self.flat.is_cuda:
            from . import ops
            ops.flush_slab_reductions()      # weight gradients of this bucket whose slabs are written but not summed yet
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            from . import ops
            wg = ops.side_stream()          # weight gradients of this bucket may still be in flight on their own stream
            if wg is not None:
                self.side.wait_stream(wg)
            with torch.cuda.stream(self.side):
                self._all_reduce(view)
        elif not self.skip_collectives:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _all_reduce(self, view):
        """sum over the ranks, on the current stream's timeline (see sync_ops)"""
        if self.skip_collectives:
            return
        if self.sync_ops:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _hook(self, t):
        gv = getattr(t, "_gview", None)     # gradient that did not land in the flat buffer yet (no sink for this op)
        if gv is not None and t.grad is not None and t.grad.data_ptr() != gv.data_ptr():
            gv.copy_(t.grad)
            t.grad = gv       # so TrainState.collect_grads() does not overwrite the reduced bucket later
        if self.paused:
            if self.capture is not None and self.capture["open"]:
                bi = self.leaf_bucket[id(t)]
                self.capture["hooks"] += 1
                self.pending[bi] -= 1
                if self.pending[bi] == 0 and not self.launched[bi]:
                    self._mark(bi)
            return
        bi = self.leaf_bucket[id(t)]
        self.pending[bi] -= 1
        if self.pending[bi] == 0 and not self.launched[bi]:
            self._launch(bi)

    # ---- a replayed (HIP-graph) backward pass with the all-reduce still overlapped ------------------------------------
    # The collectives stay outside the graph (torch.distributed), but each bucket's all-reduce may start as soon as the
    # REPLAYED backward pass has produced the bucket: while the step is captured, the hook that would launch bucket k
    # plants a one-thread kernel node behind the bucket's last weight gradient that stores the replay's tick into the
    # bucket's word of signal memory (mulan_signal_set); after graph.replay() the collective stream waits for that word
    # (hipStreamWaitValue32) and all-reduces bucket k under the rest of the graph (the reference gets the same overlap
    # from XLA inside pmap(scan(train_step)), ldm/experiment.py:89-95,341).  An event-record node is planted as well
    # (mulan_event_record_external): the fallback where the device cannot wait on values -- on this runtime such nodes
    # all fire together near the end of a multi-branch graph (profiles/DESIGN_r04.md 1, tools/overlap_timing_probe.py).
    def begin_capture(self):
        """call before capturing a train step (with paused = True)"""
        if not (self.enabled and self.side is not None):
            return
        import ctypes
        from . import lib
        L = lib.load()
        self.prepare()
        events = self.capture["events"] if self.capture else {}
        for bi in range(len(self.buckets)):          # (created here: no handle is made while the stream captures)
            if bi not in events:
                h = ctypes.c_void_p()
                lib.check(L.mulan_event_create(ctypes.byref(h)), "mulan_event_create")
                events[bi] = h.value
        signals = self.capture["signals"] if self.capture else {}
        if USE_SIGNALS and signals is not None:
            for bi in range(len(self.buckets)):
                if bi not in signals:
                    h = ctypes.c_void_p()
                    rc = L.mulan_signal_create(ctypes.byref(h))
                    if rc != 0:                      # (no stream wait-value on this device: the event nodes remain)
                        signals = None
                        break
                    signals[bi] = h.value
        if not USE_SIGNALS:
            signals = None
        if self.tick_dev is None:
            self.tick_dev = torch.zeros(1, dtype=torch.int32, device=self.flat.device)
        self.capture = {"order": [], "events": events, "signals": signals, "open": True, "hooks": 0, "hooks_at_mark": []}

    def _mark(self, bi):
        from . import lib, ops
        L = lib.load()
        ops.flush_slab_reductions()          # (as in _launch: the bucket must be complete where its signal is planted)
        ev = self.capture["events"][bi]
        # the collective stream joins the capture here only to carry the node's dependencies: the capturing stream's
        # position and the weight-gradient stream's (the bucket's last weight gradient may still be queued there)
        self.side.wait_stream(torch.cuda.current_stream())
        wg = ops.side_stream()
        chain = None
        if wg is not None:
            with torch.cuda.stream(wg):
                forked = torch.cuda.is_current_stream_capturing()
            if forked:      # (not yet part of this capture: nothing of this step is queued there, and a capturing stream
                self.side.wait_stream(wg)    # must not wait for an event recorded outside its capture)
                chain = wg.cuda_stream
        # the node becomes a link of the weight-gradient branch (the branch that reaches this point last: the main chain
        # runs ahead of it), so it fires at its place in the backward pass; as a leaf HIP's executor ran it when the main
        # branch had drained (all buckets but the last released together 4.7 ms before the end of the graph)
        # (MULAN_EVENT_NODE_CHAIN=0: dev A/B of the EVENT node's placement only -- as a leaf instead of a link of the
        # weight-gradient branch.  It must not move the signal kernel below: that one always goes behind the bucket's last
        # weight gradient, or the all-reduce would read gradients the weight-gradient stream has not written yet.)
        ev_chain = chain if os.environ.get("MULAN_EVENT_NODE_CHAIN", "1") == "1" else None
        lib.check(L.mulan_event_record_external(ev, self.side.cuda_stream, ev_chain), "mulan_event_record_external")
        sig = self.capture["signals"]
        if sig is not None:
            # the hand-off that works (see allreduce_captured): a one-thread kernel node that stores this replay's tick
            # into the bucket's signal word, in the weight-gradient branch behind the bucket's last weight gradient (that
            # branch runs behind the main chain, whose position it joins here like every weight-gradient launch does) --
            # or in the main chain while no weight gradient has been launched yet
            if chain is not None:
                wg.wait_stream(torch.cuda.current_stream())
                lib.check(L.mulan_signal_set(sig[bi], self.tick_dev.data_ptr(), wg.cuda_stream), "mulan_signal_set")
            else:
                lib.check(L.mulan_signal_set(sig[bi], self.tick_dev.data_ptr(), torch.cuda.current_stream().cuda_stream),
                          "mulan_signal_set")
        self.launched[bi] = True
        self.capture["order"].append(bi)
        self.capture["hooks_at_mark"].append(self.capture["hooks"])

    def end_capture(self):
        """call inside the capture, after the backward pass: the collective stream rejoins the capturing stream"""
        if self.capture is not None and self.capture["open"]:
            self.capture["open"] = False
            if self.capture["order"]:
                torch.cuda.current_stream().wait_stream(self.side)
                from . import ops
                wg = ops.side_stream()               # the signal kernels sit behind the last weight gradient there
                if wg is not None and self.capture.get("signals") is not None:
                    with torch.cuda.stream(wg):
                        forked = torch.cuda.is_current_stream_capturing()
                    if forked:
                        torch.cuda.current_stream().wait_stream(wg)

    def before_replay(self):
        """call right before graph.replay(): the tick the replayed signal kernels will store (stream-ordered parameter)"""
        if self.capture is not None and self.capture.get("signals") is not None and self.tick_dev is not None:
            self.tick += 1
            self.tick_dev.fill_(self.tick)

    def _wait_bucket(self, stream, bi):
        """`stream` waits until bucket bi of the replay in flight is complete"""
        from . import lib
        L = lib.load()
        sig = self.capture.get("signals")
        if sig is not None:
            lib.check(L.mulan_stream_wait_signal(stream.cuda_stream, sig[bi], self.tick), "mulan_stream_wait_signal")
        else:
            lib.check(L.mulan_stream_wait_event(stream.cuda_stream, self.capture["events"][bi]), "mulan_stream_wait_event")

    def calibrate_stream(self, replay, tries=5):
        """Pick a collective stream on which the overlap really happens.  HIP multiplexes the streams of a process (the
        graph's internal branch streams included) onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default; raising it
        slows the two-stream train step by a third, measured), and work on one queue runs in FIFO order whatever its
        dependencies say: on an unlucky queue the collective that waits for a point INSIDE the graph sits behind the
        rest of the graph (tools/overlap_timing_probe.py: the first of three otherwise identical runs lost the overlap,
        85.0 against 80.0 ms per step).  The mapping cannot be queried, so it is measured: one trial replay per
        candidate stream; the candidate waits for the first marked bucket and stamps a timing event, and the candidate
        released earliest before the graph's end is kept.  The trial
        replays run the captured forward / backward kernels only (the optimizer is outside a multi-rank graph): the
        gradient buffer they fill is rewritten by the next step."""
        if not (self.enabled and self.side is not None and self.capture and self.capture["order"]):
            return None
        if os.environ.get("MULAN_COMM_STREAM_PROBE", "1") != "1":
            return None
        from . import lib
        L = lib.load()
        first = self.capture["order"][0]
        main = torch.cuda.current_stream()
        best, best_lead, leads = None, -1.0, []
        cand = self.side
        for _ in range(tries):
            t_c, t_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.before_replay()
            replay()
            t_end.record(main)
            self._wait_bucket(cand, first)
            t_c.record(cand)
            main.synchronize()
            cand.synchronize()
            lead = t_c.elapsed_time(t_end)           # ms between the candidate's release and the end of the graph
            leads.append(round(lead, 2))
            if lead > best_lead:
                best, best_lead = cand, lead
            cand = torch.cuda.Stream()               # (every candidate is tried: a stream can be partly blocked too)
        self.side = best
        self.calibration = leads
        # for the record: when each marked bucket's node fires, in ms before the end of the graph (one more trial replay)
        stamps = []
        t_end = torch.cuda.Event(enable_timing=True)
        self.before_replay()
        replay()
        t_end.record(main)
        for bi in self.capture["order"]:
            self._wait_bucket(best, bi)
            t = torch.cuda.Event(enable_timing=True)
            t.record(best)
            stamps.append(t)
        main.synchronize()
        best.synchronize()
        self.bucket_leads = [round(t.elapsed_time(t_end), 2) for t in stamps]
        sig = self.capture.get("signals")
        if sig is not None:      # when the signal kernels themselves ran (their own clock stamps): ms before the last one
            import ctypes
            ticks = []
            for bi in self.capture["order"]:
                buf = (ctypes.c_uint32 * 2)()
                lib.check(L.mulan_signal_read(sig[bi], buf), "mulan_signal_read")
                ticks.append(int(buf[1]))
            last = ticks[-1]
            self.signal_leads = [round(((last - t) & 0xffffffff) * 1e-5, 2) for t in ticks]      # 10 ns ticks -> ms
        return leads

    def allreduce_captured(self):
        """after graph.replay(): every bucket that was marked in the capture is all-reduced behind its signal (i.e. while
        the rest of the graph still runs), whatever was not marked behind the whole graph; the current stream then waits
        for all of them"""
        if not self.enabled:
            return
        marked = self.capture["order"] if (self.capture is not None and self.side is not None) else []
        if not marked:
            return self.allreduce_now()
        from . import lib
        L = lib.load()
        self.prepare()
        for bi in marked:
            lo, hi, _ = self.buckets[bi]
            self._wait_bucket(self.side, bi)
            self.launched[bi] = True
            self.ready_order.append(bi)
            with torch.cuda.stream(self.side):
                self._all_reduce(self.flat[lo:hi])
        self.finish()

    def allreduce_now(self):
        """All buckets at once, after a graph-replayed backward (no hooks ran): same result as prepare() ... finish(),
        without the overlap.  The buckets go out in the order an eager step launches them (`eager_order`, recorded by the
        eager step every run starts with): a collective sequence must be the same on every rank, also when one rank runs
        a step eagerly while the others replay theirs (a rank whose capture failed, bench.py's timed kernel step on rank 0
        before round 5) -- found by the 8-rank shared-GPU test: gloo aborted on mismatched sizes, RCCL would hang."""
        if not self.enabled:
            return
        self.prepare()
        for bi in self.eager_order:
            if not self.launched[bi]:
                self._launch(bi)
        self.finish()

    def finish(self):
        """Launch whatever did not fire (unused leaves), then make the compute stream wait for all buckets."""
        if not self.enabled:
            return
        if not self.paused and len(self.ready_order) > len(self.eager_order):
            self.eager_order = list(self.ready_order)      # (complete only once every hooked bucket has fired)
        for bi in range(len(self.buckets)):
            if not self.launched[bi]:
                self._launch(bi)
        for w in self.works:
            w.wait()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        self.works = []


def allreduce_mean_scalars(values, device):
    """lax.pmean over a dict of python/0-d scalars (ldm/experiment.py:347-348,365-366)."""
    if world_size() == 1:
        return values
    keys = sorted(values)
    t = torch.stack([torch.as_tensor(values[k], dtype=torch.float32, device=device).reshape(()) for k in keys])
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t / world_size()
    return {k: t[i] for i, k in enumerate(keys)}


def all_gather_tensor(t):
    """jax.lax.all_gather over the data-parallel axis (the sample grid at eval checkpoints, ldm/experiment.py:287):
    [world * B, ...] on every rank"""
    if world_size() == 1:
        return t
    out = [torch.empty_like(t) for _ in range(world_size())]
    dist.all_gather(out, t.contiguous())
    return torch.cat(out, dim=0)
