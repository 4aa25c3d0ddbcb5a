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
        self.paused = False         # True while a train step is being captured into a HIP graph: hooks do not launch
        self.capture = None         # while / after a capture: {"order": [bucket, ...], "events": {bucket: handle}}
        self.pending = [0] * len(self.buckets)
        self.ready_order = []       # bucket indices in the order they were launched in the last backward
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
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            from . import ops
            wg = ops.side_stream()          # weight gradients of this bucket may still be in flight on their own stream
            if wg is not None:
                self.side.wait_stream(wg)
            with torch.cuda.stream(self.side):
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
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
    # plants an event-record node behind everything bucket k depends on (mulan_event_record_external: the capturing
    # stream's position and the weight-gradient stream's); after graph.replay() the collective stream waits for node k
    # and all-reduces bucket k under the rest of the graph (the reference gets the same overlap from XLA inside
    # pmap(scan(train_step)), ldm/experiment.py:89-95,341).
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
        self.capture = {"order": [], "events": events, "open": True}

    def _mark(self, bi):
        from . import lib, ops
        L = lib.load()
        ev = self.capture["events"][bi]
        # the collective stream joins the capture here only to carry the node's dependencies: the capturing stream's
        # position and the weight-gradient stream's (the bucket's last weight gradient may still be queued there)
        self.side.wait_stream(torch.cuda.current_stream())
        wg = ops.side_stream()
        if wg is not None:
            with torch.cuda.stream(wg):
                forked = torch.cuda.is_current_stream_capturing()
            if forked:      # (not yet part of this capture: nothing of this step is queued there, and a capturing stream
                self.side.wait_stream(wg)    # must not wait for an event recorded outside its capture)
        lib.check(L.mulan_event_record_external(ev, self.side.cuda_stream), "mulan_event_record_external")
        self.launched[bi] = True
        self.capture["order"].append(bi)

    def end_capture(self):
        """call inside the capture, after the backward pass: the collective stream rejoins the capturing stream"""
        if self.capture is not None and self.capture["open"]:
            self.capture["open"] = False
            if self.capture["order"]:
                torch.cuda.current_stream().wait_stream(self.side)

    def allreduce_captured(self):
        """after graph.replay(): every bucket that was marked in the capture is all-reduced behind its event node (i.e.
        while the rest of the graph still runs), whatever was not marked behind the whole graph; the current stream then
        waits for all of them"""
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
            lib.check(L.mulan_stream_wait_event(self.side.cuda_stream, self.capture["events"][bi]), "mulan_stream_wait_event")
            self.launched[bi] = True
            self.ready_order.append(bi)
            with torch.cuda.stream(self.side):
                self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self.finish()

    def allreduce_now(self):
        """All buckets at once, after a graph-replayed backward (no hooks ran): same result as prepare() ... finish(),
        without the overlap."""
        if not self.enabled:
            return
        self.prepare()
        self.finish()

    def finish(self):
        """Launch whatever did not fire (unused leaves), then make the compute stream wait for all buckets."""
        if not self.enabled:
            return
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
