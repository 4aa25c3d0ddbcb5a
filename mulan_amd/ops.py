"""torch.autograd wiring over the C ABI (include/mulan_hip.h).

Each Function's forward and backward are launches of hand-written HIP kernels through
`mulan_amd.lib`; torch supplies device buffers, the current stream and the autograd tape only.
Shapes: images are [B, 1024, C] (NHWC with H = W = 32 flattened), matrices row-major, all fp32.
"""
import ctypes
import math
import weakref

import torch
from torch.autograd.function import once_differentiable

from . import lib
from .lib import call, ptr, stream

H = W = 32
HW = H * W
D = HW * 3
KERNEL_TIMER = None   # bench.py sets this to a list to time the dominant kernel with HIP events
# "f32": exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).  "f16x3": 3-pass fp16 split (two scaled fp16 pieces per fp32 operand)
# with fp32-equivalent products.  "bf16x6": 6-pass bf16 split with fp32-equivalent products
# (XLA's float32/HIGHEST matmul precision) for the eligible convolutions.  MULAN_CONV_MODE overrides.
import os as _os
CONV_MODE = _os.environ.get("MULAN_CONV_MODE", "f16x3")


def _c(t):
    return t if (t is None or t.is_contiguous()) else t.contiguous()


def _chk(t, name="tensor"):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise lib.MulanHipError(f"{name}: expected a float32 CUDA(HIP) tensor, got {t.dtype} on {t.device}")


# ----------------------------------------------------------------------------- raw launches
def _timed(name, flops, launch):
    """runs launch(); under bench.py's KERNEL_TIMER brackets it with HIP events on the launch stream"""
    if KERNEL_TIMER is None:
        launch()
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    launch()
    e.record()
    KERNEL_TIMER.append((name, s, e, flops))


MAX_PARTS = 16

# ----------------------------------------------------------------------------- weight-gradient side stream
# Nothing in the backward pass waits for a weight gradient (only the all-reduce of its bucket and the optimizer do), while
# the input-gradient chain is strictly serial.  The weight-gradient kernels therefore go to a second HIP stream and
# run beside the GroupNorm backward / input-gradient convolution of the layers in front (see SIDE_WGRAD_SHARE for the
# split of the chip between the two streams), instead of taking their own slot in one queue.  Only gradients with
# a sink in the flat gradient buffer take this path (nothing on the main stream reads them before side_join()).
# Scope: `with weight_gradient_stream(): loss.backward()` (the train step does this); outside such a block everything
# stays on the current stream.
SIDE_STREAM = _os.environ.get("MULAN_SIDE_STREAM", "1") == "1"
# side launches whose operands are kept alive before the main stream waits for the oldest (2: +1.0 ms, 16: +0.2 ms per step)
SIDE_DEPTH = int(_os.environ.get("MULAN_SIDE_DEPTH", "6"))
# While the weight-gradient launches share the chip with the input-gradient chain they aim for 120 blocks instead of
# 240 (the `share_chip` argument of the plane-fed weight-gradient entry points, see wgrad_splits_p; no library-global
# state is involved): a weight-gradient block owns its CU, so 240 of them leave 16 CUs to the
# main stream; with 120 the launch takes about as long as the main stream's kernels of the same layer (GroupNorm
# backward + input-gradient convolution) and both streams keep running side by side: -2.9 % per step (scan 96 ... 240,
# profiles/DESIGN_r04.md 3.2).  MULAN_SIDE_WGRAD_SHARE=0 keeps 240.
SIDE_WGRAD_SHARE = _os.environ.get("MULAN_SIDE_WGRAD_SHARE", "1") == "1"
_SIDE = {"stream": None, "pending": None, "active": False, "scope": False}


class weight_gradient_stream:
    """context manager around a backward pass: weight gradients with a sink run on the side stream; on exit the current
    stream waits for all of them"""

    def __enter__(self):
        _SIDE["active"] = SIDE_STREAM
        _SIDE["scope"] = True
        _SLAB_PENDING.clear()            # (records of a backward pass that an exception cut short must not reach this one)
        _SLAB_KEEP.clear()
        return self

    def __exit__(self, *exc):
        flush_slab_reductions()          # (while the side stream is still the place where weight gradients run)
        _SIDE["active"] = False
        _SIDE["scope"] = False
        _SLAB_KEEP.clear()
        side_join()
        return False


def _alone():
    """the `alone` argument of the convolution entry points: 1 unless the launch is part of the backward pass of a train
    step (a weight_gradient_stream() scope, with or without MULAN_SIDE_STREAM, so that both modes sum in the same order):
    forward passes, evaluators and the ODE likelihood's vector-Jacobian products have the chip to themselves, and small
    launches may then run as k-split blocks (conv3x3_f16x3_v3.hip)"""
    return 0 if _SIDE["scope"] else 1


def _share_chip():
    """the share_chip argument of the plane-fed weight-gradient launches: 1 inside a weight_gradient_stream() scope"""
    return int(bool(_SIDE["active"] and SIDE_WGRAD_SHARE))


def side_stream():
    """the weight-gradient stream (None until the first launch went there)"""
    return _SIDE["stream"]


def _on_side(launch, keep):
    """launch() on the side stream, ordered after everything issued so far on the current stream.  `keep`: the tensors
    the launch reads -- referenced until the current stream has waited for the launch, so that the caching allocator
    (stream-ordered on the current stream) cannot hand their memory out underneath it."""
    import collections
    main = torch.cuda.current_stream()
    if _SIDE["stream"] is None:
        _SIDE["stream"], _SIDE["pending"] = torch.cuda.Stream(), collections.deque()
    side, pending = _SIDE["stream"], _SIDE["pending"]
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        launch()
        done = torch.cuda.Event()
        done.record(side)
    if _SLAB_KEEP:                   # (slab sets of earlier launches that this one summed: alive until it has run)
        keep = (keep, tuple(_SLAB_KEEP))
        _SLAB_KEEP.clear()
    pending.append((done, keep))
    while len(pending) > SIDE_DEPTH:
        main.wait_event(pending.popleft()[0])


def side_join():
    """the current stream waits for every weight gradient issued so far (before the optimizer / a gradient read)"""
    pending = _SIDE["pending"]
    if pending:
        main = torch.cuda.current_stream()
        while pending:
            main.wait_event(pending.popleft()[0])


def _side_ok(sink):
    return _SIDE["active"] and sink is not None and sink.is_cuda


def absmax_rows(x):
    """[B,16] int32: fp32 bit patterns of 16 partial maxima of |x[b]| (the per-image maximum is the max of a row);
    they feed the per-image operand scales of the f16x3 kernels"""
    B = x.shape[0]
    out = torch.empty((B, MAX_PARTS), device=x.device, dtype=torch.int32)
    call("mulan_absmax_rows", ptr(x), ptr(out), B, x.numel() // B, stream())
    return out


def cached_absmax(x):
    """maxima a producer kernel (GroupNormFn) or an earlier consumer left on the tensor, if still valid; else a pass
    over x, remembered on the tensor for its other consumers (a gradient that feeds a convolution and a 1x1 layer)"""
    c = getattr(x, "_absmax", None)
    if c is not None and c[1] == x._version and c[0].shape[0] == x.shape[0]:
        return c[0]
    m = absmax_rows(x)
    try:
        x._absmax = (m, x._version)
    except (AttributeError, RuntimeError):
        pass
    return m


def view_keep_absmax(x, *shape):
    """x.view(shape) that keeps the maxima found on x (a view is a new Python object and would lose them; the
    per-image maxima do not depend on how the trailing dimensions are folded as long as dim 0 stays the batch)"""
    v = x.view(*shape)
    c = getattr(x, "_absmax", None)
    if c is not None and c[1] == x._version and v.shape[0] == c[0].shape[0]:
        v._absmax = (c[0], v._version)
    return v


TEE_COLSUM = _os.environ.get("MULAN_TEE_COLSUM", "1") == "1"     # A/B switch: 0 = the bias gradient takes its own pass


class TeeFn(torch.autograd.Function):
    """(x, x) for a tensor with two consumers (a U-Net skip connection): the backward pass forms the sum of the two
    gradients itself, in one kernel that also leaves the maxima of the sum for the convolution behind it (instead of
    autograd's elementwise add followed by a separate maxima pass)"""

    @staticmethod
    def forward(ctx, x):
        a, b = x.view_as(x), x.view_as(x)
        for v in (a, b):
            for tag in ("_absmax", "_gnstats"):     # what the producer left on x stays valid for the aliases
                c = getattr(x, tag, None)
                if c is not None and c[1] == x._version:
                    setattr(v, tag, (c[0], v._version))
        return a, b

    @staticmethod
    @once_differentiable
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return gb if ga is None else ga
        ga, gb = _c(ga), _c(gb)
        if CONV_MODE != "f16x3" or ga.dim() < 2 or (ga.numel() // ga.shape[0]) % 4 != 0:
            return ga + gb
        out = torch.empty_like(ga)
        m = torch.empty((ga.shape[0], MAX_PARTS), device=ga.device, dtype=torch.int32)
        N = ga.shape[-1]
        if TEE_COLSUM and ga.dim() == 3 and N % 4 == 0 and 256 % (N // 4) == 0:
            # the sum is the output gradient of the convolution that produced x: leave its column sums (16 partial vectors
            # per image) for that convolution's bias gradient, which then needs no pass of its own over the tensor
            parts = torch.empty((ga.shape[0] * MAX_PARTS, N), device=ga.device, dtype=torch.float32)
            call("mulan_add_absmax_rows_colsum", ptr(ga), ptr(gb), ptr(out), ptr(m), ptr(parts), ga.shape[0],
                 ga.numel() // ga.shape[0], N, stream())
            out._colsum_parts = (parts, out._version)
        else:
            call("mulan_add_absmax_rows", ptr(ga), ptr(gb), ptr(out), ptr(m), ga.shape[0], ga.numel() // ga.shape[0], stream())
        out._absmax = (m, out._version)
        return out


# The gradient sum of a block output with two consumers inside the GroupNorm backward kernel of the first consumer
# (mulan_groupnorm_bwd_fused, add1b) instead of in a kernel of its own (TeeFn): the skip-connection gradient, which the
# backward pass produces first, waits in a box until that kernel runs.  A/B switch: 0 = TeeFn.
TEE_MAILBOX = _os.environ.get("MULAN_TEE_MAILBOX", "1") == "1"


class _GradBox:
    """where the gradient that reaches a tensor through its second consumer waits for the first consumer's backward.
    The hand-over is only right if, in every backward pass, the deposit (MailFn.backward) comes before the claiming
    consumer's backward takes the box's content: `deposits` / `takes` count both, and a deposit that finds more takes
    than deposits has missed its consumer -- an error, not a silently dropped gradient."""
    __slots__ = ("grad", "claimed", "deposits", "takes", "__weakref__")

    def __init__(self):
        self.grad = None
        self.claimed = False
        self.deposits = 0
        self.takes = 0

    def take(self):
        """the claiming consumer's backward: the waiting gradient (or None), the box empty again"""
        g, self.grad = self.grad, None
        self.takes += 1
        _PENDING_BOXES.discard(self)
        return g


_PENDING_BOXES = weakref.WeakSet()      # boxes that hold a gradient nobody has taken yet


def _check_no_gradient_left_waiting():
    """at the next forward pass: a box still holding a gradient means the last backward pass ended without the claiming
    consumer's backward (it was not on the differentiated path): that gradient never reached the tensor"""
    left = [b for b in _PENDING_BOXES if b.grad is not None]
    if left:
        _PENDING_BOXES.clear()
        for b in left:
            b.grad = None
        raise RuntimeError(
            f"{len(left)} skip-connection gradient(s) were left waiting in their tee() mailbox when the backward pass ended: "
            "the consumer that claimed the box (GnConv3x3Fn with skip=True) was not part of that backward pass, so the "
            "gradient of the tee()'d tensor is incomplete.  Differentiate through both consumers, or set MULAN_TEE_MAILBOX=0.")


class MailFn(torch.autograd.Function):
    """alias of x for its second consumer (tee_take): the gradient goes into the box instead of to x"""

    @staticmethod
    def forward(ctx, x, box):
        ctx.box = box
        y = x.view_as(x)
        for tag in ("_absmax", "_gnstats"):
            c = getattr(x, tag, None)
            if c is not None and c[1] == x._version:
                setattr(y, tag, (c[0], y._version))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if g is not None:
            box = ctx.box
            if box.takes > box.deposits:
                raise RuntimeError(
                    "tee() mailbox: the skip-connection gradient arrived AFTER the backward of the consumer that adds it "
                    "in-kernel (GnConv3x3Fn with skip=True) had run, so it would be dropped.  The mailbox needs the second "
                    "consumer's gradient first (true for the shipped U-Nets: the up path is differentiated before the "
                    "down path); set MULAN_TEE_MAILBOX=0 for graphs with another order.")
            box.deposits += 1
            box.grad = _c(g) if box.grad is None else box.grad + g       # (a second arrival: not in the U-Nets)
            _PENDING_BOXES.add(box)
        return None, None


def tee(x):
    """x for its two consumers (a block output that also feeds a U-Net skip connection): (a, b).  With TEE_MAILBOX both
    are x itself, carrying a box: if the first consumer claims it (GnConv3x3Fn with skip=True: the next ResnetBlock),
    tee_take(b) -- called where the second alias is consumed -- diverts b's gradient into the box and the first
    consumer's GroupNorm backward adds it in-kernel.  Unclaimed (another kind of first consumer): autograd sums the two
    gradients itself.  Without TEE_MAILBOX: two aliases whose gradients TeeFn.backward adds (mulan_add_absmax_rows)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    if TEE_MAILBOX:
        _check_no_gradient_left_waiting()
        x._grad_box = _GradBox()
        return x, x
    return TeeFn.apply(x)


def tee_take(b):
    """the second alias of tee(x), at the place where it is consumed (after the first consumer's forward)"""
    box = getattr(b, "_grad_box", None)
    if box is None or not box.claimed or not (torch.is_grad_enabled() and b.requires_grad):
        return b
    return MailFn.apply(b, box)


class ParamPacker:
    """Once-per-step weight preparation (f16x3 mode): maxima and both packed operands of every eligible parameter
    leaf of a TrainState in two launches; the leaves carry views of the result (`_prepacked`), valid until the
    optimizer touches the parameters again (`invalidate`)."""

    def __init__(self, flat, leaves_with_offsets):
        """leaves_with_offsets: [(leaf tensor (view of flat), element offset)]"""
        recs, self.items, off = [], [], 0
        for leaf, eoff in leaves_with_offsets:
            if leaf.dim() == 4 and tuple(leaf.shape[:2]) == (3, 3):
                kind, C, N = 0, leaf.shape[2], leaf.shape[3]
                fwd_ok, bwd_ok = C % 16 == 0 and N % 128 == 0, N % 16 == 0 and C % 128 == 0
            elif leaf.dim() == 2 and leaf.shape[0] % 128 == 0 and leaf.shape[1] % 128 == 0 and leaf.numel() <= 512 * 512:
                kind, C, N = 1, leaf.shape[0], leaf.shape[1]
                fwd_ok = bwd_ok = True
            else:
                continue
            if not (fwd_ok or bwd_ok):
                continue
            nb = leaf.numel() * 4
            d0 = off if fwd_ok else -1
            off += nb if fwd_ok else 0
            d1 = off if bwd_ok else -1
            off += nb if bwd_ok else 0
            recs.append([eoff, kind, C, N, d0, d1, leaf.numel(), 0])
            self.items.append((leaf, d0, d1, nb))
        self.n = len(recs)
        self.flat = flat
        self.valid = False
        if self.n:
            self.table = torch.tensor(recs, dtype=torch.int64, device=flat.device)
            self.maxima = torch.empty((self.n, MAX_PARTS), dtype=torch.int32, device=flat.device)
            self.packed = torch.empty(off, dtype=torch.uint8, device=flat.device)
            for i, (leaf, d0, d1, nb) in enumerate(self.items):
                leaf._prepacked = (self, self.packed[d0:d0 + nb] if d0 >= 0 else None,
                                   self.packed[d1:d1 + nb] if d1 >= 0 else None, self.maxima[i:i + 1])

    def refresh(self):
        if self.n:
            call("mulan_param_maxima", ptr(self.flat), ptr(self.table), self.n, ptr(self.maxima), stream())
            call("mulan_param_pack_f16x3", ptr(self.flat), ptr(self.table), self.n, ptr(self.maxima), ptr(self.packed),
                 stream())
            self.valid = True

    def invalidate(self):
        self.valid = False


def _prepacked(w, direction):
    """(wp, wmax) prepared by a ParamPacker for this leaf, if still valid"""
    pre = getattr(w, "_prepacked", None)
    if pre is not None and pre[0].valid and CONV_MODE == "f16x3" and pre[1 + direction] is not None:
        return pre[1 + direction], pre[3]
    return None


def _pack_weights(w, C, N, flip, wmax=None):
    """weights pre-split into the LDS tile layout of the fast convolution kernels; returns (wp, wmax or None)"""
    L = lib.load()
    pre = _prepacked(w, int(flip))
    if pre is not None:
        return pre
    if CONV_MODE == "f16x3":
        wp = torch.empty(L.mulan_conv3x3_pack_f16x3_bytes(C, N), device=w.device, dtype=torch.uint8)
        if wmax is None:
            wmax = absmax_rows(w.view(1, -1))
        call("mulan_conv3x3_pack_f16x3", ptr(w), ptr(wp), ptr(wmax), C, N, flip, stream())
        return wp, wmax
    wp = torch.empty(L.mulan_conv3x3_pack_bf16x6_bytes(C, N), device=w.device, dtype=torch.uint8)
    call("mulan_conv3x3_pack_bf16x6", ptr(w), ptr(wp), C, N, flip, stream())
    return wp, None


def planes_eligible(C, N):
    """the plane-fed weight-gradient kernel (f16x3 mode) covers 128-multiples of channels"""
    return CONV_MODE == "f16x3" and C % 128 == 0 and N % 128 == 0


def conv3x3_raw(x, w, bias=None, cbias=None, res=None, xmax=None, planes=False, wmax=None):
    """x [B,1024,C], w [3,3,C,N] -> [B,1024,N]   (xmax: absmax_rows(x) if the caller already has it, f16x3 mode).
    planes=True (f16x3 mode): returns (y, xs) with xs the split fp16 planes of x for conv3x3_wgrad_planes_raw."""
    _chk(x, "conv input")
    B, C, N = x.shape[0], x.shape[-1], w.shape[-1]
    assert w.shape[:3] == (3, 3, C), (w.shape, C)
    y = torch.empty((B, HW, N), device=x.device, dtype=torch.float32)
    mode = 0
    if cbias is not None:
        mode = 1 if cbias.dim() == 2 else 2
    fast = CONV_MODE in ("bf16x6", "f16x3") and C % 16 == 0 and N % 128 == 0
    flops = 2.0 * B * HW * 9 * C * N
    if not fast:
        variant = "<128,2,2>" if N > 64 else ("<64,2,2>" if N > 32 else "<32,4,1>")
        _timed("conv3x3_fwd_kernel" + variant, flops,
               lambda: call("mulan_conv3x3_fwd", ptr(x), ptr(w), ptr(bias), ptr(cbias), mode, ptr(res), ptr(y), B, H, W,
                            C, N, stream()))
        return y
    assert fast or not planes
    wp, wmax = _pack_weights(w, C, N, 0, wmax)
    if CONV_MODE == "f16x3":
        if xmax is None:
            xmax = absmax_rows(x)
        xs = torch.empty(B * HW * C * 4, device=x.device, dtype=torch.uint8) if planes else None
        # by-product: the maxima of y, left on the tensor for whichever f16x3 kernel reads it next
        ymax = torch.empty((B, MAX_PARTS), device=x.device, dtype=torch.int32) if (H // 8) * (N // 128) <= MAX_PARTS else None
        _timed("conv3x3_f16x3_kernel", flops,
               lambda: call("mulan_conv3x3_fwd_f16x3_alone", ptr(x), ptr(xmax), ptr(wp), ptr(wmax), ptr(bias), ptr(cbias), mode,
                            ptr(res), ptr(y), ptr(xs), ptr(ymax), _alone(), B, H, W, C, N, stream()))
        if ymax is not None:
            y._absmax = (ymax, y._version)
        if planes:
            return y, xs
    else:
        _timed("conv3x3_bf16x6_kernel", flops,
               lambda: call("mulan_conv3x3_fwd_bf16x6", ptr(x), ptr(wp), ptr(bias), ptr(cbias), mode, ptr(res), ptr(y),
                            B, H, W, C, N, stream()))
    return y


def conv3x3_dgrad_raw(dy, w, dymax=None, planes=False, wmax=None, want_max=False):
    """dx = conv3x3(dy, flipped w).  planes=True (f16x3 mode): returns (dx, dys), dys = the split planes of dy.
    want_max (f16x3 mode): the kernel leaves the maxima of dx on the tensor (a GroupNorm backward that hands its own
    result on as planes needs them for its bound)."""
    C, N = w.shape[2], w.shape[3]
    if CONV_MODE in ("bf16x6", "f16x3") and N % 16 == 0 and C % 128 == 0:
        B = dy.shape[0]
        dx = torch.empty((B, HW, C), device=dy.device, dtype=torch.float32)
        wp, wmax = _pack_weights(w, C, N, 1, wmax)
        flops = 2.0 * B * HW * 9 * C * N
        if CONV_MODE == "f16x3":
            if dymax is None:
                dymax = absmax_rows(dy)
            dys = torch.empty(B * HW * N * 4, device=dy.device, dtype=torch.uint8) if planes else None
            dxmax = (torch.empty((B, MAX_PARTS), device=dy.device, dtype=torch.int32)
                     if want_max and (H // 8) * (C // 128) <= MAX_PARTS else None)
            _timed("conv3x3_f16x3_kernel", flops,
                   lambda: call("mulan_conv3x3_fwd_f16x3_alone", ptr(dy), ptr(dymax), ptr(wp), ptr(wmax), None, None, 0, None,
                                ptr(dx), ptr(dys), ptr(dxmax), _alone(), B, H, W, N, C, stream()))
            if dxmax is not None:
                dx._absmax = (dxmax, dx._version)
            if planes:
                return dx, dys
        else:
            _timed("conv3x3_bf16x6_kernel", flops,
                   lambda: call("mulan_conv3x3_fwd_bf16x6", ptr(dy), ptr(wp), None, None, 0, None, ptr(dx), B, H, W, N, C,
                                stream()))
        return dx
    wT = torch.empty((3, 3, N, C), device=w.device, dtype=torch.float32)
    call("mulan_conv3x3_wflip", ptr(w), ptr(wT), C, N, stream())
    return conv3x3_raw(dy, wT)


def conv3x3_dgrad_planes_raw(dys, dymax, w, wmax=None, want_max=False):
    """dx = conv3x3(dy, flipped w) with dy given as split planes (scaled with dymax: what a GroupNorm backward hands on,
    mulan_groupnorm_bwd_fused_planes): the plane-fed instantiation of the convolution kernel -- no split, no plane stores"""
    C, N = w.shape[2], w.shape[3]
    B = dymax.shape[0]
    dx = torch.empty((B, HW, C), device=dys.device, dtype=torch.float32)
    wp, wmax = _pack_weights(w, C, N, 1, wmax)
    dxmax = (torch.empty((B, MAX_PARTS), device=dys.device, dtype=torch.int32)
             if want_max and (H // 8) * (C // 128) <= MAX_PARTS else None)
    # alone: no weight-gradient stream beside this launch (the ODE evaluator's vector-Jacobian product, not the backward
    # pass of a train step -- with or without MULAN_SIDE_STREAM, so that both modes sum in the same order): small launches
    # may then run as k-split blocks (conv3x3_f16x3_v3.hip)
    alone = _alone()
    _timed("conv3x3_f16x3_kernel<planes_in,dgrad>", 2.0 * B * HW * 9 * C * N,
           lambda: call("mulan_conv3x3_fwd_f16x3_planes_in_stats", ptr(dys), ptr(dymax), ptr(wp), ptr(wmax), None, None, 0,
                        None, ptr(dx), ptr(dxmax), None, alone, B, H, W, N, C, stream()))
    if dxmax is not None:
        dx._absmax = (dxmax, dx._version)
    return dx


def grad_planes_eligible(C, N):
    """a convolution C -> N whose output gradient may arrive as split planes only: plane-fed weight gradient (128-grids)
    and the two-blocks-per-CU kernel for the input gradient (N % 32 == 0, C % 128 == 0)"""
    return GRAD_PLANES and planes_eligible(C, N) and N % 32 == 0 and C % 128 == 0 and N // 32 <= MAX_PARTS


_NAN = {}


def _planes_only_grad(shape, device, planes, bound):
    """The stand-in autograd carries for a gradient that exists only as split planes: a NaN scalar expanded to the
    gradient's shape (4 bytes; anything that consumed it as numbers would turn NaN at once -- loud, not silent) with the
    real content attached as `_grad_planes = (planes, bound maxima, version)`"""
    n = _NAN.get(device)
    if n is None:
        n = _NAN[device] = torch.full((1,), float("nan"), device=device, dtype=torch.float32)
    g = n.expand(*shape)
    g._grad_planes = (planes, bound, g._version)
    return g


def _grad_planes_of(dy):
    gp = getattr(dy, "_grad_planes", None)
    return gp if (gp is not None and gp[2] == dy._version) else None


def _gv(t):
    """flat-gradient-buffer view registered for a parameter leaf by TrainState (None for ordinary tensors)"""
    return getattr(t, "_gview", None) if t is not None else None


def _fresh(view):
    """a new tensor object over the same storage: lets autograd's AccumulateGrad adopt it without a copy"""
    return view.view(view.shape)


def conv3x3_wgrad_raw(x, dy, out=None, xmax=None, dymax=None):
    B, C, N = x.shape[0], x.shape[-1], dy.shape[-1]
    fast = CONV_MODE in ("bf16x6", "f16x3") and C % 4 == 0 and N % 4 == 0
    kind = CONV_MODE if fast else ""
    fn = "mulan_conv3x3_wgrad" + ("_" + kind if fast else "")
    nbytes = getattr(lib.load(), fn + "_workspace")(B, H, W, C, N)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    dw = out if out is not None else torch.empty((3, 3, C, N), device=x.device, dtype=torch.float32)
    name = "conv3x3_wgrad" + ("_" + kind if fast else "") + "_kernel+slab_reduce"
    flops = 2.0 * B * HW * 9 * C * N
    if kind == "f16x3":
        xmax = absmax_rows(x) if xmax is None else xmax
        dymax = absmax_rows(dy) if dymax is None else dymax
        _timed(name, flops, lambda: call(fn, ptr(x), ptr(xmax), ptr(dy), ptr(dymax), ptr(dw), ptr(ws), B, H, W, C, N, 0,
                                         stream()))
    else:
        _timed(name, flops, lambda: call(fn, ptr(x), ptr(dy), ptr(dw), ptr(ws), B, H, W, C, N, 0, stream()))
    return dw


# ---- slab reductions folded into the next weight-gradient launch (round 6; mulan_conv3x3_wgrad_f16x3_planes_fold)
# Inside the backward pass of a train step (a weight_gradient_stream() scope) a 3x3 weight gradient writes its slabs and
# leaves the record (workspace, dw, S, E) here; the next one sums them in the prologue of its blocks, and whatever is
# still pending when a gradient bucket is complete (parallel.GradReducer) or the pass ends is summed by
# mulan_slab_reduce.  Same summation order, same bits as the reduction launch behind every weight gradient that this
# replaces (146 launches of a train step).  Built for VERDICT r05 item 8 (<= 1000 launches per step) and MEASURED: the
# replayed step at B = 128 takes 75.15 / 75.17 ms with the fold against 74.87 / 75.07 ms without (alternating runs, one
# box, profiles/r06_fold_slab_reduce_ab.log) -- the 8.8 us reduction launches were never the cost, their bytes are, and
# the prologue of a 120-block launch sums them no faster than 576 small blocks do.  OPT-IN (MULAN_FOLD_SLAB_REDUCE=1):
# the default keeps the reduction launch behind every weight gradient; bit-identical either way
# (tests/test_gpu_f16x3.py::test_f16x3_slab_reductions_folded_into_the_next_weight_gradient, tools/fold_check.py).
FOLD_SLAB_REDUCE = _os.environ.get("MULAN_FOLD_SLAB_REDUCE", "0") == "1"
_SLAB_PENDING = []        # [(workspace, dw, S, E)]
_SLAB_KEEP = []           # tensors the launch just issued reads beyond its own arguments (_on_side keeps them alive)


def _pending_array(recs):
    arr = (lib.SlabReduction * max(1, len(recs)))()
    for i, (ws, dw, S, E) in enumerate(recs):
        arr[i].slab, arr[i].out, arr[i].S, arr[i].E, arr[i].accumulate = ws.data_ptr(), dw.data_ptr(), S, E, 0
    return arr


def flush_slab_reductions():
    """sum every pending slab set now (before anything reads those weight gradients: the all-reduce of their bucket, the
    optimizer).  Issued where the weight gradients run: on the side stream while it is active."""
    if not _SLAB_PENDING:
        return
    recs = list(_SLAB_PENDING)
    _SLAB_PENDING.clear()

    def launch():
        for ws, dw, S, E in recs:
            call("mulan_slab_reduce", ptr(ws), ptr(dw), S, E, 0, stream())
    if _SIDE["active"] and recs[0][0].is_cuda:
        _on_side(launch, tuple(t for r in recs for t in r[:2]))
    else:
        launch()


def conv3x3_wgrad_planes_raw(xs, xmax, dys, dymax, B, C, N, out=None):
    """dw from the split planes of x (written by the forward conv) and of dy (written by the input-gradient conv)"""
    share = _share_chip()
    if FOLD_SLAB_REDUCE and _SIDE["scope"] and out is not None:
        L = lib.load()
        S = L.mulan_conv3x3_wgrad_f16x3_planes_splits(B, H, W, C, N, share)
        E = 9 * C * N
        ws = torch.empty(S * E, device=xs.device, dtype=torch.float32)
        take = _SLAB_PENDING[:2]
        del _SLAB_PENDING[:2]
        arr = _pending_array(take)
        _timed("conv3x3_wgrad_f16x3_planes_kernel+slab_reduce", 2.0 * B * HW * 9 * C * N,
               lambda: call("mulan_conv3x3_wgrad_f16x3_planes_fold", ptr(xs), ptr(xmax), ptr(dys), ptr(dymax), ptr(ws), B, H,
                            W, C, N, share, ctypes.addressof(arr) if take else None, len(take), stream()))
        _SLAB_KEEP.extend(t for r in take for t in r[:2])
        # (the record holds ANOTHER tensor object over dw's storage: a second reference to `out` itself would keep
        # autograd's AccumulateGrad from adopting it -- it would clone the not yet written gradient instead, and
        # TrainState.collect_grads would copy that clone over the real one)
        _SLAB_PENDING.append((ws, _fresh(out), S, E))
        return out
    nbytes = lib.load().mulan_conv3x3_wgrad_f16x3_planes_workspace(B, H, W, C, N, share)
    ws = torch.empty(nbytes // 4, device=xs.device, dtype=torch.float32)
    dw = out if out is not None else torch.empty((3, 3, C, N), device=xs.device, dtype=torch.float32)
    _timed("conv3x3_wgrad_f16x3_planes_kernel+slab_reduce", 2.0 * B * HW * 9 * C * N,
           lambda: call("mulan_conv3x3_wgrad_f16x3_planes", ptr(xs), ptr(xmax), ptr(dys), ptr(dymax), ptr(dw), ptr(ws), B,
                        H, W, C, N, 0, share, stream()))
    return dw


def gemm_raw(A, Bm, M, N, K, *, bias=None, R=None, transA=False, transB=False, alpha=1.0, beta=1.0,
             lda=None, ldb=None, batch=1, sA=0, sB=0, out=None):
    """C = alpha * op(A) op(B) + bias + beta * R.  Returns [batch, M, N] (or [M, N] when batch == 1)."""
    lda = lda if lda is not None else (M if transA else K)
    ldb = ldb if ldb is not None else (K if transB else N)
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), device=A.device, dtype=torch.float32)
    ws = None
    if batch == 1 and K >= 256:
        nbytes = lib.load().mulan_gemm_workspace(M, N, K, batch)
        if nbytes:
            ws = torch.empty(nbytes // 4, device=A.device, dtype=torch.float32)
    call("mulan_gemm", ptr(A), ptr(Bm), ptr(out), ptr(bias), ptr(R), M, N, K, lda, ldb, N, N, int(transA),
         int(transB), batch, sA, sB, M * N, M * N, float(alpha), float(beta), ptr(ws), stream())
    return out


def colsum_raw(x2d, nseg, seg, C, out=None, ld=None):
    """out[s][c] = sum of `seg` consecutive rows (row stride ld, default C); long segments are reduced in two stages so
    the grid always has enough blocks to stream from HBM (fixed order => deterministic)."""
    chunk = 512
    ld = C if ld is None else ld
    src = ptr(x2d) if ld == C else x2d.data_ptr()       # rows of a wider matrix: the stride goes to the kernel
    if seg >= 4 * chunk and seg % chunk == 0:
        part = torch.empty((nseg * (seg // chunk), C), device=x2d.device, dtype=torch.float32)
        call("mulan_colsum", src, ptr(part), nseg * (seg // chunk), chunk, C, ld, 0, stream())
        src, seg, ld = ptr(part), seg // chunk, C
    if out is None:
        out = torch.empty((nseg, C), device=x2d.device, dtype=torch.float32)
    call("mulan_colsum", src, ptr(out), nseg, seg, C, ld, 0, stream())
    return out


def randn(shape, seed, offset, device, out=None):
    if out is None:
        out = torch.empty(shape, device=device, dtype=torch.float32)
    call("mulan_randn", ptr(out), out.numel(), int(seed) & (2**64 - 1), int(offset), stream())
    return out


# ----------------------------------------------------------------------------- conv
class Conv3x3Fn(torch.autograd.Function):
    """y = conv3x3(x, w) + bias + cbias + res   (ldm/model_vdm.py:633-656)"""

    @staticmethod
    def forward(ctx, x, w, bias, cbias, res):
        x, w = _c(x), _c(w)
        f16 = CONV_MODE == "f16x3" and x.shape[-1] % 4 == 0        # per-image maxima: shared by fwd and wgrad
        xmax = cached_absmax(x) if f16 else None
        pre = _prepacked(w, 0)
        wmax = (pre[1] if pre is not None else absmax_rows(w.view(1, -1))) if f16 else None   # shared by both packs
        ctx.wmax = wmax
        # the forward kernel hands its split input planes to the weight-gradient kernel: saved instead of x (same bytes)
        ctx.planes = planes_eligible(x.shape[-1], w.shape[-1]) and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
        if ctx.planes:
            y, xs = conv3x3_raw(x, w, _c(bias), _c(cbias), _c(res), xmax=xmax, planes=True, wmax=wmax)
            ctx.save_for_backward(xs, w)
            ctx.xshape = x.shape
        else:
            y = conv3x3_raw(x, w, _c(bias), _c(cbias), _c(res), xmax=xmax, wmax=wmax)
            ctx.save_for_backward(x, w)
        ctx.xmax = xmax
        ctx.has = (bias is not None, None if cbias is None else cbias.dim(), res is not None)
        ctx.gv = (_gv(w), _gv(bias))
        if ctx.gv[1] is not None and ctx.needs_input_grad[2]:
            # A GroupNorm that consumes y writes this bias' gradient (the channel sums of the dy it produces) from inside
            # its backward kernel; `twin`: the bias of the shortcut layer whose output is `res` sees the same gradient.
            twin = getattr(res, "_bias_twin", None) if res is not None else None
            y._bias_sink = (ctx.gv[1], twin[0] if (twin is not None and twin[1] == res._version) else None, y._version)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return _conv3x3_backward(ctx, dy)


def _conv3x3_backward(ctx, dy):
    """backward of y = conv3x3(x, w) + bias + cbias + res -> (dx, dw, dbias, dcbias, dres); ctx: the Conv3x3Fn context
    or the stand-in GnConv3x3Fn builds (saved_tensors = (x or its planes, w), planes, xmax, wmax, has, gv,
    needs_input_grad[0..4])"""
    if True:
        x, w = ctx.saved_tensors
        has_bias, cb_dim, has_res = ctx.has
        gvw, gvb = ctx.gv
        B, N = dy.shape[0], dy.shape[-1]
        want_max = bool(getattr(ctx, "want_dx_max", False))

        def wgrad_from_planes(dys, dymax):
            if _side_ok(gvw):
                dw_, xmax = _fresh(gvw), ctx.xmax
                _on_side(lambda: conv3x3_wgrad_planes_raw(x, xmax, dys, dymax, B, w.shape[2], N, out=dw_),
                         (x, xmax, dys, dymax))
                return dw_
            return conv3x3_wgrad_planes_raw(x, ctx.xmax, dys, dymax, B, w.shape[2], N,
                                            out=_fresh(gvw) if gvw is not None else None)

        gp = _grad_planes_of(dy)
        if gp is not None:
            # dy exists only as split planes (written by the GroupNorm backward behind this convolution,
            # mulan_groupnorm_bwd_fused_planes): the plane-fed convolution kernel forms dx, the weight-gradient kernel
            # reads the same planes, bias / FiLM gradients come from the channel sums that kernel left.  Nothing here
            # may read dy's numbers (the tensor is a NaN stand-in).
            assert not has_res and cb_dim != 3 and getattr(dy, "_colsum", None) is not None, "planes-only gradient misrouted"
            dys, dymax = gp[0], gp[1]
            dx = conv3x3_dgrad_planes_raw(dys, dymax, w, wmax=ctx.wmax, want_max=want_max)
            dw = wgrad_from_planes(dys, dymax) if ctx.needs_input_grad[1] else None
        else:
            nan = _NAN.get(dy.device)
            if nan is not None and dy.numel() > 1 and dy.data_ptr() == nan.data_ptr() and all(s == 0 for s in dy.stride()):
                # the NaN stand-in of _planes_only_grad without (valid) planes: autograd handed on another tensor object than
                # the GroupNorm backward returned (a tensor hook, retain_grad or an accumulation on conv1's output)
                raise RuntimeError("planes-only gradient arrived without its planes: hooks / retain_grad / a second consumer "
                                   "on the output of a ResnetBlock's conv1 are not supported with MULAN_GRAD_PLANES=1")
            dy = _c(dy)
            dymax = cached_absmax(dy) if (CONV_MODE == "f16x3" and N % 4 == 0) else None   # shared by dgrad and wgrad
        if gp is not None:
            pass
        elif ctx.planes:                # x is the plane tensor here
            dx, dys = conv3x3_dgrad_raw(dy, w, dymax=dymax, planes=True, wmax=ctx.wmax, want_max=want_max)
            try:       # the shortcut layer that shares this dy takes its weight gradient from the same planes
                dy._planes = (dys, dymax, dy._version)
            except (AttributeError, RuntimeError):
                pass
            dw = wgrad_from_planes(dys, dymax)
        else:
            dx = (conv3x3_dgrad_raw(dy, w, dymax=dymax, wmax=ctx.wmax, want_max=want_max)
                  if ctx.needs_input_grad[0] else None)
            dw = None
            if ctx.needs_input_grad[1]:   # written straight into the flat gradient buffer when the weight is a leaf
                if _side_ok(gvw) and (dymax is None or ctx.xmax is not None):
                    dw, xmax = _fresh(gvw), ctx.xmax
                    _on_side(lambda: conv3x3_wgrad_raw(x, dy, out=dw, xmax=xmax, dymax=dymax), (x, dy, xmax, dymax))
                else:
                    dw = conv3x3_wgrad_raw(x, dy, out=_fresh(gvw) if gvw is not None else None, xmax=ctx.xmax,
                                           dymax=dymax)
        dbias = dcb = None
        per_sample = None
        parts = getattr(dy, "_colsum_parts", None)         # left by TeeFn.backward's add: 16 partial column sums per image
        if parts is not None and not (parts[1] == dy._version and parts[0].shape[1] == N):
            parts = None
        if parts is not None and has_bias and ctx.needs_input_grad[2] and not (cb_dim == 2 and ctx.needs_input_grad[3]):
            dbias = colsum_raw(parts[0], 1, parts[0].shape[0], N, out=_fresh(gvb).view(1, N) if gvb is not None else None).view(N)
            try:
                dy._biasgrad = (dbias.view(N), dy._version, None)
            except (AttributeError, RuntimeError):
                pass
        elif (has_bias and ctx.needs_input_grad[2]) or (cb_dim == 2 and ctx.needs_input_grad[3]):
            cs = getattr(dy, "_colsum", None)              # left by the GroupNorm backward that produced dy
            per_sample = cs[0] if (cs is not None and cs[1] == dy._version and cs[0].shape == (B, N)) \
                else colsum_raw(dy, B, HW, N)              # [B,N]
        if dbias is None and has_bias and ctx.needs_input_grad[2]:
            done = getattr(dy, "_biasdone", None)          # the GroupNorm backward already summed it into the sink
            twin = None
            if (done is not None and done[2] == dy._version and gvb is not None and
                    done[0].data_ptr() == gvb.data_ptr()):
                dbias, twin = _fresh(gvb), done[1]
            else:
                dbias = colsum_raw(per_sample, 1, B, N, out=_fresh(gvb).view(1, N) if gvb is not None else None,
                                   ld=per_sample.stride(0)).view(N)
            try:       # the shortcut layer that shares this dy (nin_shortcut + bias) needs the very same column sum
                dy._biasgrad = (dbias.view(N), dy._version, twin)   # (a separate view: autograd adopts `dbias` itself)
            except (AttributeError, RuntimeError):
                pass
        if cb_dim is not None and ctx.needs_input_grad[3]:
            dcb = per_sample if cb_dim == 2 else dy
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        assert gp is None or (dres is None and dcb is not dy)
        return dx, dw, dbias, dcb, dres


def conv3x3(x, w, bias=None, cbias=None, res=None):
    return Conv3x3Fn.apply(x, w, bias, cbias, res)


class _Ctx:
    """a plain attribute bag standing in for an autograd context (see GnConv3x3Fn.backward)"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


# ----------------------------------------------------------------------------- dense
def _tag_bias_twin(y, gvb, wanted):
    """A dense layer whose output becomes the residual input of a 3x3 convolution (nin_shortcut, ldm/model_vdm.py:652-656)
    has the same bias gradient as that convolution: the convolution passes this sink on (Conv3x3Fn.forward)."""
    if gvb is not None and wanted:
        y._bias_twin = (gvb, y._version)


def _dense_bias_grad(dy, M, N, gvb):
    """sum of dy over all rows; re-used from the convolution that consumed the same dy when there is one"""
    out = _fresh(gvb) if gvb is not None else None
    c = getattr(dy, "_biasgrad", None)
    if c is not None and c[1] == dy._version and c[0].numel() == N:
        if out is None:
            return c[0].view(N)
        if not (len(c) > 2 and c[2] is not None and c[2].data_ptr() == out.data_ptr()):   # else: written there already
            out.view(N).copy_(c[0].view(N))
        return out.view(N)
    return colsum_raw(dy.reshape(M, N), 1, M, N, out=out.view(1, N) if out is not None else None).view(N)


def linear_fast_ok(x, K1, K2, N1, N2):
    """per-pixel dense layers go through the f16x3 kernel (linear_f16x3.hip) when the shapes are image shaped"""
    return (CONV_MODE == "f16x3" and x.dim() == 3 and x.shape[1] == HW and K1 % 32 == 0 and K2 % 32 == 0 and
            N1 % 128 == 0 and N2 % 128 == 0)


def linear_pack(w, transpose, wmax=None):
    """w [K,N] -> packed operand of y = x @ w (transpose=False) or of dx = dy @ w^T (transpose=True)"""
    K, N = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    pre = _prepacked(w, int(transpose))
    if pre is not None:
        return pre
    if wmax is None:
        wmax = absmax_rows(w.reshape(1, -1))
    wp = torch.empty(lib.load().mulan_linear_pack_f16x3_bytes(K, N), device=w.device, dtype=torch.uint8)
    call("mulan_linear_pack_f16x3", ptr(w), ptr(wp), ptr(wmax), K, N, int(transpose), stream())
    return wp, wmax


def linear_f16x3_raw(x1, x2, wp, wmax, N1, N2, bias=None, res=None, planes=False):
    """[x1 | x2] @ W + bias + res -> (y1 [B,1024,N1], y2 [B,1024,N2] or None); x* are [B,1024,K*] pixel tensors.
    planes=True: also returns (xs, xmax): the split planes of [x1 | x2] and the per-image maxima they were scaled with
    (the input of linear_wgrad_planes_raw)."""
    B, K1 = x1.shape[0], x1.shape[-1]
    K2 = 0 if x2 is None else x2.shape[-1]
    M = B * HW
    y1 = torch.empty((B, HW, N1), device=x1.device, dtype=torch.float32)
    y2 = torch.empty((B, HW, N2), device=x1.device, dtype=torch.float32) if N2 else None
    m1 = cached_absmax(x1)
    m2 = cached_absmax(x2) if x2 is not None else None
    xs = torch.empty(M * (K1 + K2) * 4, device=x1.device, dtype=torch.uint8) if planes else None
    _timed("linear_f16x3_kernel", 2.0 * M * (K1 + K2) * (N1 + N2),
           lambda: call("mulan_linear_f16x3", ptr(x1), ptr(m1), ptr(x2), ptr(m2), K1, K2, ptr(wp), ptr(wmax), ptr(bias),
                        ptr(res), ptr(y1), ptr(y2), ptr(xs), N1, N2, M, HW, stream()))
    if planes:
        return y1, y2, xs, (m1 if m2 is None else torch.maximum(m1, m2))
    return y1, y2


def linear_wgrad_planes_raw(xs, xmax, dys, dymax, B, K, N, out=None):
    """dw[K,N] = [x1|x2]^T dy from the planes handed on by linear_f16x3_raw (xs) and by the convolution that consumed
    the same dy (dys)"""
    share = _share_chip()
    nbytes = lib.load().mulan_linear_wgrad_f16x3_planes_workspace(B, H, W, K, N, share)
    ws = torch.empty(nbytes // 4, device=xs.device, dtype=torch.float32)
    dw = out if out is not None else torch.empty((K, N), device=xs.device, dtype=torch.float32)
    _timed("linear_wgrad_f16x3_planes_kernel+slab_reduce", 2.0 * B * HW * K * N,
           lambda: call("mulan_linear_wgrad_f16x3_planes", ptr(xs), ptr(xmax), ptr(dys), ptr(dymax), ptr(dw), ptr(ws), B, H,
                        W, K, N, 0, share, stream()))
    return dw


def linear_wgrad_x32_raw(x1, x2, xmax, xmax2, dys, dymax, B, N, out=None):
    """dw[K1 + K2, N] = [x1 | x2]^T dy with x in fp32 (split while staged: the kernel is memory bound) and dy as the
    planes handed on by the convolution that consumed the same dy; xmax / xmax2 = the maxima of x1 / x2 (the kernel
    scales the concat with their elementwise max, as the forward kernel did: no separate launch for it).  Bit-identical to
    linear_wgrad_planes_raw on the planes the forward kernel would have written -- which it then need not write."""
    K1 = x1.shape[-1]
    K2 = 0 if x2 is None else x2.shape[-1]
    share = _share_chip()
    nbytes = lib.load().mulan_linear_wgrad_f16x3_x32_workspace(B, H, W, K1 + K2, N, share)
    ws = torch.empty(nbytes // 4, device=x1.device, dtype=torch.float32)
    dw = out if out is not None else torch.empty((K1 + K2, N), device=x1.device, dtype=torch.float32)
    _timed("linear_wgrad_f16x3_planes_kernel+slab_reduce", 2.0 * B * HW * (K1 + K2) * N,
           lambda: call("mulan_linear_wgrad_f16x3_x32", ptr(x1), ptr(x2), K1, K2, ptr(xmax), ptr(xmax2), ptr(dys), ptr(dymax), ptr(dw),
                        ptr(ws), B, H, W, N, 0, share, stream()))
    return dw


# the dense weight gradient reads the layer's fp32 input and splits it in its staging path (round 3) instead of taking
# planes the forward kernel wrote as a by-product; A/B switch: 0 = the round-2 plane hand-over
LINEAR_WGRAD_X32 = _os.environ.get("MULAN_LINEAR_WGRAD_X32", "1") == "1"


class LinearFn(torch.autograd.Function):
    """y[M,N] = x[M,K] @ w[K,N] + bias + res   (flax nn.Dense: y = x @ kernel + bias)"""

    @staticmethod
    def forward(ctx, x, w, bias, res):
        x, w = _c(x), _c(w)
        K, N = w.shape
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        ctx.fast = linear_fast_ok(x, K, 0, N, 0) and K % 128 == 0
        ctx.wmax = None
        if ctx.fast:
            wp, ctx.wmax = linear_pack(w, False)
            y, _ = linear_f16x3_raw(x, None, wp, ctx.wmax, N, 0, bias=_c(bias), res=_c(res))
        else:
            y = gemm_raw(x2, w, M, N, K, bias=_c(bias), R=None if res is None else _c(res).reshape(M, N))
        ctx.save_for_backward(x2, w)
        ctx.meta = (x.shape, bias is not None, res is not None)
        ctx.gv = (_gv(w), _gv(bias))
        y = y.view(*x.shape[:-1], N)
        _tag_bias_twin(y, ctx.gv[1], ctx.needs_input_grad[2])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        xshape, has_bias, has_res = ctx.meta
        K, N = w.shape
        M = x2.shape[0]
        dy = _c(dy)
        dy2 = dy.reshape(M, N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.fast:
                wpt, _ = linear_pack(w, True, ctx.wmax)
                dx = linear_f16x3_raw(view_keep_absmax(dy, -1, HW, N), None, wpt, ctx.wmax, K, 0)[0].view(xshape)
            else:
                dx = gemm_raw(dy2, w, M, K, N, transB=True).view(xshape)
        gvw, gvb = ctx.gv
        if ctx.needs_input_grad[1]:
            if _side_ok(gvw) and M >= 4096:          # (the small per-sample layers stay in line: nothing to overlap)
                dw = _fresh(gvw)
                _on_side(lambda: gemm_raw(x2, dy2, K, N, M, transA=True, out=dw), (x2, dy2))
            else:
                dw = gemm_raw(x2, dy2, K, N, M, transA=True, out=_fresh(gvw) if gvw is not None else None)
        if has_bias and ctx.needs_input_grad[2]:
            db = _dense_bias_grad(dy, M, N, gvb)
        dres = dy if (has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres


def linear(x, w, bias=None, res=None):
    return LinearFn.apply(x, w, bias, res)


class CondProjAllFn(torch.autograd.Function):
    """outs[g] = cond @ W[g] for all G FiLM projections of a U-Net at once (cond_proj of every ResnetBlock,
    ldm/model_vdm.py:639-641: they share their input): one batched GEMM forward, two backward, instead of three small
    latency-bound GEMMs per block."""

    @staticmethod
    def forward(ctx, cond, W):
        cond, W = _c(cond), _c(W)
        G, K, N = W.shape
        B = cond.shape[0]
        out = gemm_raw(cond, W, B, N, K, batch=G, sA=0, sB=K * N)            # [G,B,N]
        ctx.save_for_backward(cond, W)
        ctx.gv = _gv(W)
        return tuple(out[g] for g in range(G))

    @staticmethod
    @once_differentiable
    def backward(ctx, *douts):
        cond, W = ctx.saved_tensors
        G, K, N = W.shape
        B = cond.shape[0]
        zero = None
        parts = []
        for d in douts:
            if d is None:
                if zero is None:
                    zero = torch.zeros((B, N), device=cond.device, dtype=torch.float32)
                d = zero
            parts.append(d)
        dO = torch.stack(parts)                                              # [G,B,N]
        dcond = dW = None
        if ctx.needs_input_grad[0]:
            dcond = gemm_raw(dO, W, B, K, N, transB=True, batch=G, sA=B * N, sB=K * N).sum(0)
        if ctx.needs_input_grad[1]:
            gvw = ctx.gv
            dW = gemm_raw(cond, dO, K, N, B, transA=True, batch=G, sA=0, sB=B * N,
                          out=_fresh(gvw) if gvw is not None else None)
        return dcond, dW


class CondProjGroup:
    """The strided super-parameter [G,K,N] over G back-to-back cond_proj kernels (TrainState lays them out that way)
    and the per-forward cache of their outputs."""

    def __init__(self, weight):
        self.weight = weight
        self._cond = None
        self._key = None
        self._outs = None

    def clear(self):
        """drops the cached outputs (and with them the autograd graph they hold, including the AccumulateGrad node of
        the super-parameter, which remembers the stream it was created on)"""
        self._cond = self._key = self._outs = None

    def outputs(self, cond):
        import weakref
        key = (cond._version, torch.is_grad_enabled())
        if self._cond is None or self._cond() is not cond or self._key != key:
            self._outs = CondProjAllFn.apply(cond, self.weight)
            self._cond, self._key = weakref.ref(cond), key
        return self._outs


GROUP_COND_PROJ = _os.environ.get("MULAN_GROUP_COND_PROJ", "1") == "1"


def cond_proj(cond, w):
    """cond @ w for a ResnetBlock's FiLM projection; grouped over all blocks of the U-Net when the parameters come
    from a TrainState (per-sample conditioning only; the per-pixel ldm variant is an ordinary per-pixel dense layer)"""
    grp = getattr(w, "_group", None)
    if grp is None or cond.dim() != 2 or not GROUP_COND_PROJ:
        return linear(cond, w)
    return grp[0].outputs(cond)[grp[1]]


class Linear2Fn(torch.autograd.Function):
    """y = [x1 | x2] @ w + bias over a virtual channel concat (nin_shortcut on concat[h, skip],
    ldm/model_vdm.py:369,652-653) without materialising the concat."""

    @staticmethod
    def forward(ctx, x1, x2, w, bias):
        x1, x2, w = _c(x1), _c(x2), _c(w)
        K1, K2, N = x1.shape[-1], x2.shape[-1], w.shape[1]
        a1, a2 = x1.reshape(-1, K1), x2.reshape(-1, K2)
        M = a1.shape[0]
        ctx.fast = linear_fast_ok(x1, K1, K2, N, 0) and K1 % 128 == 0 and K2 % 128 == 0
        ctx.wmax = None
        ctx.xs = None
        if ctx.fast:      # one pass over both inputs, no intermediate
            wp, ctx.wmax = linear_pack(w, False)
            ctx.x32 = None
            if ctx.needs_input_grad[2] and LINEAR_WGRAD_X32 and K1 % 128 == 0 and K2 % 128 == 0 and N % 128 == 0:
                y = linear_f16x3_raw(x1, x2, wp, ctx.wmax, N, 0, bias=_c(bias))[0].view(M, N)
                ctx.x32 = (cached_absmax(x1), cached_absmax(x2))      # the maxima the forward kernel just used: [B, 16] each
            elif ctx.needs_input_grad[2] and (K1 + K2) % 128 == 0 and N % 128 == 0:
                y, _, ctx.xs, ctx.xsmax = linear_f16x3_raw(x1, x2, wp, ctx.wmax, N, 0, bias=_c(bias), planes=True)
                y = y.view(M, N)
            else:
                y = linear_f16x3_raw(x1, x2, wp, ctx.wmax, N, 0, bias=_c(bias))[0].view(M, N)
        else:
            y = gemm_raw(a1, w[:K1], M, N, K1, bias=_c(bias))
            y = gemm_raw(a2, w[K1:], M, N, K2, R=y, out=torch.empty_like(y))
        ctx.save_for_backward(a1, a2, w)
        ctx.shape = x1.shape[:-1]
        ctx.gv = (_gv(w), _gv(bias))
        y = y.view(*x1.shape[:-1], N)
        _tag_bias_twin(y, ctx.gv[1], ctx.needs_input_grad[3])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        a1, a2, w = ctx.saved_tensors
        K1, K2, N = a1.shape[1], a2.shape[1], w.shape[1]
        M = a1.shape[0]
        dy = _c(dy)
        dy2 = dy.reshape(M, N)
        if ctx.fast and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            wpt, _ = linear_pack(w, True, ctx.wmax)          # dy @ w^T, split into the two inputs' gradients on the way out
            dx1, dx2 = linear_f16x3_raw(view_keep_absmax(dy, -1, HW, N), None, wpt, ctx.wmax, K1, K2)
            dx1, dx2 = dx1.view(*ctx.shape, K1), dx2.view(*ctx.shape, K2)
        else:
            dx1 = gemm_raw(dy2, w[:K1], M, K1, N, transB=True).view(*ctx.shape, K1) if ctx.needs_input_grad[0] else None
            dx2 = gemm_raw(dy2, w[K1:], M, K2, N, transB=True).view(*ctx.shape, K2) if ctx.needs_input_grad[1] else None
        dw = None
        gvw, gvb = ctx.gv
        if ctx.needs_input_grad[2]:
            dw = _fresh(gvw) if gvw is not None else torch.empty_like(w)
            pl = getattr(dy, "_planes", None)
            x32 = getattr(ctx, "x32", None)
            if x32 is not None and pl is not None and pl[2] == dy._version and pl[0].numel() == M * N * 4:
                B_ = M // HW
                x1v, x2v = a1.view(B_, HW, K1), a2.view(B_, HW, K2)
                if _side_ok(gvw):
                    _on_side(lambda: linear_wgrad_x32_raw(x1v, x2v, x32[0], x32[1], pl[0], pl[1], B_, N, out=dw),
                             (x1v, x2v, x32[0], x32[1], pl[0], pl[1]))
                else:
                    linear_wgrad_x32_raw(x1v, x2v, x32[0], x32[1], pl[0], pl[1], B_, N, out=dw)
            elif ctx.xs is not None and pl is not None and pl[2] == dy._version and pl[0].numel() == M * N * 4:
                if _side_ok(gvw):
                    xs, xsmax = ctx.xs, ctx.xsmax
                    _on_side(lambda: linear_wgrad_planes_raw(xs, xsmax, pl[0], pl[1], M // HW, K1 + K2, N, out=dw),
                             (xs, xsmax, pl[0], pl[1]))
                else:
                    linear_wgrad_planes_raw(ctx.xs, ctx.xsmax, pl[0], pl[1], M // HW, K1 + K2, N, out=dw)
            elif _side_ok(gvw):
                def both():
                    gemm_raw(a1, dy2, K1, N, M, transA=True, out=dw[:K1])
                    gemm_raw(a2, dy2, K2, N, M, transA=True, out=dw[K1:])
                _on_side(both, (a1, a2, dy2))
            else:
                gemm_raw(a1, dy2, K1, N, M, transA=True, out=dw[:K1])
                gemm_raw(a2, dy2, K2, N, M, transA=True, out=dw[K1:])
        db = None
        if ctx.needs_input_grad[3]:
            db = _dense_bias_grad(dy, M, N, gvb)
        return dx1, dx2, dw, db


def linear2(x1, x2, w, bias):
    return Linear2Fn.apply(x1, x2, w, bias)


# ----------------------------------------------------------------------------- activations
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind, shift):
        x = _c(x)
        y = torch.empty_like(x)
        call("mulan_act_fwd", ptr(x), ptr(y), x.numel(), kind, float(shift), stream())
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        call("mulan_act_bwd", ptr(x), ptr(dy), ptr(dx), x.numel(), ctx.kind, stream())
        return dx, None, None


def silu(x):
    return ActFn.apply(x, 1, 0.0)


def softplus_shift(x, shift):
    return ActFn.apply(x, 2, shift)


# ----------------------------------------------------------------------------- group norm
def _seed_args(seed):
    """(seed by value, device pointer or None): a seed given as a 1-element int64 device tensor is read by the kernel
    when it runs (stream-ordered parameter: HIP-graph replay draws a fresh dropout mask per step)"""
    if torch.is_tensor(seed):
        return 0, seed
    return int(seed), None


GN_FUSED_REDUCE = _os.environ.get("MULAN_GN_FUSED_REDUCE", "1") == "1"   # A/B switch: 0 = separate mulan_colsum launches
_TICKETS = {}


def _gn_tickets(device):
    """arrival counters of the in-kernel reductions (zero between launches).  One array per (device, stream): launches
    on one stream are ordered and each leaves its counters zero; two streams must never share them."""
    key = (device, torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(16, device=device, dtype=torch.int32)
    return t


def reset_tickets():
    """zeroes the arrival counters (TrainState.zero_grad calls this once per step: a launch that died half-way in an
    earlier step must not leave a count behind)"""
    for t in _TICKETS.values():
        t.zero_()


def _gn_forward(ctx, x1, x2, gamma, beta, groups, eps, act, keep, seed, offset):
    x1, x2 = _c(x1), _c(x2)
    B, C1 = x1.shape[0], x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    y = torch.empty((B, HW, C1 + C2), device=x1.device, dtype=torch.float32)
    mean = torch.empty((B, groups), device=x1.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    # by-product for a following f16x3 convolution: the per-image maxima of y
    ymax = (torch.empty((B, MAX_PARTS), device=x1.device, dtype=torch.int32)
            if CONV_MODE == "f16x3" and (C1 + C2) // 32 <= MAX_PARTS else None)
    sv, sd = _seed_args(seed)
    call("mulan_groupnorm_fwd_dyn", ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), B,
         HW, groups, float(eps), int(act), float(keep), sv, int(offset), ptr(sd), ptr(ymax), stream())
    if ymax is not None:
        y._absmax = (ymax, y._version)
    ctx.save_for_backward(x1, x2, gamma, beta, mean, rstd)
    ctx.meta = (groups, int(act), float(keep), seed, int(offset))
    ctx.gv = (_gv(gamma), _gv(beta))
    bs = getattr(x1, "_bias_sink", None)       # left by the convolution that produced x1 (see Conv3x3Fn.forward)
    ctx.bias_sink = bs[:2] if (bs is not None and bs[2] == x1._version and bs[0].numel() == C1) else None
    return y, x1, x2


def _gn_backward(ctx, dy, add1=None, add2=None, planes_out=False, add1b=None):
    """dx1, dx2 (+ the gradients add1 / add2 that reach x1 / x2 through a skip path), dgamma, dbeta.  The written dx1
    carries its maxima and per-sample channel sums for the convolution in front (whose dy it is).
    planes_out: the caller vouches that dx1 is consumed ONLY by the f16x3 kernels of the convolution that produced x1
    (GnConv3x3Fn with x1_grad_planes): where the kernel variant applies (single input, no skip-path gradient, dy with
    its maxima) dx1 is then written as split planes and returned as a planes-only stand-in (_planes_only_grad)."""
    x1, x2, gamma, beta, mean, rstd = ctx.saved_tensors
    groups, act, keep, seed, offset = ctx.meta
    dy = _c(dy)
    B, C1 = x1.shape[0], x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    Ct = C1 + C2
    dymax_in = getattr(dy, "_absmax", None)
    if add1b is not None and add1 is None:
        add1, add1b = add1b, None
    if (planes_out and GN_FUSED_REDUCE and x2 is None and add1 is None and dymax_in is not None and
            dymax_in[1] == dy._version and C1 % 32 == 0 and C1 // 32 <= 16 and (C1 // groups) % 4 == 0 and
            32 % (C1 // groups) == 0 and B * HW * C1 * 4 < 2 ** 31):
        dxp = torch.empty(B * HW * C1 * 4, device=dy.device, dtype=torch.uint8)
        bound = torch.empty((B, MAX_PARTS), device=dy.device, dtype=torch.int32)
        parts = torch.empty((2, B, C1), device=dy.device, dtype=torch.float32)
        csum = torch.empty((B, C1), device=dy.device, dtype=torch.float32)
        sv, sd = _seed_args(seed)
        gvg, gvb = ctx.gv
        dgamma = _fresh(gvg) if gvg is not None else torch.empty(C1, device=dy.device, dtype=torch.float32)
        dbeta = _fresh(gvb) if gvb is not None else torch.empty(C1, device=dy.device, dtype=torch.float32)
        sink, sink2 = ctx.bias_sink if ctx.bias_sink is not None else (None, None)
        call("mulan_groupnorm_bwd_fused_planes", ptr(dy), ptr(dymax_in[0]), ptr(x1), C1, ptr(gamma), ptr(beta), ptr(mean),
             ptr(rstd), ptr(dxp), ptr(parts[0]), ptr(parts[1]), B, HW, groups, act, keep, sv, offset, ptr(sd), ptr(bound),
             ptr(csum), ptr(dgamma), ptr(dbeta), ptr(sink), ptr(sink2), ptr(_gn_tickets(dy.device)),
             ptr(getattr(ctx, "keepbits", None)), stream())
        dx1 = _planes_only_grad(x1.shape, dy.device, dxp, bound)
        if sink is not None:
            dx1._biasdone = (sink, sink2, dx1._version)
        dx1._colsum = (csum, dx1._version)
        return dx1, None, dgamma, dbeta
    dx1 = torch.empty_like(x1)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    parts = torch.empty((2, B, Ct), device=dy.device, dtype=torch.float32)     # dgamma / dbeta per-sample partials
    dgp, dbp = parts[0], parts[1]
    f16 = CONV_MODE == "f16x3"
    m1 = torch.empty((B, MAX_PARTS), device=dy.device, dtype=torch.int32) if f16 and C1 // 32 <= MAX_PARTS else None
    m2 = (torch.empty((B, MAX_PARTS), device=dy.device, dtype=torch.int32)
          if f16 and x2 is not None and C2 // 32 <= MAX_PARTS else None)
    csum = torch.empty((B, Ct), device=dy.device, dtype=torch.float32)
    sv, sd = _seed_args(seed)
    gvg, gvb = ctx.gv
    dgamma = _fresh(gvg) if gvg is not None else torch.empty(Ct, device=dy.device, dtype=torch.float32)
    dbeta = _fresh(gvb) if gvb is not None else torch.empty(Ct, device=dy.device, dtype=torch.float32)
    cpg = Ct // groups
    fused = (GN_FUSED_REDUCE and C1 % 32 == 0 and C2 % 32 == 0 and Ct // 32 <= 16 and cpg % 4 == 0 and 32 % cpg == 0)
    if fused:         # the sums over the samples (dgamma, dbeta, the bias gradient of the convolution in front) in-kernel
        sink, sink2 = ctx.bias_sink if ctx.bias_sink is not None else (None, None)
        call("mulan_groupnorm_bwd_fused", ptr(dy), ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
             ptr(dx1), ptr(dx2), ptr(dgp), ptr(dbp), B, HW, groups, act, keep, sv, offset, ptr(sd), ptr(m1), ptr(m2),
             ptr(_c(add1)), ptr(_c(add2)), ptr(_c(add1b)), ptr(csum), ptr(dgamma), ptr(dbeta), ptr(sink), ptr(sink2),
             ptr(_gn_tickets(dy.device)), stream())
        if sink is not None:
            dx1._biasdone = (sink, sink2, dx1._version)
    else:
        if add1b is not None:
            add1 = add1 + add1b
        call("mulan_groupnorm_bwd_dyn", ptr(dy), ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
             ptr(dx1), ptr(dx2), ptr(dgp), ptr(dbp), B, HW, groups, act, keep, sv, offset, ptr(sd), 0, ptr(m1), ptr(m2),
             ptr(_c(add1)), ptr(_c(add2)), ptr(csum), stream())
        if Ct % 4 == 0 and dgamma.data_ptr() % 16 == 0 and dbeta.data_ptr() % 16 == 0:
            call("mulan_colsum_pair", ptr(parts), ptr(dgamma), ptr(dbeta), B, Ct, stream())     # both sums, one launch
        else:
            colsum_raw(dgp, 1, B, Ct, out=dgamma.view(1, Ct))
            colsum_raw(dbp, 1, B, Ct, out=dbeta.view(1, Ct))
    if m1 is not None:
        dx1._absmax = (m1, dx1._version)
    if m2 is not None:
        dx2._absmax = (m2, dx2._version)
    dx1._colsum = (csum[:, :C1], dx1._version)          # a strided view when there is an x2: consumers take its row stride
    return dx1, dx2, dgamma, dbeta


class GroupNormFn(torch.autograd.Function):
    """y = dropout(act(GroupNorm([x1|x2])))  -> [B,1024,C1+C2]"""

    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, groups, eps, act, keep, seed, offset):
        return _gn_forward(ctx, x1, x2, gamma, beta, groups, eps, act, keep, seed, offset)[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return _gn_backward(ctx, dy) + (None,) * 6


class GroupNormSkipFn(torch.autograd.Function):
    """(y, s1, s2) with y as GroupNormFn and s1 / s2 aliases of x1 / x2 for the block's skip path (the ResnetBlock
    residual or nin_shortcut, ldm/model_vdm.py:652-656): the gradients that come back through s1 / s2 are added
    inside the GroupNorm backward kernel instead of by a separate accumulation pass."""

    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, groups, eps, act, keep, seed, offset):
        y, x1c, x2c = _gn_forward(ctx, x1, x2, gamma, beta, groups, eps, act, keep, seed, offset)
        s1 = x1c.view_as(x1c)
        s2 = x2c.view_as(x2c) if x2c is not None else None
        for src, dst in ((x1, s1), (x2, s2)):                 # the maxima a producer left on x stay valid for the alias
            c = getattr(src, "_absmax", None) if src is not None else None
            if c is not None and c[1] == src._version:
                dst._absmax = (c[0], dst._version)
        ctx.has2 = x2c is not None
        return (y, s1, s2) if ctx.has2 else (y, s1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, ds1, ds2=None):
        if dy is None:     # only the skip path was used downstream
            return (ds1, ds2) + (None,) * 8
        return _gn_backward(ctx, dy, ds1, ds2) + (None,) * 6


def group_norm(x1, x2, gamma, beta, *, groups=32, eps=1e-6, act=True, keep=1.0, seed=0, offset=0):
    return GroupNormFn.apply(x1, x2, gamma, beta, groups, eps, int(act), keep, seed, offset)   # seed: int or device tensor


def group_norm_skip(x1, x2, gamma, beta, *, groups=32, eps=1e-6, act=True, keep=1.0, seed=0, offset=0):
    """-> (y, s1, s2): use s1 / s2 instead of x1 / x2 on the skip path of the block (see GroupNormSkipFn)"""
    out = GroupNormSkipFn.apply(x1, x2, gamma, beta, groups, eps, int(act), keep, seed, offset)
    return out if len(out) == 3 else (out[0], out[1], None)


GN_CONV_PLANES = _os.environ.get("MULAN_GN_CONV_PLANES", "1") == "1"    # A/B switch: 0 = fp32 hand-over (two ops)
# Backward counterpart (round 3): the gradient a GroupNorm backward hands to the convolution in front of it (norm2 ->
# conv1 of a ResnetBlock) goes as split planes too -- the input-gradient convolution then runs on the plane-fed
# instantiation of the kernel (no split, no plane stores out of the MFMA kernel).  A/B switch: 0 = fp32 hand-over.
GRAD_PLANES = _os.environ.get("MULAN_GRAD_PLANES", "1") == "1"
# the dropout keep-bits of norm2 are stored by the forward kernel and re-used by the backward kernel (A/B: 0 = re-drawn)
KEEP_BITS = _os.environ.get("MULAN_KEEP_BITS", "1") == "1"
# GroupNorm normalised inside the convolution's patch fill (mulan_groupnorm_stats + mulan_conv3x3_fwd_f16x3_gn_in) where no
# dropout is drawn.  GN_FILL: wherever the convolution's weight needs no gradient (evaluators, sampler, the ODE
# evaluator's input-only differentiation) and the convolution has one 128-wide block of output channels: the normalised
# tensor never reaches HBM (with the statistics hand-over below: sampler step -15 %, ODE function evaluation -7.5 % at E = 128).  With N = 256 every input
# element is normalised by two blocks and the fill's arithmetic costs more than the GroupNorm pass it saves (dense
# evaluation at E = 256: +4.5 %), so those layers keep the plane hand-over; GN_FILL_MAX_N is that limit.
# GN_FILL_TRAIN (A/B switch, off: the train step is 1.2 % slower with it, profiles/DESIGN_r04.md 3.2): also in training, the convolution
# then stores the planes for its weight gradient.
GN_FILL = _os.environ.get("MULAN_GN_FILL", "1") == "1"
GN_FILL_MAX_N = int(_os.environ.get("MULAN_GN_FILL_MAX_N", "128"))
GN_FILL_TRAIN = _os.environ.get("MULAN_GN_FILL_TRAIN", "0") == "1"
# ... and the statistics handed from convolution to convolution: a GroupNorm-fed convolution leaves the partial sums of
# its output (per image, 8-row tile, channel quad) on the tensor, the next one forms mean / rstd from them in its prologue:
# no pass over the tensor between two convolutions of a forward-only chain.  A/B switch: 0 = mulan_groupnorm_stats in
# front of every convolution (bit-identical to the plane hand-over; with the hand-over the statistics agree to rounding).
GN_FILL_STATS = _os.environ.get("MULAN_GN_FILL_STATS", "1") == "1"
# training step: GroupNorm forward as the streaming kernel on the statistics the producing convolution left, where that
# kernel is the faster one (GnConv3x3Fn.forward): dropout layers and the 2 x 128-channel concat, at 32 <= images per launch
# <= 96 -- the per-GPU batches of an 8-GPU job.  Measured in the train step of BASELINE configs[2]
# (profiles/r05_gn_fwd_stream_in_step.log): 64 images 42.78 -> 42.50 ms, 32 images 28.52 -> 28.42, 16 images 23.42 -> 23.65,
# 128 images 78.98 -> 79.08 (what the kernel gains alone, 4 us per launch, the statistics in the convolution epilogue cost
# back).  MULAN_GN_FWD_STREAM=0: the slab kernel everywhere (the round-4 path); GN_FWD_STREAM_B: the batch window.
GN_FWD_STREAM = _os.environ.get("MULAN_GN_FWD_STREAM", "1") == "1"
GN_FWD_STREAM_B = tuple(int(v) for v in _os.environ.get("MULAN_GN_FWD_STREAM_B", "32,96").split(","))


def _gn_fwd_stream_on(B):
    return GN_FWD_STREAM and GN_FWD_STREAM_B[0] <= B <= GN_FWD_STREAM_B[1]


def _gn_stats_of(x, C):
    """the partial sums the convolution that produced x left on it: [B, row tiles (4, 8 or 16), C / 4, 2], or None"""
    c = getattr(x, "_gnstats", None) if x is not None else None
    if (c is not None and c[1] == x._version and c[0].dim() == 4 and c[0].shape[0] == x.shape[0] and
            c[0].shape[1] in (H // 8, H // 4, H // 2) and tuple(c[0].shape[2:]) == (C // 4, 2)):
        return c[0]
    return None


def gn_conv_ok(C1, C2, N, groups):
    """GroupNorm -> 3x3 convolution with the normalised tensor handed over as split fp16 planes (GnConv3x3Fn)"""
    Ct = C1 + C2
    cpg = Ct // groups
    return (GN_CONV_PLANES and CONV_MODE == "f16x3" and Ct % 128 == 0 and N % 128 == 0 and Ct // 32 <= MAX_PARTS and
            C1 % 32 == 0 and C2 % 32 == 0 and Ct % groups == 0 and cpg % 4 == 0 and 32 % cpg == 0)


class GnConv3x3Fn(torch.autograd.Function):
    """y = conv3x3(dropout(act(GroupNorm([x1|x2]))), w) + bias + cbias + res  (norm1 + swish -> conv1 and norm2 + swish +
    dropout -> conv2 of the ResnetBlock, ldm/model_vdm.py:622-656) as ONE autograd node, so that the normalised tensor
    never exists in fp32: the GroupNorm kernel writes it as the split fp16 operand planes of the f16x3 kernels (scaled by
    an a-priori bound, mulan_groupnorm_fwd_planes), the convolution copies them into LDS (no split, no plane stores out
    of the MFMA kernel: 5-13 % of a launch) and its weight-gradient kernel reads the same tensor.  With skip=True also
    returns the aliases s1 (, s2) of x1 (, x2) for the block's skip path, whose gradients are added inside the
    GroupNorm backward kernel (as GroupNormSkipFn)."""

    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, w, bias, cbias, res, groups, eps, act, keep, seed, offset, skip,
                x1_grad_planes=False):
        # x1_grad_planes: the caller vouches that x1 has no consumer besides this op (conv1's output inside a
        # ResnetBlock); if x1's producer accepts it (tag below), d x1 then travels as split planes only
        acc = getattr(x1, "_accepts_grad_planes", None)
        ctx.x1_grad_planes = bool(x1_grad_planes and not skip and x2 is None and acc is not None and acc == x1._version)
        # x1 is a block output that also feeds a skip connection (tee): this op's GroupNorm backward will add the gradient
        # that arrives through that connection (it is there first: the up path runs first in the backward pass)
        box = getattr(x1, "_grad_box", None)
        ctx.grad_box = None
        if box is not None and skip and not box.claimed and ctx.needs_input_grad[0]:
            box.claimed = True
            ctx.grad_box = box
        x1, x2, w = _c(x1), _c(x2), _c(w)
        B, C1 = x1.shape[0], x1.shape[-1]
        C2 = 0 if x2 is None else x2.shape[-1]
        Ct, N = C1 + C2, w.shape[-1]
        dev = x1.device
        bound = torch.empty((B, MAX_PARTS), device=dev, dtype=torch.int32)
        mean = torch.empty((B, groups), device=dev, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        fill = (float(keep) >= 1.0 and (C2 == 0 or C2 == C1) and Ct <= 512 and N <= GN_FILL_MAX_N and
                (GN_FILL_TRAIN if ctx.needs_input_grad[4] else GN_FILL))
        if fill:
            return GnConv3x3Fn._forward_fill(ctx, x1, x2, gamma, beta, w, bias, cbias, res, groups, eps, act, skip, bound,
                                             mean, rstd)
        ys = torch.empty(B * HW * Ct * 4, device=dev, dtype=torch.uint8)
        sv, sd = _seed_args(seed)
        # dropout layer whose backward will write planes (x1_grad_planes): keep the 4 keep-bits per float4 as drawn (2 MB at
        # B = 128, C = 128), so that the backward kernel does not repeat the Philox rounds
        keepbits = None
        want_bits = KEEP_BITS and float(keep) < 1.0 and ctx.x1_grad_planes and any(ctx.needs_input_grad[:5])
        # the convolutions that produced x1 (, x2) left the partial sums of the statistics: the streaming kernel (no
        # statistics phase, 62 registers) where it is the faster one -- dropout layers and the 2 x 128-channel concat
        # (profiles/r05_gn_stream_bench.log: 31.8 vs 36.0 us and 52.4 vs 55.6 us at B = 128; 128 channels without
        # dropout: 30.2 vs 27.7 us, the slab kernel stays), see GN_FWD_STREAM
        st1 = st2 = None
        if _gn_fwd_stream_on(B) and (float(keep) < 1.0 or C2 > 0) and Ct // 32 <= MAX_PARTS:
            st1 = _gn_stats_of(x1, C1)
            st2 = _gn_stats_of(x2, C2) if (st1 is not None and x2 is not None) else None
            if st1 is None or (x2 is not None and (st2 is None or st2.shape[1] != st1.shape[1])):
                st1 = st2 = None
        if st1 is not None:
            if want_bits:
                keepbits = torch.empty(B * (Ct // 32) * 1024, device=dev, dtype=torch.int32)
            call("mulan_groupnorm_fwd_stream", ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), None, ptr(ys), ptr(mean),
                 ptr(rstd), ptr(st1), ptr(st2), int(st1.shape[1]), B, HW, groups, float(eps), int(act), float(keep), sv,
                 int(offset), ptr(sd), ptr(bound), ptr(keepbits), stream())
        elif want_bits:
            keepbits = torch.empty(B * (Ct // 32) * 1024, device=dev, dtype=torch.int32)
            call("mulan_groupnorm_fwd_planes_keepbits", ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(ys), ptr(mean),
                 ptr(rstd), B, HW, groups, float(eps), int(act), float(keep), sv, int(offset), ptr(sd), ptr(bound),
                 ptr(keepbits), stream())
        else:
            call("mulan_groupnorm_fwd_planes", ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(ys), ptr(mean), ptr(rstd),
                 B, HW, groups, float(eps), int(act), float(keep), sv, int(offset), ptr(sd), ptr(bound), stream())
        ctx.keepbits = keepbits
        wp, wmax = _pack_weights(w, Ct, N, 0)
        y = torch.empty((B, HW, N), device=dev, dtype=torch.float32)
        ymax = torch.empty((B, MAX_PARTS), device=dev, dtype=torch.int32) if (H // 8) * (N // 128) <= MAX_PARTS else None
        mode = 0 if cbias is None else (1 if cbias.dim() == 2 else 2)
        bias_c, cb_c, res_c = _c(bias), _c(cbias), _c(res)
        ystats = None
        if _gn_fwd_stream_on(B) and N // 32 <= MAX_PARTS:   # by-product for the GroupNorm behind this convolution (see above)
            rows = lib.load().mulan_conv3x3_f16x3_tile_rows(B, H, N, int(ymax is not None))
            ystats = torch.empty((B, H // rows, N // 4, 2), device=dev, dtype=torch.float32)
        _timed("conv3x3_f16x3_kernel<planes_in>", 2.0 * B * HW * 9 * Ct * N,
               lambda: call("mulan_conv3x3_fwd_f16x3_planes_in_stats", ptr(ys), ptr(bound), ptr(wp), ptr(wmax), ptr(bias_c),
                            ptr(cb_c), mode, ptr(res_c), ptr(y), ptr(ymax), ptr(ystats), 1, B, H, W, Ct, N, stream()))
        if ystats is not None:
            y._gnstats = (ystats, y._version)
        return GnConv3x3Fn._finish_forward(ctx, x1, x2, gamma, beta, w, bias, cbias, res, mean, rstd, ys, bound, wmax, y, ymax,
                                           (groups, int(act), float(keep), seed, int(offset)), skip, mode)

    @staticmethod
    def _forward_fill(ctx, x1, x2, gamma, beta, w, bias, cbias, res, groups, eps, act, skip, bound, mean, rstd):
        """statistics pass + the convolution that normalises inside its patch fill (no dropout): bit for bit the result of
        the two-kernel path above"""
        B, C1 = x1.shape[0], x1.shape[-1]
        C2 = 0 if x2 is None else x2.shape[-1]
        Ct, N, dev = C1 + C2, w.shape[-1], x1.device
        want_planes = bool(ctx.needs_input_grad[4])
        st1 = _gn_stats_of(x1, C1) if GN_FILL_STATS and not want_planes else None
        st2 = _gn_stats_of(x2, C2) if (st1 is not None and x2 is not None) else None
        if st1 is None or (x2 is not None and (st2 is None or st2.shape[1] != st1.shape[1])):
            st1 = st2 = None
            call("mulan_groupnorm_stats", ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(bound), B,
                 HW, groups, float(eps), stream())
        ys = torch.empty(B * HW * Ct * 4 if want_planes else 0, device=dev, dtype=torch.uint8)
        ctx.keepbits = None
        wp, wmax = _pack_weights(w, Ct, N, 0)
        y = torch.empty((B, HW, N), device=dev, dtype=torch.float32)
        ymax = torch.empty((B, MAX_PARTS), device=dev, dtype=torch.int32) if (H // 8) * (N // 128) <= MAX_PARTS else None
        mode = 0 if cbias is None else (1 if cbias.dim() == 2 else 2)
        bias_c, cb_c, res_c = _c(bias), _c(cbias), _c(res)
        ystats = None
        if GN_FILL_STATS and not want_planes:     # one row of partial sums per row tile of THIS launch (8, 4 or 2 image rows)
            rows = lib.load().mulan_conv3x3_f16x3_tile_rows(B, H, N, int(ymax is not None))
            ystats = torch.empty((B, H // rows, N // 4, 2), device=dev, dtype=torch.float32)
        _timed("conv3x3_f16x3_kernel<gn_in>", 2.0 * B * HW * 9 * Ct * N,
               lambda: call("mulan_conv3x3_fwd_f16x3_gn_in", ptr(x1), ptr(x2), C1, C2, ptr(gamma), ptr(beta), ptr(mean),
                            ptr(rstd), groups, int(act), float(eps), ptr(bound), ptr(st1), ptr(st2),
                            0 if st1 is None else int(st1.shape[1]), ptr(wp), ptr(wmax),
                            ptr(bias_c), ptr(cb_c), mode, ptr(res_c), ptr(y), ptr(ymax), ptr(ystats),
                            ptr(ys) if want_planes else None, B, H, W, N, stream()))
        if ystats is not None:
            y._gnstats = (ystats, y._version)
        return GnConv3x3Fn._finish_forward(ctx, x1, x2, gamma, beta, w, bias, cbias, res, mean, rstd, ys, bound, wmax, y, ymax,
                                           (groups, int(act), 1.0, 0, 0), skip, mode)

    @staticmethod
    def _finish_forward(ctx, x1, x2, gamma, beta, w, bias, cbias, res, mean, rstd, ys, bound, wmax, y, ymax, meta, skip, mode):
        B, C1 = x1.shape[0], x1.shape[-1]
        Ct, N = C1 + (0 if x2 is None else x2.shape[-1]), w.shape[-1]
        if ymax is not None:
            y._absmax = (ymax, y._version)
        ctx.save_for_backward(x1, x2, gamma, beta, mean, rstd, ys, w, bound)
        ctx.meta = meta
        ctx.wmax = wmax
        ctx.has = (bias is not None, None if cbias is None else cbias.dim(), res is not None)
        ctx.gv_gn = (_gv(gamma), _gv(beta))
        ctx.gv_conv = (_gv(w), _gv(bias))
        bs = getattr(x1, "_bias_sink", None)       # left by the convolution that produced x1
        ctx.bias_sink = bs[:2] if (bs is not None and bs[2] == x1._version and bs[0].numel() == C1) else None
        if ctx.gv_conv[1] is not None and ctx.needs_input_grad[5]:
            twin = getattr(res, "_bias_twin", None) if res is not None else None
            y._bias_sink = (ctx.gv_conv[1], twin[0] if (twin is not None and twin[1] == res._version) else None, y._version)
        ctx.skip = bool(skip)
        ctx.has2 = x2 is not None
        if res is None and mode != 2 and grad_planes_eligible(Ct, N):
            y._accepts_grad_planes = y._version      # this op's backward understands a planes-only output gradient
        if not skip:
            return y
        s1 = x1.view_as(x1)
        s2 = x2.view_as(x2) if x2 is not None else None
        for src, dst in ((x1, s1), (x2, s2)):                 # the maxima a producer left on x stay valid for the alias
            c = getattr(src, "_absmax", None) if src is not None else None
            if c is not None and c[1] == src._version:
                dst._absmax = (c[0], dst._version)
        return (y, s1, s2) if ctx.has2 else (y, s1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, ds1=None, ds2=None):
        nones = (None,) * 8
        box = getattr(ctx, "grad_box", None)
        add1b = None
        if box is not None:
            add1b = box.take()
        if dy is None:     # only the skip path was used downstream
            if add1b is not None:
                ds1 = add1b if ds1 is None else ds1 + add1b
            return (ds1, ds2) + (None,) * 6 + nones
        x1, x2, gamma, beta, mean, rstd, ys, w, bound = ctx.saved_tensors
        need = ctx.needs_input_grad
        planes_out = ctx.x1_grad_planes and GRAD_PLANES and ds1 is None and add1b is None
        # (planes = False when the kernel needs no gradient -- the ODE evaluator differentiates with respect to the input
        # only: input gradient without plane output, no weight-gradient launch)
        conv = _Ctx(saved_tensors=(ys, w), planes=bool(need[4]), xmax=bound, wmax=ctx.wmax, has=ctx.has, gv=ctx.gv_conv,
                    needs_input_grad=(True, need[4], need[5], need[6], need[7]), want_dx_max=planes_out)
        dh, dw, dbias, dcb, dres = _conv3x3_backward(conv, dy)
        gn = _Ctx(saved_tensors=(x1, x2, gamma, beta, mean, rstd), meta=ctx.meta, gv=ctx.gv_gn,
                  bias_sink=ctx.bias_sink, keepbits=ctx.keepbits)
        dx1, dx2, dgamma, dbeta = _gn_backward(gn, dh, ds1, ds2, planes_out=planes_out, add1b=add1b)
        return (dx1, dx2, dgamma, dbeta, dw, dbias, dcb, dres) + nones


def gn_conv3x3(x1, x2, gamma, beta, w, bias=None, cbias=None, res=None, *, groups=32, eps=1e-6, act=True, keep=1.0,
               seed=0, offset=0, skip=False, x1_grad_planes=False):
    """-> y, or (y, s1, s2) with skip=True.  Falls back to group_norm(_skip) + conv3x3 where the plane hand-over does not
    apply (other arithmetic modes, channel counts off the 128 grid)."""
    C1 = x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    if gn_conv_ok(C1, C2, w.shape[-1], groups) and x1.is_cuda:
        out = GnConv3x3Fn.apply(x1, x2, gamma, beta, w, bias, cbias, res, groups, eps, int(act), keep, seed, offset, skip,
                                x1_grad_planes)
        if not skip:
            return out
        return out if len(out) == 3 else (out[0], out[1], None)
    if skip:
        h, s1, s2 = group_norm_skip(x1, x2, gamma, beta, groups=groups, eps=eps, act=act, keep=keep, seed=seed, offset=offset)
        return conv3x3(h, w, bias, cbias, res), s1, s2
    return conv3x3(group_norm(x1, x2, gamma, beta, groups=groups, eps=eps, act=act, keep=keep, seed=seed, offset=offset),
                   w, bias, cbias, res)


# ----------------------------------------------------------------------------- attention core
ATTN_F16X3 = True      # attention products on the split-operand kernels in f16x3 mode (else the exact-fp32 MFMA GEMM)


def _attn_fast_ok(q):
    B, S, C = q.shape
    return CONV_MODE == "f16x3" and ATTN_F16X3 and S == HW and C % 128 == 0 and B * S * S * 4 < 2 ** 31


def _const_max(B, value, device):
    """[B,16] maxima array holding an a-priori bound (softmax outputs are <= 1): the split scheme is accurate relative
    to each element, so a loose bound costs nothing until values fall 2^-24 below it"""
    import struct
    return torch.full((B, MAX_PARTS), struct.unpack('<i', struct.pack('<f', float(value)))[0], device=device,
                      dtype=torch.int32)


def _pack_batched(w, transpose, wmax):
    """w [B, K, N] (or [B, N, K] with transpose) -> packed per-image operands of linear_batched_raw"""
    B = w.shape[0]
    K, N = (w.shape[2], w.shape[1]) if transpose else (w.shape[1], w.shape[2])
    wp = torch.empty(B * K * N * 4, device=w.device, dtype=torch.uint8)
    call("mulan_linear_pack_f16x3_batched", ptr(w), ptr(wp), ptr(wmax), K, N, int(transpose), B, stream())
    return wp


def linear_batched_raw(x, xmax, wp, wmax, N, planes=False):
    """y[b] = x[b] @ W[b]: x [B, 1024, K]; returns y [B, 1024, N] (+ the split planes of x when planes=True)"""
    B, R, K = x.shape
    y = torch.empty((B, R, N), device=x.device, dtype=torch.float32)
    xs = torch.empty(B * R * K * 4, device=x.device, dtype=torch.uint8) if planes else None
    _timed("linear_f16x3_kernel(batched)", 2.0 * B * R * K * N,
           lambda: call("mulan_linear_f16x3_batched", ptr(x), ptr(xmax), K, ptr(wp), ptr(wmax), None, ptr(y), ptr(xs), N,
                        B * R, R, stream()))
    return (y, xs) if planes else y


def bmm_tn_planes_raw(xs, xmax, dys, dymax, B, C, N):
    """out[b] = x[b]^T @ dy[b] -> [B, C, N] from split planes"""
    out = torch.empty((B, C, N), device=xs.device, dtype=torch.float32)
    _timed("bmm_tn_f16x3_planes", 2.0 * B * HW * C * N,
           lambda: call("mulan_bmm_tn_f16x3_planes", ptr(xs), ptr(xmax), ptr(dys), ptr(dymax), ptr(out), B, H, W, C, N,
                        stream()))
    return out


ATTN_FUSED = True      # fused (flash-style) attention kernels, C = 128 and 256: S and P never reach HBM


def _attn_fused_ok(q):
    """forward and backward kernels exist for C = 128 and, since round 4, C = 256 (the ImageNet-32 width: the backward
    pass with its output channels split over blocks, attention_f16x3.hip); blockIdx.y carries the image"""
    B, S, C = q.shape
    return CONV_MODE == "f16x3" and ATTN_F16X3 and ATTN_FUSED and S == HW and C in (128, 256) and B <= 65535


def _attn_fused_fwd_only_ok(q, k, v):
    """kept for callers that want the bare forward kernel (no autograd node at all)"""
    B, S, C = q.shape
    return (_attn_fused_ok(q) and
            not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)))


def fused_attention_forward(q, k, v):
    """softmax((q / sqrt(C)) k^T) v by the fused forward kernel alone (no tape): C = 128 or 256"""
    q, k, v = _c(q), _c(k), _c(v)
    B, S, C = q.shape
    qm, km, vm = cached_absmax(q), cached_absmax(k), cached_absmax(v)
    qt, _ = _attn_packs(q, qm, True, False)
    kt, _ = _attn_packs(k, km, True, False)
    _, vn = _attn_packs(v, vm, False, True)
    o = torch.empty_like(q)
    lse = torch.empty((B, S), device=q.device, dtype=torch.float32)
    _timed("attn_f16x3_kernel<fwd>", 4.0 * B * S * S * C,
           lambda: call("mulan_attention_fwd_f16x3", ptr(qt), ptr(kt), ptr(vn), ptr(qm), ptr(km), ptr(vm), ptr(o), ptr(lse),
                        B, S, C, 1.0 / math.sqrt(C), stream()))
    return o


def _attn_packs(x, xmax, want_t=True, want_n=True):
    """("T" pack, "N" pack) of x [B, 1024, C] (C = 128 / 256) for the fused attention kernels, one pass over x"""
    B, S, C = x.shape
    xt = torch.empty(B * S * C * 4, device=x.device, dtype=torch.uint8) if want_t else None
    xn = torch.empty(B * S * C * 4, device=x.device, dtype=torch.uint8) if want_n else None
    call("mulan_attention_pack_f16x3", ptr(x), ptr(xmax), ptr(xt), ptr(xn), B, S, C, stream())
    return xt, xn


class FusedAttentionFn(torch.autograd.Function):
    """softmax((q / sqrt(C)) k^T) v on the fused f16x3 kernels (attention_f16x3.hip): the forward kernel keeps the online
    softmax state and writes o + the per-query log-sum-exp, the backward kernels recompute the probabilities from it.
    Saved for backward: the split packs of q, k, v (which replace the fp32 tensors), o and lse -- no [B, 1024, 1024]
    tensor."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = _c(q), _c(k), _c(v)
        B, S, C = q.shape
        alpha = 1.0 / math.sqrt(C)
        need = any(ctx.needs_input_grad)          # (grad mode is off inside Function.forward: this is the signal)
        qm, km, vm = cached_absmax(q), cached_absmax(k), cached_absmax(v)
        qt, qn = _attn_packs(q, qm, True, need)
        kt, kn = _attn_packs(k, km, True, need)
        vt, vn = _attn_packs(v, vm, need, True)
        o = torch.empty_like(q)
        lse = torch.empty((B, S), device=q.device, dtype=torch.float32)
        _timed("attn_f16x3_kernel<fwd>", 4.0 * B * S * S * C,
               lambda: call("mulan_attention_fwd_f16x3", ptr(qt), ptr(kt), ptr(vn), ptr(qm), ptr(km), ptr(vm), ptr(o),
                            ptr(lse), B, S, C, alpha, stream()))
        if need:
            ctx.save_for_backward(o, lse, qt, qn, kt, kn, vt, qm, km, vm)
        return o

    @staticmethod
    @once_differentiable
    def backward(ctx, do):
        o, lse, qt, qn, kt, kn, vt, qm, km, vm = ctx.saved_tensors
        do = _c(do)
        B, S, C = o.shape
        alpha = 1.0 / math.sqrt(C)
        dom = cached_absmax(do)
        delta = torch.empty((B, S), device=o.device, dtype=torch.float32)
        call("mulan_attention_delta", ptr(do), ptr(o), ptr(delta), B, S, C, stream())
        dm = absmax_rows(delta)
        dot, don = _attn_packs(do, dom)
        dq, dk, dv = torch.empty_like(o), torch.empty_like(o), torch.empty_like(o)
        _timed("attn_f16x3_kernel<bwd>", 14.0 * B * S * S * C,
               lambda: call("mulan_attention_bwd_f16x3", ptr(qt), ptr(qn), ptr(kt), ptr(kn), ptr(vt), ptr(dot), ptr(don),
                            ptr(qm), ptr(km), ptr(vm), ptr(dom), ptr(dm), ptr(lse), ptr(delta), ptr(dq), ptr(dk),
                            ptr(dv), B, S, C, alpha, stream()))
        return dq, dk, dv


class AttentionFn(torch.autograd.Function):
    """softmax((q / sqrt(C)) k^T) v for one head over 1024 positions (ldm/model_vdm.py:679-683,704-802).
    f16x3 mode: S = q k^T, O = P v, dP = dO v^T and dQ = dS k run on the split-operand kernel with one packed operand
    per image; the 1/sqrt(C) is folded into the softmax kernels."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = _c(q), _c(k), _c(v)
        B, S, C = q.shape
        alpha = 1.0 / math.sqrt(C)
        ctx.fast = _attn_fast_ok(q)
        if not ctx.fast:
            s = gemm_raw(q, k, S, S, C, transB=True, alpha=alpha, batch=B, sA=S * C, sB=S * C)
            p = torch.empty_like(s)
            call("mulan_softmax_fwd", ptr(s), ptr(p), B * S, S, stream())
            del s
            o = gemm_raw(p, v, S, C, S, batch=B, sA=S * S, sB=S * C)
            ctx.save_for_backward(q, k, v, p)
            return o.view(B, S, C)
        qm, km, vm = cached_absmax(q), cached_absmax(k), cached_absmax(v)
        s = linear_batched_raw(q, qm, _pack_batched(k, True, km), km, S)                   # q @ k^T
        p = torch.empty_like(s)
        call("mulan_softmax_scaled_fwd", ptr(s), ptr(p), B * S, S, alpha, stream())
        del s
        o = linear_batched_raw(p, _const_max(B, 1.0, q.device), _pack_batched(v, False, vm), vm, C)   # P @ v
        ctx.save_for_backward(q, k, v, p)
        ctx.aux = (km, vm)
        return o

    @staticmethod
    @once_differentiable
    def backward(ctx, do):
        q, k, v, p = ctx.saved_tensors
        do = _c(do)
        B, S, C = q.shape
        alpha = 1.0 / math.sqrt(C)
        if not ctx.fast:
            dv = gemm_raw(p, do, S, C, S, transA=True, batch=B, sA=S * S, sB=S * C).view(B, S, C)
            dp = gemm_raw(do, v, S, S, C, transB=True, batch=B, sA=S * C, sB=S * C)
            ds = torch.empty_like(dp)
            call("mulan_softmax_bwd", ptr(p), ptr(dp), ptr(ds), B * S, S, stream())
            del dp
            dq = gemm_raw(ds, k, S, C, S, alpha=alpha, batch=B, sA=S * S, sB=S * C).view(B, S, C)
            dk = gemm_raw(ds, q, S, C, S, transA=True, alpha=alpha, batch=B, sA=S * S, sB=S * C).view(B, S, C)
            return dq, dk, dv
        # The four products whose long operand is read once (S, O, dP, dQ) run on the split-operand kernel.  The two
        # transposed ones (dV = P^T dO, dK = dS^T q) would need P and dS as split planes (mulan_bmm_tn_f16x3_planes):
        # writing and re-reading those 4 B S^2-byte tensors costs as much as the exact-fp32 GEMM they would replace
        # (measured at B = 128: 203 us + 150 us of plane traffic against 338 us), so they stay on the fp32 MFMA GEMM.
        km, vm = ctx.aux
        ctx.aux = None
        dv = gemm_raw(p, do, S, C, S, transA=True, batch=B, sA=S * S, sB=S * C).view(B, S, C)      # P^T @ dO
        dp = linear_batched_raw(do, cached_absmax(do), _pack_batched(v, True, vm), vm, S)          # dO @ v^T
        ds = torch.empty_like(dp)
        rowmax = torch.empty((B, S), device=q.device, dtype=torch.float32)
        call("mulan_softmax_scaled_bwd", ptr(p), ptr(dp), ptr(ds), B * S, S, alpha, ptr(rowmax), stream())
        dsm = absmax_rows(rowmax)
        del dp
        dq = linear_batched_raw(ds, dsm, _pack_batched(k, False, km), km, C)                       # dS @ k
        dk = gemm_raw(ds, q, S, C, S, transA=True, batch=B, sA=S * S, sB=S * C).view(B, S, C)      # dS^T @ q
        return dq, dk, dv


def attention(q, k, v):
    if _attn_fused_ok(q):
        return FusedAttentionFn.apply(q, k, v)
    if _attn_fused_fwd_only_ok(q, k, v):
        return fused_attention_forward(q, k, v)
    return AttentionFn.apply(q, k, v)


# ----------------------------------------------------------------------------- embeddings
class FourierFn(torch.autograd.Function):
    """[z, sin(2^{6,7} 2pi z), cos(...), 0] -> 16 channels (ldm/model_vdm.py:341-343,812-829)"""

    @staticmethod
    def forward(ctx, z):
        z = _c(z)
        B = z.shape[0]
        out = torch.empty((B, HW, 16), device=z.device, dtype=torch.float32)
        call("mulan_fourier_fwd", ptr(z), ptr(out), B * HW, stream())
        ctx.save_for_backward(z)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (z,) = ctx.saved_tensors
        dout = _c(dout)
        dz = torch.empty_like(z)
        call("mulan_fourier_bwd", ptr(z), ptr(dout), ptr(dz), z.shape[0] * HW, 0, stream())
        return dz


def fourier_features(z):
    return FourierFn.apply(z)


class CondInputFn(torch.autograd.Function):
    """concat([timestep_embedding(t, E), conditioning]) -> [n, E + K]  (ldm/model_vdm.py:335-336).
    With rep > 1 the conditioning rows are broadcast `rep` times (ldm/ldm_unet.py:82-88)."""

    @staticmethod
    def forward(ctx, t, cond, E, rep):
        t, cond = _c(t), _c(cond)
        n, K = t.numel(), cond.shape[-1]
        out = torch.empty((n, E + K), device=t.device, dtype=torch.float32)
        call("mulan_temb_fwd", ptr(t), ptr(out), n, E, E + K, 0, stream())
        call("mulan_rowbcast", ptr(cond), ptr(out), n, K, rep, E + K, E, stream())
        ctx.save_for_backward(t)
        ctx.meta = (E, K, rep, cond.shape)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (t,) = ctx.saved_tensors
        E, K, rep, cshape = ctx.meta
        dout = _c(dout)
        n = t.numel()
        dt = dc = None
        if ctx.needs_input_grad[0]:
            dt = torch.empty_like(t)
            call("mulan_temb_bwd", ptr(t), ptr(dout), ptr(dt), n, E, E + K, 0, stream())
        if ctx.needs_input_grad[1]:
            dcr = dout[:, E:].contiguous()               # [n, K]
            dc = dcr if rep == 1 else colsum_raw(dcr, n // rep, rep, K)
            dc = dc.view(cshape)
        return dt, dc, None, None


def cond_input(t, cond, E, rep=1):
    return CondInputFn.apply(t, cond, E, rep)


class RowBcastFn(torch.autograd.Function):
    """y[r] = x[r // rep]: per-pixel broadcast of a per-sample vector (ldm/ldm_unet.py:85-87)"""

    @staticmethod
    def forward(ctx, x, rep):
        x = _c(x)
        n, K = x.shape
        y = torch.empty((n * rep, K), device=x.device, dtype=torch.float32)
        call("mulan_rowbcast", ptr(x), ptr(y), n * rep, K, rep, K, 0, stream())
        ctx.meta = (n, K, rep)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, K, rep = ctx.meta
        return colsum_raw(_c(dy), n, rep, K), None


def row_broadcast(x, rep):
    return RowBcastFn.apply(x, rep)


def encode_u8(x_u8):
    """EncDec.encode (ldm/model_vdm.py:274-280): uint8 [B, ...] -> fp32 in (-1, 1), same shape"""
    x_u8 = _c(x_u8)
    f = torch.empty(x_u8.shape, device=x_u8.device, dtype=torch.float32)
    call("mulan_encode_u8", ptr(x_u8), ptr(f), x_u8.numel(), stream())
    return f


# ----------------------------------------------------------------------------- MuLAN closed forms
class PolyGammaFn(torch.autograd.Function):
    """gamma_0, gamma_1, gamma_t, gamma'_t of NoiseSchedule_polynomial_fixedend
    (ldm/model_mulan_epsilon.py:514-555); gradients flow into (a, b, c) through gamma_t and gamma'_t."""

    @staticmethod
    def forward(ctx, a, b, c, t, gmin, gmax):
        a, b, c, t = _c(a), _c(b), _c(c), _c(t)
        B = a.shape[0]
        g0, g1, gt, gp = (torch.empty_like(a) for _ in range(4))
        call("mulan_poly_gamma_fwd", ptr(a), ptr(b), ptr(c), ptr(t), ptr(g0), ptr(g1), ptr(gt), ptr(gp), B, D,
             float(gmin), float(gmax), stream())
        ctx.save_for_backward(a, b, c, t)
        ctx.lim = (float(gmin), float(gmax))
        ctx.mark_non_differentiable(g0, g1)
        return g0, g1, gt, gp

    @staticmethod
    @once_differentiable
    def backward(ctx, _d0, _d1, dgt, dgp):
        a, b, c, t = ctx.saved_tensors
        B = a.shape[0]
        da, db, dc = (torch.empty_like(a) for _ in range(3))
        call("mulan_poly_gamma_bwd", ptr(a), ptr(b), ptr(c), ptr(t), ptr(_c(dgt)) if dgt is not None else None,
             ptr(_c(dgp)) if dgp is not None else None, ptr(da), ptr(db), ptr(dc), B, D, ctx.lim[0], ctx.lim[1],
             stream())
        return da, db, dc, None, None, None


def poly_gamma(a, b, c, t, gmin, gmax):
    return PolyGammaFn.apply(a, b, c, t, gmin, gmax)


class Expm1WeightFn(torch.autograd.Function):
    """w = T * expm1(gamma_t - gamma_s): discrete-time loss weight (ldm/model_mulan_epsilon.py:348-355)"""

    @staticmethod
    def forward(ctx, gt, gs, T):
        gt, gs = _c(gt), _c(gs)
        w = torch.empty_like(gt)
        call("mulan_expm1_weight_fwd", ptr(gt), ptr(gs), ptr(w), gt.numel(), float(T), stream())
        ctx.save_for_backward(gt, gs)
        ctx.T = float(T)
        return w

    @staticmethod
    @once_differentiable
    def backward(ctx, dw):
        gt, gs = ctx.saved_tensors
        dgt, dgs = torch.empty_like(gt), torch.empty_like(gs)
        call("mulan_expm1_weight_bwd", ptr(gt), ptr(gs), ptr(_c(dw)), ptr(dgt), ptr(dgs), gt.numel(), ctx.T, stream())
        return dgt, dgs, None


def expm1_weight(gt, gs, T):
    return Expm1WeightFn.apply(gt, gs, T)


class QSampleFn(torch.autograd.Function):
    """(z_t, mean gamma_t, loss_recon, loss_klz, var0, var1) from (x, gamma_0, gamma_1, gamma_t, eps0, eps)
    (ldm/model_mulan_velocity.py:208-236; ldm/model_vdm.py:119-151,274-303)."""

    @staticmethod
    def forward(ctx, x_u8, g0, g1, gt, eps0, eps):
        g0, g1, gt = _c(g0), _c(g1), _c(gt)
        B = x_u8.shape[0]
        per_elem = int(gt.numel() == B * D)
        dev = x_u8.device
        zt = torch.empty((B, D), device=dev, dtype=torch.float32)
        gbar, recon, klz, v0, v1 = (torch.empty(B, device=dev, dtype=torch.float32) for _ in range(5))
        call("mulan_qsample_fwd", ptr(x_u8), ptr(g0), ptr(g1), ptr(gt), per_elem, ptr(eps0), ptr(eps), ptr(zt),
             ptr(gbar), ptr(recon), ptr(klz), ptr(v0), ptr(v1), B, D, stream())
        ctx.save_for_backward(x_u8, g0, g1, gt, eps0, eps)
        ctx.per_elem = per_elem
        ctx.mark_non_differentiable(v0, v1)
        return zt, gbar, recon, klz, v0, v1

    @staticmethod
    @once_differentiable
    def backward(ctx, dzt, dgbar, drecon, dklz, _dv0, _dv1):
        x, g0, g1, gt, eps0, eps = ctx.saved_tensors
        B = x.shape[0]
        dev = x.device
        need0, need1 = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dzt = _c(dzt) if dzt is not None else torch.zeros((B, D), device=dev)
        dgt = torch.empty((B, D), device=dev, dtype=torch.float32)
        dg0 = torch.empty((B, D), device=dev, dtype=torch.float32) if need0 else None
        dg1 = torch.empty((B, D), device=dev, dtype=torch.float32) if need1 else None
        zero = None
        if (need0 and drecon is None) or (need1 and dklz is None):
            zero = torch.zeros(B, device=dev)
        call("mulan_qsample_bwd", ptr(x), ptr(g0), ptr(g1), ptr(gt), ctx.per_elem, ptr(eps0), ptr(eps), ptr(dzt),
             ptr(_c(dgbar)) if dgbar is not None else None,
             ptr(_c(drecon) if drecon is not None else zero) if need0 else None,
             ptr(_c(dklz) if dklz is not None else zero) if need1 else None,
             ptr(dgt), ptr(dg0), ptr(dg1), B, D, stream())
        if not ctx.per_elem:   # scalar schedule: reduce the per-element gradients to per-sample
            dgt = colsum_raw(dgt.view(B, D, 1), B, D, 1).view(gt.shape)
            dg0 = colsum_raw(dg0.view(B, D, 1), B, D, 1).view(g0.shape) if need0 else None
            dg1 = colsum_raw(dg1.view(B, D, 1), B, D, 1).view(g1.shape) if need1 else None
        else:
            dgt = dgt.view(gt.shape)
        return None, dg0, dg1, dgt, None, None


def qsample(x_u8, g0, g1, gt, eps0, eps):
    return QSampleFn.apply(x_u8, g0, g1, gt, eps0, eps)


class DiffLossFn(torch.autograd.Function):
    """loss_diff[B].  mode 0 velocity, 1 velocity_from_epsilon, 2 epsilon
    (ldm/model_mulan_velocity.py:246-260; ldm/model_mulan_epsilon.py:338-355; ldm/model_vdm.py:156-170)."""

    @staticmethod
    def forward(ctx, mode, x_u8, gt, gp, eps, zt, net):
        gt, gp, zt, net = _c(gt), _c(gp), _c(zt), _c(net)
        B = x_u8.shape[0]
        per_elem = int(gt.numel() == B * D)
        loss = torch.empty(B, device=x_u8.device, dtype=torch.float32)
        call("mulan_diffloss_fwd", mode, ptr(x_u8), ptr(gt), ptr(gp), per_elem, ptr(eps), ptr(zt), ptr(net),
             ptr(loss), B, D, stream())
        ctx.save_for_backward(x_u8, gt, gp, eps, zt, net)
        ctx.meta = (mode, per_elem)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        x, gt, gp, eps, zt, net = ctx.saved_tensors
        mode, per_elem = ctx.meta
        B = x.shape[0]
        dev = x.device
        dnet, dgt, dgp = (torch.empty((B, D), device=dev, dtype=torch.float32) for _ in range(3))
        dzt = torch.empty((B, D), device=dev, dtype=torch.float32) if mode == 1 else None
        call("mulan_diffloss_bwd", mode, ptr(x), ptr(gt), ptr(gp), per_elem, ptr(eps), ptr(zt), ptr(net),
             ptr(_c(dloss)), ptr(dnet), ptr(dgt), ptr(dgp), ptr(dzt), B, D, stream())
        if not per_elem:
            dgt = colsum_raw(dgt.view(B, D, 1), B, D, 1).view(gt.shape)
            dgp = colsum_raw(dgp.view(B, D, 1), B, D, 1).view(gp.shape)
        else:
            dgt, dgp = dgt.view(gt.shape), dgp.view(gp.shape)
        return None, None, dgt, dgp, None, (dzt.view(zt.shape) if dzt is not None else None), dnet.view(net.shape)


def diffusion_loss(mode, x_u8, gt, gp, eps, zt, net):
    return DiffLossFn.apply(mode, x_u8, gt, gp, eps, zt, net)


class TopKFn(torch.autograd.Function):
    """(embedding, kl_z) = relaxed top-k straight-through latent (ldm/model_mulan_velocity.py:78-120)."""

    @staticmethod
    def forward(ctx, logits, gnoise, k, tau):
        logits, gnoise = _c(logits), _c(gnoise)
        B, L = logits.shape
        emb, soft = torch.empty_like(logits), torch.empty_like(logits)
        kl = torch.empty(B, device=logits.device, dtype=torch.float32)
        nrm = torch.empty_like(kl)
        call("mulan_topk_fwd", ptr(logits), ptr(gnoise), ptr(emb), ptr(kl), ptr(soft), ptr(nrm), B, L, int(k),
             float(tau), stream())
        ctx.save_for_backward(logits, soft, nrm)
        return emb, kl

    @staticmethod
    @once_differentiable
    def backward(ctx, demb, dkl):
        logits, soft, nrm = ctx.saved_tensors
        B, L = logits.shape
        demb = _c(demb) if demb is not None else torch.zeros_like(logits)
        dkl = _c(dkl) if dkl is not None else torch.zeros(B, device=logits.device)
        dl = torch.empty_like(logits)
        call("mulan_topk_bwd", ptr(logits), ptr(soft), ptr(nrm), ptr(demb), ptr(dkl), ptr(dl), B, L, stream())
        return dl, None, None, None


def topk_embedding(logits, gnoise, k, tau=10.0):
    return TopKFn.apply(logits, gnoise, k, tau)


# ----------------------------------------------------------------------------- optimiser
def adamw_ema_step(p, g, m, v, ema, n_decay, lr, b1, b2, eps, wd, step, ema_rate, grad_scale=1.0, clip_norm=None, dyn=None):
    """one AdamW + EMA step on the flat buffers; clip_norm: optax.clip_by_global_norm in front (the factor is
    computed and applied on the device).  Returns the [2] device tensor (factor, gradient norm) when clipping.
    dyn: optional fp32 device tensor [lr, 1 - b1^t, 1 - b2^t] read by the kernel when it runs (graph replay); lr / step
    are ignored then."""
    if dyn is not None:
        out = None
        if clip_norm is not None:
            ws = torch.empty(lib.load().mulan_global_norm_clip_workspace() // 8, device=p.device, dtype=torch.float64)
            out = torch.empty(2, device=p.device, dtype=torch.float32)
            call("mulan_global_norm_clip", ptr(g), g.numel(), float(clip_norm), float(grad_scale), ptr(ws), ptr(out), stream())
        call("mulan_adamw_ema_step_dyn", ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), int(n_decay), float(b1),
             float(b2), float(eps), float(wd), float(ema_rate), float(grad_scale), ptr(out), ptr(dyn), stream())
        return out
    if clip_norm is None:
        call("mulan_adamw_ema_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), int(n_decay), float(lr),
             float(b1), float(b2), float(eps), float(wd), int(step), float(ema_rate), float(grad_scale), stream())
        return None
    ws = torch.empty(lib.load().mulan_global_norm_clip_workspace() // 8, device=p.device, dtype=torch.float64)
    out = torch.empty(2, device=p.device, dtype=torch.float32)
    call("mulan_global_norm_clip", ptr(g), g.numel(), float(clip_norm), float(grad_scale), ptr(ws), ptr(out), stream())
    call("mulan_adamw_ema_step_scaled", ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), int(n_decay), float(lr),
         float(b1), float(b2), float(eps), float(wd), int(step), float(ema_rate), float(grad_scale), ptr(out), stream())
    return out


# ----------------------------------------------------------------------------- ancestral sampler (eval only)
def ancestral_step(zt, net, gt, gs, eps, mode):
    """one reverse step z_t -> z_s (mulan_ancestral_step); mode 0: net is the velocity, 1: net is eps_hat; gt / gs per
    element ([B,3072]) or per sample ([B])"""
    zt, net, gt, gs, eps = _c(zt), _c(net), _c(gt), _c(gs), _c(eps)
    zs = torch.empty_like(zt)
    per = zt.numel() // gt.numel()
    call("mulan_ancestral_step", ptr(zt), ptr(net), ptr(gt), ptr(gs), ptr(eps), ptr(zs), zt.numel(), int(mode),
         0 if per == 1 else per, stream())
    return zs


def decode_argmax(z0, g0):
    """uint8 argmax over the 256 decoder bins at z_0 / sqrt(1 - sigmoid(g_0)) (VDM.generate_x, sample_softmax=False)"""
    z0, g0 = _c(z0), _c(g0)
    out = torch.empty(z0.shape, device=z0.device, dtype=torch.uint8)
    per = z0.numel() // g0.numel()
    call("mulan_decode_argmax", ptr(z0), ptr(g0), ptr(out), z0.numel(), 0 if per == 1 else per, stream())
    return out


def decode_logprobs(z, g0):
    """[..., 256] decoder log-probabilities of every element of z (EncDec.decode, ldm/model_vdm.py:282-296); g0 per
    element or one value per sample (broadcast over the trailing elements)"""
    z, g0 = _c(z.to(torch.float32)), _c(g0.to(torch.float32))
    out = torch.empty(tuple(z.shape) + (256,), device=z.device, dtype=torch.float32)
    per = z.numel() // g0.numel()
    call("mulan_decode_logprobs", ptr(z), ptr(g0), ptr(out), z.numel(), 0 if per == 1 else per, stream())
    return out


def decode_sample(z0, g0, seed, offset=0):
    """uint8 categorical draw over the 256 decoder bins (VDM.generate_x with sample_softmax=True)"""
    z0, g0 = _c(z0), _c(g0)
    out = torch.empty(z0.shape, device=z0.device, dtype=torch.uint8)
    per = z0.numel() // g0.numel()
    call("mulan_decode_sample", ptr(z0), ptr(g0), ptr(out), z0.numel(), 0 if per == 1 else per,
         int(seed) & (2**64 - 1), int(offset), stream())
    return out


def rowmean(x):
    x = _c(x)
    rows = x.shape[0]
    out = torch.empty(rows, device=x.device, dtype=torch.float32)
    call("mulan_rowmean", ptr(x), ptr(out), rows, x.numel() // rows, stream())
    return out


# ----------------------------------------------------------------------------- exact-likelihood ODE (eval only)
def topk_hard(logits, k):
    """(k-hot embedding of the plain logits, KL(softmax(logits) || uniform)): notebook_utils.logits_to_embeddings and
    _gumbel_kl_loss (ldm/notebook_utils.py:548-551, 222-229)"""
    logits = _c(logits)
    B, L = logits.shape
    emb = torch.empty_like(logits)
    kl = torch.empty(B, device=logits.device, dtype=torch.float32)
    call("mulan_topk_fwd", ptr(logits), None, ptr(emb), ptr(kl), None, None, B, L, int(k), 0.0, stream())
    return emb, kl


def ode_drift(net, x, gt, gp, hutch, mode, drift_out=None, want_cot=True):
    """(drift, cotangent for net) of VDM.reverse_ode; gamma per element or per sample"""
    net, x, gt, gp = _c(net), _c(x), _c(gt), _c(gp)
    hutch = _c(hutch) if hutch is not None else None
    drift = drift_out if drift_out is not None else torch.empty_like(net)
    cot = torch.empty_like(net) if (want_cot and hutch is not None) else None
    per = net.numel() // gt.numel()
    call("mulan_ode_drift", ptr(net), ptr(x), ptr(gt), ptr(gp), ptr(hutch) if hutch is not None else None, ptr(drift),
         ptr(cot), net.numel(), int(mode), 0 if per == 1 else per, stream())
    return drift, cot


def ode_div(gx, gt, gp, hutch, mode, div_out=None):
    gx, gt, gp, hutch = _c(gx), _c(gt), _c(gp), _c(hutch)
    B = gx.shape[0]
    d = gx.numel() // B
    div = div_out if div_out is not None else torch.empty(B, device=gx.device, dtype=torch.float32)
    call("mulan_ode_div", ptr(gx), ptr(gt), ptr(gp), ptr(hutch), ptr(div), B, d, int(mode),
         0 if gt.numel() == gx.numel() else d, stream())
    return div


def _coef(values):
    import ctypes
    return (ctypes.c_double * 7)(*(list(values) + [0.0] * (7 - len(values))))


def rk_combine(y, K, coef, h, out=None, out32=None):
    """out = y + h sum_j coef_j K[j] as float64 and / or fp32; K [7, n] fp32"""
    call("mulan_rk_combine", ptr(y), ptr(K), K.stride(0), _coef(coef), len(coef), float(h), ptr(out), ptr(out32),
         y.numel(), stream())


def rk_workspace(device):
    from .lib import load
    return torch.empty(load().mulan_rk_workspace_bytes() // 8, device=device, dtype=torch.float64)


def rk_error_norm(y, ynew, K, e, h, rtol, atol, ws, out):
    call("mulan_rk_error_norm", ptr(y), ptr(ynew), ptr(K), K.stride(0), _coef(e), float(h), float(rtol), float(atol),
         ptr(ws), ptr(out), y.numel(), stream())


def rk_init_norms(y0, f0, f1, rtol, atol, ws, out3):
    call("mulan_rk_init_norms", ptr(y0), ptr(f0), ptr(f1) if f1 is not None else None, float(rtol), float(atol), ptr(ws),
         ptr(out3), y0.numel(), stream())


def normal_logp(x):
    x = _c(x)
    rows = x.shape[0]
    out = torch.empty(rows, device=x.device, dtype=torch.float32)
    call("mulan_normal_logp", ptr(x), ptr(out), rows, x.numel() // rows, stream())
    return out


def noise(shape, seed, offset, device, kind, lo=-3.0, hi=3.0):
    """kind: 'uniform' U[0,1) | 'rademacher' +-1 | 'truncated_normal' on [lo, hi]"""
    out = torch.empty(shape, device=device, dtype=torch.float32)
    call("mulan_noise", ptr(out), out.numel(), int(seed) & (2**64 - 1), int(offset),
         {"uniform": 0, "rademacher": 1, "truncated_normal": 2, "gumbel": 3}[kind], float(lo), float(hi), stream())
    return out


def dequantize(x_u8, u, uniform, scale=1.0):
    """(data = encode(x) + noise as fp32, the re-quantised uint8 image the encoder sees)"""
    x_u8, u = _c(x_u8), _c(u)
    data = torch.empty(x_u8.shape, device=x_u8.device, dtype=torch.float32)
    rq = torch.empty_like(x_u8)
    call("mulan_dequantize", ptr(x_u8), ptr(u), ptr(data), ptr(rq), x_u8.numel(), 1 if uniform else 0, float(scale),
         stream())
    return data, rq
