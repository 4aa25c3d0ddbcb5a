"""TrainState on flat fp32 device buffers (mirror of ldm/train_state.py:33-119).

State = {step, params, ema_params, opt_state(mu, nu)}: every parameter leaf is a view into one
contiguous buffer (weight-decayed leaves first, `bias` leaves last -- the reference's decay mask,
ldm/experiment.py:139-146), so AdamW+EMA is one kernel launch (mulan_adamw_ema_step) and the
data-parallel gradient exchange is a few large all-reduces over slices of one flat gradient buffer.
"""
import torch

from . import ops
from .model import tree_leaves, tree_set


def _is_decayed(path):
    """decay_mask_fn of ldm/experiment.py:139-146: everything except leaves named 'bias' (and the
    never-occurring ('layer_norm'|'final_layer_norm', 'scale')) -- GroupNorm scales ARE decayed."""
    return path[-1] != 'bias' and tuple(path[-2:]) not in (('layer_norm', 'scale'), ('final_layer_norm', 'scale'))


_NET_ORDER = {'score_model': 0, 'gamma': 1, 'encoder_model': 2}      # backward visits them in this order


def grad_ready_rank(path):
    """Sort key: position of a parameter leaf in the order gradients are produced by one backward pass
    (ldm/model_vdm.py:335-386 and ldm/model_mulan_epsilon.py:101-154,531-538 read backwards)."""
    net = _NET_ORDER.get(path[0], 3)
    mod = path[1] if len(path) > 1 else ''
    sub = path[2] if len(path) > 2 else ''
    if net == 1:                                   # gamma MLP: the three output layers, then dense_2, dense_1
        pos = {'dense_out_c': 0, 'dense_out_b': 0, 'dense_out_a': 0, 'dense_2': 1, 'dense_1': 2}.get(mod, 3)
        return (net, pos, 0, mod)
    if sub == 'cond_proj' or mod in ('dense0', 'dense1'):
        return (net, 9, {'dense1': 1, 'dense0': 2}.get(mod, 0), '')          # after every block of the U-Net
    if mod == 'dense_layer_final':
        return (net, 0, 0, mod)
    if mod == 'conv_out':
        return (net, 1, 0, mod)
    if mod == 'GroupNorm_0':
        return (net, 2, 0, mod)
    kind, _, name = mod.partition('.')
    idx = int(name.rsplit('_', 1)[1]) if '_' in name and name.rsplit('_', 1)[1].isdigit() else 0
    if kind == 'up':
        return (net, 3, -idx, name)
    if kind == 'mid':
        return (net, 4, -idx, name)
    if kind == 'down':
        return (net, 5, -idx, name)
    if mod == 'conv_in':
        return (net, 6, 0, mod)
    return (net, 8, 0, mod)


class TrainState:
    def __init__(self, apply_fn, template, device, optimizer_args=None):
        self.apply_fn = apply_fn
        leaves = list(tree_leaves(template))
        for path, _ in leaves:   # same assertion as ldm/experiment.py:165,168
            assert path[0] in {'encoder_model', 'score_model', 'gamma'}, path
        decayed = [(p, v) for p, v in leaves if _is_decayed(p)]
        plain = [(p, v) for p, v in leaves if not _is_decayed(p)]
        # Both classes are laid out in the order their gradients become ready during backward (reverse execution order:
        # score U-Net from conv_out back to conv_in, its conditioning MLP, the gamma MLP, the encoder), so that
        # parallel.GradReducer's contiguous buckets complete -- and start their all-reduce -- one after the other while
        # backward is still running.  The FiLM projection kernels (cond_proj) of one U-Net share their input: they get one
        # common rank (their batched weight gradient is formed once all blocks are done, next to the conditioning MLP)
        # and therefore sit back to back, so that one strided view [G, K, N] over the flat buffers serves a single
        # batched GEMM (ops.cond_proj).
        decayed.sort(key=lambda pv: grad_ready_rank(pv[0]))
        plain.sort(key=lambda pv: grad_ready_rank(pv[0]))
        self.layout = []   # (path, offset, shape)
        off = 0
        self.n_decay = 0
        for i, (path, v) in enumerate(decayed + plain):
            n = v.numel()
            self.layout.append((path, off, tuple(v.shape)))
            off += (n + 3) // 4 * 4          # 16-byte aligned leaves
            if i == len(decayed) - 1:
                self.n_decay = off
        self.numel = off
        self.device = device
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros_like(self.flat)
        self.ema = torch.zeros_like(self.flat)
        self.mu = torch.zeros_like(self.flat)
        self.nu = torch.zeros_like(self.flat)
        self.step = 0
        self.opt = dict(b1=0.9, b2=0.99, eps=1e-8, weight_decay=0.01)
        if optimizer_args:
            self.opt.update(optimizer_args)
        self.params = self._views(self.flat, requires_grad=True)
        with torch.no_grad():
            for path, off_, shape in self.layout:
                src = template
                for k in path:
                    src = src[k]
                self.flat[off_:off_ + src.numel()].copy_(src.reshape(-1).to(device))
        self.ema.copy_(self.flat)            # ema_params = deepcopy(params), ldm/train_state.py:110
        self.ema_params = self._views(self.ema, requires_grad=False)
        self._leaves = []
        for (path, off_, shape), (_, leaf) in zip(self.layout, tree_leaves_in_layout(self.params, self.layout)):
            n = leaf.numel()
            # gradient sink: backward kernels write this leaf's gradient straight into the flat buffer
            # (mulan_amd.ops reads `_gview`), autograd then adopts the view as .grad without an extra add
            leaf._gview = self.grad[off_:off_ + n].view(shape)
            self._leaves.append(leaf)
        self._build_groups()

    def _build_groups(self):
        """one strided super-parameter per run of equally shaped, contiguous cond_proj kernels (see __init__)"""
        self._supers = []
        run = []

        def flush_params():
            if len(run) >= 2:
                (path0, off0, shape), _ = run[0]
                G, n = len(run), run[0][1].numel()
                w = self.flat[off0:off0 + G * n].view(G, *shape).detach().requires_grad_(True)
                w._gview = self.grad[off0:off0 + G * n].view(G, *shape)
                grp = ops.CondProjGroup(w)
                for i, (_, leaf) in enumerate(run):
                    leaf._group = (grp, i)
                self._supers.append((w, off0, G * n, [leaf for _, leaf in run]))
            run.clear()

        def scan(leaves, flush):
            for entry, leaf in zip(self.layout, leaves):
                path, off_, shape = entry
                ok = path[-2:] == ('cond_proj', 'kernel') and leaf.numel() % 4 == 0
                if ok and run and (run[-1][0][0][0] != path[0] or run[-1][0][2] != shape or
                                   run[-1][0][1] + run[-1][1].numel() != off_):
                    flush()
                if ok:
                    run.append((entry, leaf))
                else:
                    flush()
            flush()

        scan(self._leaves, flush_params)
        # ... and the same runs of the EMA parameters (evaluators, sampler, ODE likelihood: forward-only, no gradient
        # sink): one batched GEMM per U-Net pass instead of a split-k GEMM + reduction per ResnetBlock
        self._ema_groups = []

        def flush_ema():
            if len(run) >= 2:
                (path0, off0, shape), _ = run[0]
                G, n = len(run), run[0][1].numel()
                grp = ops.CondProjGroup(self.ema[off0:off0 + G * n].view(G, *shape))
                for i, (_, leaf) in enumerate(run):
                    leaf._group = (grp, i)
                self._ema_groups.append(grp)
            run.clear()

        scan([leaf for _, leaf in tree_leaves_in_layout(self.ema_params, self.layout)], flush_ema)

    def drop_graph_refs(self):
        """forget every cached autograd graph (the per-forward cache of the grouped FiLM projections): needed before a
        train step is captured on another stream, so that no gradient-accumulation node of the old stream survives"""
        for _, _, _, leaves in self._supers:
            for leaf in leaves:
                leaf._group[0].clear()
        for grp in self._ema_groups:
            grp.clear()

    def reducer_leaves(self):
        """(tensor, offset, numel) per gradient-carrying tensor for parallel.GradReducer: grouped leaves are
        represented by their super-parameter (that is where autograd delivers their gradient)"""
        grouped = {id(l): None for _, _, _, ls in self._supers for l in ls}
        out = [(leaf, off, leaf.numel()) for (path, off, shape), leaf in zip(self.layout, self._leaves)
               if id(leaf) not in grouped]
        out += [(w, off0, n) for w, off0, n, _ in self._supers]
        return out

    @classmethod
    def create(cls, *, apply_fn, variables, device, optimizer_args=None):
        return cls(apply_fn, variables["params"] if "params" in variables else variables, device, optimizer_args)

    def _views(self, flat, requires_grad):
        tree = {}
        for path, off, shape in self.layout:
            n = 1
            for s in shape:
                n *= s
            v = flat[off:off + n].view(shape)
            if requires_grad:
                v = v.detach().requires_grad_(True)
            d = tree
            for k in path[:-1]:
                d = d.setdefault(k, {})
            d[path[-1]] = v
        return tree

    def param_packer(self, which="params"):
        """f16x3 mode: the once-per-step weight preparation of all eligible leaves (ops.ParamPacker), else None.
        which = "params" (training) or "ema" (the tree evaluators / the sampler run on)"""
        from . import ops
        if ops.CONV_MODE != "f16x3" or not self.flat.is_cuda:
            return None
        attr = "_packer" if which == "params" else "_packer_ema"
        if getattr(self, attr, None) is None:
            tree, flat = (self.params, self.flat) if which == "params" else (self.ema_params, self.ema)
            leaves = [(leaf, off) for (path, off, shape), (_, leaf) in
                      zip(self.layout, tree_leaves_in_layout(tree, self.layout))]
            setattr(self, attr, ops.ParamPacker(flat, leaves))
        return getattr(self, attr)

    def zero_grad(self):
        """Call before each backward: clears the flat buffer and detaches stale .grad handles."""
        self.grad.zero_()
        ops.reset_tickets()
        for leaf in self._leaves:
            leaf.grad = None
        for w, _, _, _ in self._supers:
            w.grad = None

    def collect_grads(self):
        """Call after backward: any gradient that did not land in the flat buffer (ops without a sink, or a
        copy made by autograd) is copied in, so `self.grad` is complete for the all-reduce / optimizer."""
        for leaf in self._leaves + [w for w, _, _, _ in self._supers]:
            g = leaf.grad
            if g is not None and g.data_ptr() != leaf._gview.data_ptr():
                if getattr(leaf, "_group", None) is not None:      # a grouped leaf used on its own as well
                    leaf._gview.add_(g)
                else:
                    leaf._gview.copy_(g)
                leaf.grad = leaf._gview

    def apply_gradients(self, *, lr, ema_rate, grad_scale=1.0, clip_norm=None, dyn=None, count_step=True):
        """TrainState.apply_gradients (ldm/train_state.py:70-102) on the flat gradient buffer; clip_norm = the optional
        optimizer.gradient_clip_norm (optax.clip_by_global_norm in front of AdamW, ldm/experiment.py:176-178).
        dyn: device tensor [lr, 1 - b1^t, 1 - b2^t] for the graph-captured step (see dynamic_scalars); count_step=False
        leaves the host step counter to the caller (a captured launch is replayed, not re-run)."""
        if count_step:
            self.step += 1
        o = self.opt
        self.last_clip = ops.adamw_ema_step(self.flat, self.grad, self.mu, self.nu, self.ema, self.n_decay, lr, o["b1"],
                                            o["b2"], o["eps"], o["weight_decay"], max(1, self.step), ema_rate, grad_scale,
                                            clip_norm=clip_norm, dyn=dyn)
        return self

    def dynamic_scalars(self, lr, count):
        """[lr, 1 - b1^count, 1 - b2^count, 0]: what mulan_adamw_ema_step derives on the host from (lr, step)"""
        import numpy as np
        o = self.opt      # (b1, b2 reach the kernel launcher as C floats: the same rounding here keeps replay bit-identical)
        b1, b2 = float(np.float32(o["b1"])), float(np.float32(o["b2"]))
        return [float(lr), 1.0 - b1 ** count, 1.0 - b2 ** count, 0.0]

    # -- checkpoint form: {step, params, ema_params, opt_state} (ldm/train_state.py:62-68)
    def state_dict(self):
        def tree_of(flat):
            out = {}
            for path, off, shape in self.layout:
                n = 1
                for s in shape:
                    n *= s
                d = out
                for k in path[:-1]:
                    d = d.setdefault(k, {})
                d[path[-1]] = flat[off:off + n].view(shape).detach().cpu().clone()
            return out
        return {"step": self.step, "params": tree_of(self.flat), "ema_params": tree_of(self.ema),
                "opt_state": {"mu": tree_of(self.mu), "nu": tree_of(self.nu)}}

    def load_state_dict(self, sd, strict=True):
        def load(flat, tree):
            for path, off, shape in self.layout:
                src = tree
                try:
                    for k in path:
                        src = src[k]
                except (KeyError, TypeError):
                    if strict:
                        raise KeyError("/".join(path))
                    continue
                src = torch.as_tensor(src, dtype=torch.float32)
                if path[-2:] == ("conv_in", "kernel") and src.shape[2] == 15:
                    src = torch.cat([src, torch.zeros(3, 3, 1, src.shape[3])], dim=2)
                flat[off:off + src.numel()].copy_(src.reshape(-1).to(flat.device))
        with torch.no_grad():
            if "params" in sd:
                load(self.flat, sd["params"])
            if "ema_params" in sd:
                load(self.ema, sd["ema_params"])
            opt = sd.get("opt_state")
            if isinstance(opt, dict) and "mu" in opt:
                load(self.mu, opt["mu"])
                load(self.nu, opt["nu"])
            if "step" in sd:
                self.step = int(sd["step"])


def tree_leaves_in_layout(tree, layout):
    for path, _, _ in layout:
        v = tree
        for k in path:
            v = v[k]
        yield path, v
