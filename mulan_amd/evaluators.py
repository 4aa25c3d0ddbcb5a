"""Variational-bound BPD evaluators (mirror of ldm/notebook_utils.py:28-39,157-191).

dense : per test image, a batch of `n_timesteps` copies -> antithetic t covers [0,1) with spacing
        1/n_timesteps, i.e. a Riemann estimate of the diffusion-loss integral; the SAME rng key for every
        image (PRNGKey(0), notebook_utils.py:178).  Images are independent, so under torchrun the test set
        is sharded by index across ranks and (sum bpd, count) is all-reduced once at the end.
sparse: batches of batch_size_eval distinct images, same fixed key (notebook_utils.py:157-173).
"""
import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist

from . import checkpoint as ckpt_lib
from . import data as dataset
from . import parallel
from .experiment import Experiment_VDM
from .rng import PRNGKey


class Experiment_Colab(Experiment_VDM):
    """Experiment_VDM + EMA parameters restored from `<dir>/ckpt-<N>` (ldm/notebook_utils.py:28-39)."""

    def __init__(self, config, checkpoint_dir, checkpoint_num=None):
        super().__init__(config)
        if checkpoint_num is None:
            sd = ckpt_lib.restore_dict(checkpoint_dir)
        else:
            sd = ckpt_lib.restore_dict(os.path.join(checkpoint_dir, f'ckpt-{checkpoint_num}'))
        self.state.load_state_dict({"ema_params": sd["ema_params"]}, strict=True)
        self.orig_params = self.state.ema_params
        self.params = self.orig_params
        self.rng, sample_rng = self.rng.split()
        self.rngs = {'sample': sample_rng}

    # ---- samplers of the notebook front end (ldm/notebook_utils.py:54-135) --------------------------------------
    def _embedding_samples(self, embedding, rng, T):
        """T ancestral steps of model.conditional_sample under a fixed [B, 50] embedding, then generate_x"""
        from . import ops  # noqa: F401
        B = embedding.shape[0]
        rng = rng.fold_in(self.rank)
        rng, sample_rng = rng.split()
        packer = self.state.param_packer("ema")
        with torch.no_grad():
            if packer is not None:
                packer.refresh()
            try:
                z = sample_rng.normal((B, 3072), self.device)
                conditioning = torch.zeros(B, dtype=torch.uint8, device=self.device)
                coeffs = self.model.sample_coefficients(self.params, embedding)
                step = self.model.reverse_stepper(self.params, B, self.device, embedding, conditioning, coeffs, T)
                for i in range(T):
                    z = step(i, z, rng)
                samples = self.model.generate_x(self.params, z, rng=rng.fold_in(T))
            finally:
                if packer is not None:
                    packer.invalidate()
        return parallel.all_gather_tensor(samples)

    def sample_conditionally(self, embedding, T=1000):
        """Experiment_Colab.sample_conditionally: an image grid sampled under one 50-dim k-hot embedding"""
        from . import ops  # noqa: F401
        B = self.eval_iter.local
        emb = torch.as_tensor(embedding, dtype=torch.float32, device=self.device).reshape(1, -1)
        assert emb.shape[1] == 50
        samples = self._embedding_samples(emb.expand(B, 50).contiguous(), self.rng, T)
        return ckpt_lib.generate_image_grids(samples).astype(np.uint8)

    def sample_randomly(self, T=1000):
        """Experiment_Colab.sample_randomly: every image under the hard top-15 embedding of its own random logits"""
        from . import ops
        B = self.eval_iter.local
        _, embeddings_rng = self.rng.fold_in(self.rank).split()
        emb, _ = ops.topk_hard(embeddings_rng.normal((B, 50), self.device), 15)
        samples = self._embedding_samples(emb, self.rng, T)
        return ckpt_lib.generate_image_grids(samples).astype(np.uint8)

    def test(self, loader):
        """Experiment_Colab.test: mean of the eval scalars over a loader"""
        eval_metrics = []
        for eval_step, batch in enumerate(loader):
            m = self.p_eval_step(self.params, batch, eval_step)
            eval_metrics.append({k: float(v) for k, v in m['scalars'].items()})
        return {k: float(np.mean([m[k] for m in eval_metrics])) for k in eval_metrics[0]}


@contextlib.contextmanager
def _packed_weights(experiment):
    """The evaluated weights are constant over an evaluation: their maxima and split fp16 operands are prepared once
    (ParamPacker, two launches) instead of per layer and call."""
    st = experiment.state
    params = getattr(experiment, "orig_params", None)
    which = "ema" if params is st.ema_params else ("params" if params is st.params else None)
    packer = st.param_packer(which) if which else None
    if packer is not None:
        packer.refresh()
    try:
        yield
    finally:
        if packer is not None:
            packer.invalidate()


def _reduce_mean(total, count, device):
    if parallel.world_size() > 1:
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total, count = float(t[0]), int(t[1])
    return total / max(count, 1), count


def _dense_partial(experiment, config, n_timesteps, rank, world, max_images=0):
    """(sum of per-image BPDs, image count) of the images rank `rank` of `world` evaluates: image i goes to rank
    i % world; every image is a batch of n_timesteps copies under the SAME key PRNGKey(0) (notebook_utils.py:176-191)"""
    loader = dataset.create_one_time_eval_dataset(config, 1, experiment.device, rank, world)
    rng = PRNGKey(0)
    total, count = 0.0, 0
    with _packed_weights(experiment):
        for eval_step, batch in enumerate(loader):
            if max_images and eval_step * world + rank >= max_images:
                break
            images = batch['images'].reshape(1, 32, 32, 3).expand(n_timesteps, 32, 32, 3).contiguous()
            tiled = {'images': images, 'labels': batch['labels'].expand(n_timesteps),
                     'conditioning': torch.zeros(n_timesteps, dtype=torch.uint8, device=experiment.device)}
            with torch.no_grad():
                bpd, _ = experiment.loss_fn(experiment.orig_params, tiled, eval_step, rng=rng, is_train=False,
                                            same_image=True)
            total += float(bpd)
            count += 1
            if count % 100 == 0 and rank == 0:
                print(f'eval_step {count} cum_avg_bpd {total / count} ')
    return total, count


def eval_bpd_dense_sampling(experiment, config, n_timesteps=128, max_images=0):
    total, count = _dense_partial(experiment, config, n_timesteps, experiment.rank, experiment.world, max_images)
    mean, n = _reduce_mean(total, count, experiment.device)
    if experiment.rank == 0:
        print('Num eval steps:', n)
    return mean


def _sparse_partial(experiment, config, rank, world, max_images=0):
    """(sum of per-batch BPDs, batch count): batches of config.training.batch_size_eval DISTINCT images, the reference's
    global batch on every rank that evaluates one (the antithetic time grid spans the whole batch,
    notebook_utils.py:157-173), whole batches dealt round-robin to the ranks"""
    batch_size = config.training.batch_size_eval
    loader = dataset.create_one_time_eval_dataset(config, batch_size, experiment.device, rank, world)
    rng = PRNGKey(0)
    total, count = 0.0, 0
    with _packed_weights(experiment):
        for eval_step, batch in enumerate(loader):
            if max_images and (eval_step * world + rank) * batch_size >= max_images:
                break
            with torch.no_grad():
                bpd, _ = experiment.loss_fn(experiment.orig_params, batch, eval_step, rng=rng, is_train=False)
            total += float(bpd)
            count += 1
            if count % 100 == 0 and rank == 0:
                print(f'eval_step {count} cum_avg_bpd {total / count} ')
    return total, count


def eval_bpd_sparse_sampling(experiment, config, max_images=0):
    total, count = _sparse_partial(experiment, config, experiment.rank, experiment.world, max_images)
    mean, n = _reduce_mean(total, count, experiment.device)
    if experiment.rank == 0:
        print('Num eval steps:', n)
    return mean


# ------------------------------------------------------------------ exact likelihood (probability-flow ODE)
GAMMA_TN = -13.3                  # `gt` hard-coded in get_ode_likelihood_fn / _get_bpd_offset (notebook_utils.py:321,452)
TN_LOG_Z = float(np.log(0.9974613))          # mass of N(0,1) on [-3, 3] as the reference rounds it (:331)


class Hutchinson:
    """notebook_utils.Hutchinson (:232-260): one probe per function evaluation, or one fixed probe"""

    def __init__(self, hutchinson_type, shape, rng, device, deterministic=False):
        if hutchinson_type not in ('Rademacher', 'Gaussian'):
            raise ValueError(f'hutchinson_type {hutchinson_type}')
        self.kind, self.shape, self.rng, self.device = hutchinson_type, shape, rng, device
        self.deterministic = deterministic
        if deterministic:
            self.det_noise = self._draw(rng)

    def _draw(self, key):
        from . import ops
        if self.kind == 'Gaussian':
            return key.normal(self.shape, self.device)
        return ops.noise(self.shape, key.v, 0, self.device, 'rademacher')

    def noise(self):
        if self.deterministic:
            return self.det_noise
        self.rng, key = self.rng.split()
        return self._draw(key)


def get_ode_likelihood_fn(experiment, hutchinson_type='Rademacher', rtol=1e-5, atol=1e-5, method='RK45',
                          dequantization='uniform', high_precision=False):
    """get_ode_likelihood_fn (ldm/notebook_utils.py:263-373).  Returns likelihood_fn(rng, data_u8 [B,32,32,3],
    deterministic_noise=False, u=None, probes=None) -> (log_p [B] float64, log_q_eps [B] | None, aux_loss [B], info).
    `u` (dequantisation noise: U[0,1) for 'uniform', standard normal on [-3, 3] for 'tn') and `probes` (a callable
    returning the Hutchinson probe of each function evaluation) override the Philox draws and `t_grid` replaces the
    adaptive controller by fixed Dormand-Prince steps: parity tests pass them."""
    from . import ops
    from .model import ode_function
    from .ode import solve_fixed, solve_rk45
    if method != 'RK45':
        raise NotImplementedError("the reference only ever passes method='RK45' (ldm/notebook_utils.py:264)")
    if dequantization not in ('uniform', 'tn'):
        raise AssertionError(dequantization)
    model, params, dev = experiment.model, experiment.orig_params, experiment.device
    packer = experiment.state.param_packer("ema") if params is experiment.state.ema_params else None
    graphs = {}          # captured function evaluations, re-used from batch to batch (model.ode_function)

    def likelihood_fn(rng, data, deterministic_noise=False, u=None, probes=None, t_grid=None):
        rng, init_noise_rng = rng.split()
        x = data.reshape(-1, 3072).contiguous()
        B = x.shape[0]
        if dequantization == 'uniform':
            if u is None:
                u = ops.noise((B, 3072), init_noise_rng.v, 0, dev, 'uniform')
            y, requant = ops.dequantize(x, u, True)
            log_q_eps = None
        else:
            if u is None:
                u = ops.noise((B, 3072), init_noise_rng.v, 0, dev, 'truncated_normal', -3.0, 3.0)
            log_q_eps = ops.normal_logp(u).double() - 3072 * TN_LOG_Z
            y, requant = ops.dequantize(x, u, False, float(np.exp(np.float32(0.5 * GAMMA_TN))))
        if packer is not None:
            packer.refresh()
        try:
            ctx = model.ode_context(params, requant)
            rng, hutchinson_rng = rng.split()
            hutch = Hutchinson(hutchinson_type, (B, 3072), hutchinson_rng, dev, deterministic=deterministic_noise)
            draw = probes if probes is not None else hutch.noise
            n_x = B * 3072

            f = ode_function(model, params, ctx, B, dev, True, cache=graphs, high_precision=high_precision)   # a replayed HIP graph

            def ode_func(t, y32, out):
                f(t, y32[:n_x].view(B, 3072), draw(), out[:n_x].view(B, 3072), out[n_x:])

            y0 = torch.cat([y.reshape(-1).double(), torch.zeros(B, device=dev, dtype=torch.float64)])
            sol = (solve_rk45(ode_func, y0, (0.0, 1.0), rtol=rtol, atol=atol) if t_grid is None
                   else solve_fixed(ode_func, y0, t_grid))
        finally:
            if packer is not None:
                packer.invalidate()
        z = sol.y[:n_x].float().view(B, 3072)
        log_p = ops.normal_logp(z).double() + sol.y[n_x:].float().double()
        return log_p, log_q_eps, ctx["kl"], dict(nfev=sol.nfev, steps=sol.steps, rejected=sol.rejected, z=z)

    return likelihood_fn


def get_sample_fn(experiment, hutchinson_type='Rademacher', rtol=1e-5, atol=1e-5, method='RK45', high_precision=False):
    """get_sample_fn (ldm/notebook_utils.py:376-443): samples by integrating the probability-flow ODE from the prior
    at t = 1 down to t = 0, conditioned on the hard top-k embedding of random normal logits.  Returns
    sample_fn(rng, deterministic_noise=False, sample_size=32, t_grid=None) -> (z [sample_size, 32, 32, 3] fp32, nfev).
    The reference evaluates the Hutchinson divergence at every step and throws it away; only the drift is computed
    here (so hutchinson_type / deterministic_noise have no effect on the result, as in the reference)."""
    from . import ops
    from .model import ode_function
    from .ode import solve_fixed, solve_rk45
    if method != 'RK45':
        raise NotImplementedError("the reference only ever passes method='RK45'")
    model, params, dev = experiment.model, experiment.orig_params, experiment.device
    packer = experiment.state.param_packer("ema") if params is experiment.state.ema_params else None

    def sample_fn(rng, deterministic_noise=False, sample_size=32, t_grid=None):
        rng, logits_rng = rng.split()
        emb, _ = ops.topk_hard(logits_rng.normal((sample_size, 50), dev), 15)
        rng, _hutchinson_rng = rng.split()
        rng, prior_rng = rng.split()
        prior = prior_rng.normal((sample_size, 3072), dev)
        if packer is not None:
            packer.refresh()
        try:
            ctx = model.ode_context_from_embedding(params, emb)

            f = ode_function(model, params, ctx, sample_size, dev, False, high_precision=high_precision)

            def ode_func(t, y32, out):
                f(t, y32.view(sample_size, 3072), None, out.view(sample_size, 3072))

            y0 = prior.reshape(-1).double()
            sol = (solve_rk45(ode_func, y0, (1.0, 0.0), rtol=rtol, atol=atol) if t_grid is None
                   else solve_fixed(ode_func, y0, t_grid))
        finally:
            if packer is not None:
                packer.invalidate()
        return sol.y.float().view(sample_size, 32, 32, 3), sol.nfev

    return sample_fn


def get_logits(experiment, num_batches=30):
    """notebook_utils.get_logits (:534-545): encoder logits and the images they belong to, over eval batches"""
    logits, images = [], []
    for _ in range(num_batches):
        batch = experiment.eval_iter.next()
        logits.append(experiment.model.apply_encoder(experiment.orig_params, batch['images']))
        images.append(batch['images'])
    return torch.cat(logits), torch.cat(images)


def logits_to_embeddings(logits):
    """notebook_utils.logits_to_embeddings (:548-551): hard top-15 k-hot"""
    from . import ops
    return ops.topk_hard(logits, 15)[0]


def _get_bpd_offset(dequantization, num_is):
    """notebook_utils._get_bpd_offset (:446-458)"""
    if dequantization == 'uniform':
        return float(np.log2(128))
    if dequantization == 'tn':
        log_sigma = 0.5 * (GAMMA_TN - float(np.logaddexp(GAMMA_TN, 0.0)))
        extra = 0.5 * (1 + float(np.log(2 * np.pi))) - 0.01522 if num_is == 1 else 0.0
        return -(extra + log_sigma) / float(np.log(2))
    raise AssertionError(dequantization)


def _logsumexp0(a):
    m = a.max(axis=0)
    return m + np.log(np.exp(a - m).sum(axis=0))


def _eval_bpd_ode(experiment, config, rng, deterministic_noise, hutchinson_type, dequantization='tn', num_is=1,
                  rtol=1e-5, atol=1e-5, max_images=0):
    """notebook_utils._eval_bpd_ode (:480-531): per batch, num_is likelihood draws, importance-weighted bound,
    running mean over batches.  Under torchrun whole global batches (config.training.batch_size_eval images, as in the
    reference) are dealt round-robin to the ranks: every batch is integrated by one rank exactly as a single device
    would (same step-size controller input), so the result does not depend on the world size."""
    batch_size = config.training.batch_size_eval
    loader = dataset.create_one_time_eval_dataset(config, batch_size, experiment.device, experiment.rank,
                                                  experiment.world)
    likelihood_function = get_ode_likelihood_fn(experiment, rtol=rtol, atol=atol, hutchinson_type=hutchinson_type,
                                                dequantization=dequantization)
    bpd_offset = _get_bpd_offset(dequantization, num_is)
    total, count = 0.0, 0
    for eval_step, batch in enumerate(loader):
        if max_images and (eval_step * experiment.world + experiment.rank) * batch_size >= max_images:
            break
        log_ps, log_qs, aux_loss = [], [], None
        for _ in range(num_is):
            rng, likelihood_rng = rng.split()
            log_p, log_q_eps, aux, _ = likelihood_function(likelihood_rng, batch['images'],
                                                           deterministic_noise=deterministic_noise)
            log_ps.append(log_p.cpu().numpy())
            log_qs.append(None if log_q_eps is None else log_q_eps.cpu().numpy())
            aux_loss = aux.double().cpu().numpy()
        log_ps = np.asarray(log_ps)
        if num_is == 1:
            iws = log_ps[0]
        else:
            if log_qs[0] is None:
                raise TypeError("importance weighting needs dequantization='tn' (the reference subtracts None here)")
            iws = _logsumexp0(log_ps - np.asarray(log_qs)) - np.log(num_is)
        bpd = float(np.mean(-iws + aux_loss) / (32 * 32 * 3 * np.log(2)) + bpd_offset)
        total += bpd
        count += 1
        if experiment.rank == 0:
            print('Eval step:{}\tcum. bpd: {:.3f}'.format(eval_step, total / count))
    mean, n = _reduce_mean(total, count, experiment.device)
    if experiment.rank == 0:
        print('Num eval steps:', n)
    return mean


def eval_bpd_ode(experiment, config, deterministic_noise, hutchinson_type, dequantization='tn', num_is=1, num_iters=1,
                 rtol=1e-5, atol=1e-5, max_images=0):
    """notebook_utils.eval_bpd_ode (:461-477)"""
    bpd_means = []
    rng = PRNGKey(0)
    for i in range(num_iters):
        rng, iter_rng = rng.split()
        mean = _eval_bpd_ode(experiment=experiment, config=config, rng=iter_rng, deterministic_noise=deterministic_noise,
                             hutchinson_type=hutchinson_type, dequantization=dequantization, num_is=num_is, rtol=rtol,
                             atol=atol, max_images=max_images)
        if experiment.rank == 0:
            print(f'[Iter {i}] Test BPD:{mean}')
        bpd_means.append(mean)
    return float(np.mean(bpd_means))
