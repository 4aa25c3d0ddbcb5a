"""Experiment.train_step as ONE unit -- rng fold-in, value_and_grad, lr(step), AdamW x 2
masked, EMA, logged scalars (ldm/experiment.py:335-356, ldm/train_state.py:70-102, ldm/experiment.py:106-182) -- over
several optimiser steps against the float64 oracle, with the eager step and with the HIP-graph replay
(GraphedStep, the build's lax.scan).

Every step is checked three ways:
  * value_and_grad: the oracle (float64 autograd through its own forward, with the step's noise and dropout masks
    re-derived from the step's keys: fold_in(rank), fold_in(step), the 'sample' / 'dropout' splits) is evaluated at
    the parameters the device holds BEFORE the step; the logged train_bpd agrees to 1e-4 (bar: +-0.005 absolute) and
    every parameter gradient to the per-leaf bar of test_gpu_model.run_case (2e-3 of the leaf's scale);
  * optimizer, teacher-forced: the oracle's AdamW + EMA + lr schedule, run in float64 from step 0 on the gradients
    the HIP steps produced, reproduces parameters, EMA and both Adam moments after EVERY step to fp32 rounding
    (1e-5 of each leaf's scale): this pins the composition -- lr(step) with the warm-up index (lr(0) = 0: the first
    step moves the moments only), the bias-correction count, the decay-mask boundary in the flat buffer, grad_scale,
    the EMA order -- on the eager and on the graph-replayed path (stream-ordered lr / bias corrections);
  * optimizer, on the oracle's own gradients: the Adam moments accumulated from the oracle's gradients agree with the
    device's at the gradient bar.
Why the parameters are not compared free-running: Adam divides every element by its own magnitude, so an element whose
gradient is zero to fp32 noise moves by +-lr whichever way the noise points, and the hard top-k latent turns such a
difference into a different discrete code within two steps (measured: encoder gradients 100 % apart at step 2).  Two
correct implementations diverge like that; the three checks above leave no part of the step unpinned.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mulan_np as onp
from oracle import torch_ref as tr

from tests.oracle_dev import run_oracle
from tests.test_gpu_model import block_names, oracle_masks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, E, STEPS, WARMUP = 4, 128, 3, 2
ONE_ELEMENT_BAR = 5e-2


def _experiment(graph, vdm_type):
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.vdm_type = vdm_type
    config.data.dataset = "synthetic"
    config.model.sm_n_layer = 1
    config.model.forward_n_layer = 1
    config.training.batch_size_train = B
    config.training.batch_size_eval = B
    config.training.substeps = 1
    config.training.num_steps_lr_warmup = WARMUP
    config.training.hip_graph = graph
    config.optimizer.ema_rate = 0.9            # (0.9999 would hide an EMA mistake below the tolerance)
    return Experiment_VDM(config), config


def _leaf(tree, path):
    for k in path:
        tree = tree[k]
    return tree


def _flax_state(exp):
    """(params, ema, mu, nu) of the product as float64 numpy trees in the reference layout"""
    from mulan_amd import model as M
    st = exp.state

    def tree_of(flat):
        views = {}
        for path, off, shape in st.layout:
            d = views
            for k in path[:-1]:
                d = d.setdefault(k, {})
            d[path[-1]] = flat[off:off + int(np.prod(shape))].view(shape)
        return M.tree_map(lambda t: t.detach().double().cpu().numpy(), M.to_flax_layout(views))
    return tuple(tree_of(f) for f in (st.flat, st.ema, st.mu, st.nu)), tree_of(st.grad)


def _oracle_step_inputs(exp, step, images, vdm_type):
    """noise and dropout masks of train step `step`, re-derived from the experiment's keys the way
    Experiment.train_step / loss_fn / VDM.apply derive them (ldm/experiment.py:336-337, experiment_vdm.py:48-53)"""
    rng = exp._train_rng.fold_in(exp.rank).fold_in(step)
    keys = exp.step_keys(rng, True)
    _, sample_rng = rng.split()
    noise = exp.model._noise({"sample": sample_rng}, None, B, exp.device, True)
    enc_masks = oracle_masks(block_names(1, False), keys["enc"], B, E, 0.9)
    score_masks = oracle_masks(block_names(1, True), keys["score"], B, E, 0.9)
    f64 = lambda t: t.double().cpu()
    return dict(t0=float(noise["t0"]), raw=f64(noise["gamma_raw"]), e0=f64(noise["eps_0"]).view(B, 32, 32, 3),
                e=f64(noise["eps"]).view(B, 32, 32, 3), enc_masks=enc_masks, score_masks=score_masks)


@pytest.mark.parametrize("vdm_type", ["mulan_epsilon", "mulan_velocity"])
def test_train_step_trajectory_matches_oracle(vdm_type):
    from mulan_amd import model as M
    ocfg = dict(vdm_type=vdm_type, n_embd=E, n_layer=1, forward_n_layer=1, latent_k=15, unet_type="vdm",
                velocity_from_epsilon=False, with_attention=False)
    init = tr.init_params(ocfg, seed=13, dtype=torch.float64)           # non-zero everywhere: every gradient is live
    paths = [p for p, _ in tr.tree_leaves(init)]
    decay_mask = {p: float(p[-1] != "bias" and tuple(p[-2:]) not in (("layer_norm", "scale"), ("final_layer_norm", "scale")))
                  for p in paths}                                        # ldm/experiment.py:139-146
    g = torch.Generator().manual_seed(7)
    batches = [torch.randint(0, 256, (B, 32, 32, 3), generator=g, dtype=torch.uint8) for _ in range(STEPS)]
    keep = float(np.float32(0.9))
    lr0, ema_rate = 2e-4, 0.9

    exp0, _ = _experiment(False, vdm_type)
    step_inputs = [_oracle_step_inputs(exp0, k, batches[k], vdm_type) for k in range(STEPS)]     # depend on keys only
    del exp0

    def oracle_value_and_grad(k, flax_params):
        """(bpd, {path: gradient}) of train step k at the given parameters (a float64 numpy tree, reference layout)"""
        tree = tr.tree_map(lambda t: t, init)
        leaves = {}
        for p in paths:
            t = torch.tensor(_leaf(flax_params, p), dtype=torch.float64, requires_grad=True)
            d = tree
            for key in p[:-1]:
                d = d[key]
            d[p[-1]] = t
            leaves[p] = t
        si = step_inputs[k]
        out = run_oracle(lambda P, *a, **kw: tr.mulan_forward(P, ocfg, *a, keep=keep, **kw), tree, batches[k], si["t0"],
                         si["raw"], si["e0"], si["e"], enc_masks=si["enc_masks"], score_masks=si["score_masks"],
                         backward="bpd")
        return float(out["bpd"].detach()), {p: (leaves[p].grad.numpy() if leaves[p].grad is not None
                                                else np.zeros(tuple(leaves[p].shape))) for p in paths}

    def rel_leaf(a, b, floor=1e-30):
        return float(np.abs(a - b).max() / (np.abs(b).max() + floor))

    oracle_cache = {}
    for graph in (False, True):
        exp, config = _experiment(graph, vdm_type)
        M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), init), exp.state.params)
        with torch.no_grad():
            exp.state.ema.copy_(exp.state.flat)
        (p0, _, _, _), _ = _flax_state(exp)
        zeros = lambda: {p: np.zeros_like(_leaf(p0, p)) for p in paths}
        # teacher-forced oracle state (float64, fed the device's gradients) and the moments of the oracle's own gradients
        tf_p = {p: _leaf(p0, p).copy() for p in paths}
        tf_ema = {p: v.copy() for p, v in tf_p.items()}
        tf_m, tf_v, own_m, own_v = zeros(), zeros(), zeros(), zeros()
        worst = dict(tf=0.0, grad=0.0, mom=0.0, bpd=0.0)
        for k in range(STEPS):
            (before, _, _, _), _ = _flax_state(exp)
            # (the replayed step is bit-identical to the eager one: the oracle's float64 pass of a step is reused when
            # the device holds the very same parameters)
            hit = oracle_cache.get(k)
            if hit is not None and all(np.array_equal(_leaf(hit[0], p), _leaf(before, p)) for p in paths):
                want_bpd, want_g = hit[1], hit[2]
            else:
                want_bpd, want_g = oracle_value_and_grad(k, before)
                oracle_cache[k] = (before, want_bpd, want_g)
            batch = {"images": batches[k].cuda(), "labels": torch.zeros(B, dtype=torch.int32).cuda(),
                     "conditioning": torch.zeros(B, dtype=torch.uint8).cuda()}
            assert exp.state.step == k
            _, metrics = exp.train_step(exp._train_rng, exp.state, batch)
            torch.cuda.synchronize()
            assert exp.state.step == k + 1
            (got_p, got_ema, got_m, got_v), got_g = _flax_state(exp)
            lr = onp.lr_schedule(k, lr0, WARMUP)
            bpd = float(metrics["scalars"]["train_bpd"])
            worst["bpd"] = max(worst["bpd"], abs(bpd - want_bpd))
            assert abs(bpd - want_bpd) < 1e-3, (graph, k, bpd, want_bpd)          # (the north-star bar is 0.005)
            bad = []
            for p in paths:
                gk = _leaf(got_g, p)
                tf_p[p], tf_m[p], tf_v[p], tf_ema[p] = onp.adamw_ema_step(
                    tf_p[p], gk, tf_m[p], tf_v[p], tf_ema[p], lr, k + 1, decay_mask[p], ema_rate=ema_rate)
                for name, got, want in (("params", got_p, tf_p), ("ema", got_ema, tf_ema), ("mu", got_m, tf_m),
                                        ("nu", got_v, tf_v)):
                    err = rel_leaf(_leaf(got, p), want[p])
                    worst["tf"] = max(worst["tf"], err)
                    assert err < 1e-5, (graph, k, name, "/".join(p), err)
                _, own_m[p], own_v[p], _ = onp.adamw_ema_step(tf_p[p], want_g[p], own_m[p], own_v[p], tf_ema[p], lr, k + 1,
                                                              decay_mask[p], ema_rate=ema_rate)
                if np.abs(want_g[p]).max() < 1e-12:
                    # a gradient that vanishes identically (the bias of the attention keys: softmax ignores a constant
                    # added to a row of scores; float64 leaves 1e-17 of rounding noise, fp32 1e-8): noise on both sides
                    assert np.abs(gk).max() < 1e-6, (graph, k, "/".join(p), float(np.abs(gk).max()))
                    continue
                eg = rel_leaf(gk, want_g[p], 1e-6)
                em = rel_leaf(_leaf(got_m, p), own_m[p], 1e-7)
                ev = rel_leaf(_leaf(got_v, p), own_v[p], 1e-13)
                worst["grad"], worst["mom"] = max(worst["grad"], eg), max(worst["mom"], em)
                # (a one-element leaf -- the bias of the encoder's single-channel conv_out -- is the sum of B * 1024
                # cancelling terms: its own magnitude says nothing about the size of its error.  Round 4 moved that sum to
                # float64 accumulation (mulan_colsum, narrow tensors) and the measured error did not move (2.3e-2 before,
                # 3.1e-2 after, another box): it is the fp32 rounding of the 4096 TERMS -- each the end of a ~40-kernel fp32
                # chain, 1e-6 relative to the float64 oracle -- amplified by the cancellation sum |t| / |sum t| ~ 1e4, not
                # the summation order; no bar below that amplification can hold for a correct fp32 implementation)
                bar = 2e-3 if want_g[p].size > 16 else ONE_ELEMENT_BAR
                if p[0] == "gamma":      # the schedule network's gradients sum sigma(gamma) / exp terms over a 18-unit range
                    bar = 1e-2           # of gamma in fp32 (measured up to 5e-3 once the parameters have moved)
                if eg >= bar or em >= bar or ev >= 2 * bar:
                    bad.append((round(max(eg, em, ev / 2) / bar, 2), "/".join(p), f"grad {eg:.2e} mu {em:.2e} nu {ev:.2e}",
                                f"scale {np.abs(want_g[p]).max():.2e}"))
            assert not bad, (graph, k, sorted(bad, reverse=True)[:6])
            if lr == 0:                                      # lr(0) = 0: the first step moves the moments only
                for p in paths:
                    assert np.array_equal(_leaf(got_p, p), _leaf(before, p)), (graph, k, "/".join(p))
            else:
                assert any(not np.array_equal(_leaf(got_p, p), _leaf(before, p)) for p in paths)
        assert (exp._graphed is not None) == graph
        print(f"trajectory {vdm_type} graph={graph}: teacher-forced {worst['tf']:.2e}, gradient {worst['grad']:.2e}, "
              f"moments {worst['mom']:.2e}, |bpd - oracle| {worst['bpd']:.2e}")
        del exp
        torch.cuda.empty_cache()


def test_config1_plain_vdm_ten_train_steps_through_the_flag_surface(tmp_path):
    """BASELINE configs[0] as worded -- `python -m ldm.main --config.vdm_type=vdm --config.model.gamma_type=learnable_nnet
    --config.training.batch_size_train=2 --config.training.substeps=1`, 10 train steps -- on the HIP path
    (model_vdm.VDM: ldm/model_vdm.py:110-180; Experiment.train_step: ldm/experiment.py:335-356):
      * the entry point itself runs the 10 optimiser steps at the config's shipped depth (32 + 2 + 33 ResnetBlocks) and
        leaves a checkpoint at step 10;
      * the same flags, parsed by the same flag surface, at one ResnetBlock per stage: every one of the 10 steps checked
        like test_train_step_trajectory_matches_oracle -- train_bpd and every parameter gradient against
        oracle.torch_ref.plain_vdm_forward (float64 autograd, the step's noise and dropout masks re-derived from its keys)
        at the parameters the device holds before the step, and the oracle's AdamW + EMA + lr schedule, teacher-forced on
        the device's gradients, reproducing parameters / EMA / both moments after every step to fp32 rounding."""
    import ldm.main
    from mulan_amd import checkpoint as ck
    from mulan_amd import model as M
    from mulan_amd.experiment import Experiment_VDM
    cfgp = os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py")
    flags = ["--config=" + cfgp, "--config.vdm_type=vdm", "--config.model.gamma_type=learnable_nnet",
             "--config.training.batch_size_train=2", "--config.training.substeps=1", "--config.data.dataset=synthetic"]
    NB, NSTEPS, WU = 2, 10, 4
    # ---- (1) the entry point, shipped depth
    ldm.main.main(flags + ["--config.training.batch_size_eval=2", "--config.training.num_steps_train=10",
                           "--config.training.num_steps_eval=1", "--config.training.steps_per_logging=1",
                           "--config.training.steps_per_eval=10", "--config.training.steps_per_save=10",
                           "--config.training.sample_timesteps=2", "--workdir=" + str(tmp_path / "run")])
    ckdirs = [os.path.join(dp, d) for dp, dn, _ in os.walk(tmp_path / "run") for d in dn if d == "checkpoints"]
    assert len(ckdirs) == 1
    sd = ck.restore_dict(ckdirs[0])
    assert sd["step"] == 10 and set(sd["params"]) == {"score_model", "gamma"} and set(sd["params"]["gamma"]) == {"l1", "l2", "l3"}
    torch.cuda.empty_cache()
    # ---- (2) the same flags at depth 1, step by step against the oracle
    ldm.main.FLAGS.parse(flags + ["--config.model.sm_n_layer=1", "--config.training.batch_size_eval=2",
                                  f"--config.training.num_steps_lr_warmup={WU}", "--config.optimizer.ema_rate=0.9",
                                  "--workdir=" + str(tmp_path / "unused")])
    config = ldm.main.FLAGS.config
    assert config.vdm_type == "vdm" and config.model.gamma_type == "learnable_nnet" and config.training.batch_size_train == NB
    exp = Experiment_VDM(config)
    assert not hasattr(exp.model, "parameterization")            # model_vdm.VDM, not a MuLAN model
    ocfg = dict(vdm_type="mulan_velocity", n_embd=E, n_layer=1, forward_n_layer=1, latent_k=15, unet_type="vdm")
    full = tr.init_params(ocfg, seed=23, dtype=torch.float64)
    gg = torch.Generator().manual_seed(8)
    init = {"score_model": full["score_model"],
            "gamma": {"l1": {"kernel": torch.tensor([[-16.0]], dtype=torch.float64), "bias": torch.tensor([-12.0], dtype=torch.float64)},
                      "l2": {"kernel": torch.randn(1, 1024, generator=gg, dtype=torch.float64) * 3,
                             "bias": torch.randn(1024, generator=gg, dtype=torch.float64)},
                      "l3": {"kernel": torch.randn(1024, 1, generator=gg, dtype=torch.float64) * 2}}}
    init["score_model"]["dense0"]["kernel"] = init["score_model"]["dense0"]["kernel"][:129].clone()   # conditioning is [B, 1]
    paths = [p for p, _ in tr.tree_leaves(init)]
    decay_mask = {p: float(p[-1] != "bias") for p in paths}
    M.from_flax_layout(M.tree_map(lambda t: t.detach().float(), init), exp.state.params)
    with torch.no_grad():
        exp.state.ema.copy_(exp.state.flat)
    g = torch.Generator().manual_seed(17)
    batches = [torch.randint(0, 256, (NB, 32, 32, 3), generator=g, dtype=torch.uint8) for _ in range(NSTEPS)]
    keep = float(np.float32(1.0 - float(config.model.sm_pdrop)))
    lr0, ema_rate = float(config.optimizer.learning_rate), 0.9
    rel_leaf = lambda a, b, floor=1e-30: float(np.abs(a - b).max() / (np.abs(b).max() + floor))
    (p0, _, _, _), _ = _flax_state(exp)
    tf_p = {p: _leaf(p0, p).copy() for p in paths}
    tf_ema = {p: v.copy() for p, v in tf_p.items()}
    tf_m = {p: np.zeros_like(v) for p, v in tf_p.items()}
    tf_v = {p: np.zeros_like(v) for p, v in tf_p.items()}
    worst = dict(tf=0.0, grad=0.0, bpd=0.0)
    for k in range(NSTEPS):
        (before, _, _, _), _ = _flax_state(exp)
        # the step's keys as Experiment.train_step / loss_fn / model_vdm.VDM.apply derive them: fold_in(rank), fold_in(step),
        # then the 'sample' and the 'dropout' split; the plain VDM hands the dropout key to the score U-Net unsplit
        rng = exp._train_rng.fold_in(exp.rank).fold_in(k)
        rng, sample_rng = rng.split()
        rng, dropout_rng = rng.split()
        noise = exp.model._noise({"sample": sample_rng}, None, NB, exp.device, False)
        masks = oracle_masks(block_names(1, True), dropout_rng, NB, E, keep)
        tree = tr.tree_map(lambda t: t, init)
        leaves = {}
        for p in paths:
            t = torch.tensor(_leaf(before, p), dtype=torch.float64, requires_grad=True)
            d = tree
            for key in p[:-1]:
                d = d[key]
            d[p[-1]] = t
            leaves[p] = t
        f64 = lambda t: t.double().cpu()
        ref = tr.plain_vdm_forward(tree, ocfg, batches[k], float(noise["t0"]), f64(noise["eps_0"]).view(NB, 32, 32, 3),
                                   f64(noise["eps"]).view(NB, 32, 32, 3), score_masks=masks, keep=keep)
        ref["bpd"].backward()
        batch = {"images": batches[k].cuda(), "labels": torch.zeros(NB, dtype=torch.int32).cuda(),
                 "conditioning": torch.zeros(NB, dtype=torch.uint8).cuda()}
        _, metrics = exp.train_step(exp._train_rng, exp.state, batch)
        torch.cuda.synchronize()
        assert exp.state.step == k + 1 and exp._graphed is None
        (got_p, got_ema, got_m, got_v), got_g = _flax_state(exp)
        bpd = float(metrics["scalars"]["train_bpd"])
        worst["bpd"] = max(worst["bpd"], abs(bpd - float(ref["bpd"].detach())))
        assert abs(bpd - float(ref["bpd"].detach())) < 1e-3, (k, bpd, float(ref["bpd"].detach()))
        lr = onp.lr_schedule(k, lr0, WU)
        bad = []
        for p in paths:
            gk = _leaf(got_g, p)
            want = leaves[p].grad.numpy() if leaves[p].grad is not None else np.zeros(tuple(leaves[p].shape))
            tf_p[p], tf_m[p], tf_v[p], tf_ema[p] = onp.adamw_ema_step(tf_p[p], gk, tf_m[p], tf_v[p], tf_ema[p], lr, k + 1,
                                                                      decay_mask[p], ema_rate=ema_rate)
            for name, got, ref_t in (("params", got_p, tf_p), ("ema", got_ema, tf_ema), ("mu", got_m, tf_m), ("nu", got_v, tf_v)):
                err = rel_leaf(_leaf(got, p), ref_t[p])
                worst["tf"] = max(worst["tf"], err)
                assert err < 1e-5, (k, name, "/".join(p), err)
            if np.abs(want).max() < 1e-12:
                assert np.abs(gk).max() < 1e-6, (k, "/".join(p))
                continue
            eg = rel_leaf(gk, want, 1e-6)
            worst["grad"] = max(worst["grad"], eg)
            bar = 5e-3 if p[0] == "gamma" else (2e-3 if want.size > 16 else ONE_ELEMENT_BAR)   # (bars of test_plain_vdm_matches_oracle)
            if eg >= bar:
                bad.append((round(eg / bar, 2), "/".join(p), f"{eg:.2e}", f"scale {np.abs(want).max():.2e}"))
        assert not bad, (k, sorted(bad, reverse=True)[:6])
    print(f"config #1 (model_vdm.VDM, learnable_nnet, batch 2, 10 steps): teacher-forced {worst['tf']:.2e}, gradient "
          f"{worst['grad']:.2e}, |bpd - oracle| {worst['bpd']:.2e}")
