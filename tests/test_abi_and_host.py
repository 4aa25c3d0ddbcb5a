"""CPU tests: the C-ABI library loads and exports every symbol include/mulan_hip.h declares (no compute), and the
host logic around the kernels (config/flags, checkpoints, TrainState layout, rng, data, schedules)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from mulan_amd import build
    return build.build_library()


def _header_decls():
    src = open(os.path.join(ROOT, "include", "mulan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(mulan_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(2)] = n
    return decls


def test_header_and_binding_table_agree():
    from mulan_amd import lib
    decls = _header_decls()
    assert set(decls) == set(lib.SIGNATURES), set(decls) ^ set(lib.SIGNATURES)
    for name, n in decls.items():
        assert len(lib.SIGNATURES[name]) == n, (name, n, len(lib.SIGNATURES[name]))


def test_shared_object_exports_every_declared_symbol(built_lib):
    h = ctypes.CDLL(built_lib)
    for name in _header_decls():
        assert getattr(h, name) is not None
    h.mulan_version.restype = ctypes.c_char_p
    assert b"gfx950" in h.mulan_version()


def test_header_is_plain_c_and_the_cpp_driver_compiles_against_it(built_lib, tmp_path):
    """the boundary needs neither torch nor HIP headers: include/mulan_hip.h passes a C99 compiler on its own, and the C++
    caller of tests/abi_driver.cpp (run on the GPU by tests/test_gpu_kernels.py) compiles and links against the library.
    The library's own sources include the same header (csrc/common.h), so a definition that drifts from its declaration
    does not build."""
    import shutil
    import subprocess
    hdr = os.path.join(ROOT, "include", "mulan_hip.h")
    r = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wpedantic", "-Werror", hdr],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "mulan_hip.h" in open(os.path.join(ROOT, "mulan_amd", "csrc", "common.h")).read()
    from mulan_amd import build
    exe = build.build_abi_driver(str(tmp_path / "abi_driver"))
    assert os.path.exists(exe)
    ldd = shutil.which("ldd")
    if ldd:
        out = subprocess.run([ldd, exe], capture_output=True, text=True).stdout
        assert "libmulan_hip.so" in out and "libtorch" not in out and "libpython" not in out, out


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mulan_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.MulanHipError):
        lib.load()


def test_product_never_imports_the_oracle():
    for pkg in ("mulan_amd", "ldm"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith(".py"):
                    text = open(os.path.join(dirpath, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)


# ------------------------------------------------------------------------------ config / flags
def test_shipped_configs_and_overrides():
    from mulan_amd.config import Flags, load_config_file
    from mulan_amd.model import VDMConfig
    c = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    i = load_config_file(os.path.join(ROOT, "ldm", "configs", "imagenet32.py"))
    assert (c.vdm_type, c.model.sm_n_embd, c.training.batch_size_train) == ("mulan_velocity", 128, 128)
    assert (i.vdm_type, i.model.sm_n_embd, i.training.batch_size_train) == ("mulan_epsilon", 256, 512)
    assert c.optimizer.args.b2 == 0.99 and c.optimizer.ema_rate == 0.9999 and c.model.gamma_min == -13.3
    VDMConfig(**c.model.to_dict())
    VDMConfig(**i.model.to_dict())
    with pytest.raises(TypeError):
        VDMConfig(**dict(c.model.to_dict(), no_such_key=1))
    F = Flags()
    F.DEFINE_config_file("config")
    F.DEFINE_string("workdir", None)
    F.DEFINE_integer("n_timesteps", 128)
    F.DEFINE_bool("deterministic_noise", False)
    F.mark_flags_as_required(["config", "workdir"])
    F.parse(["--config=" + os.path.join(ROOT, "ldm", "configs", "imagenet32.py"), "--workdir", "/tmp/w",
             "--config.vdm_type=mulan_velocity", "--config.model.velocity_from_epsilon=True",
             "--config.training.batch_size_train=64", "--n_timesteps=1000", "--deterministic_noise"])
    assert F.config.vdm_type == "mulan_velocity" and F.config.model.velocity_from_epsilon is True
    assert F.config.training.batch_size_train == 64 and F.n_timesteps == 1000 and F.deterministic_noise is True
    with pytest.raises(SystemExit):
        Flags().parse(["--bogus=1"])


def test_reference_style_config_file_loads(tmp_path):
    from mulan_amd.config import load_config_file
    p = tmp_path / "cfg.py"
    p.write_text("import ml_collections\n\ndef get_config():\n    c = ml_collections.ConfigDict()\n"
                 "    c.a = ml_collections.ConfigDict(initial_dictionary=dict(b=3))\n    return c\n")
    assert load_config_file(str(p)).a.b == 3


def test_workdir_name_follows_overrides():
    from ldm.utils import get_workdir
    w = get_workdir(["prog", "--config=ldm/configs/cifar10-conditioned.py", "--workdir=/x",
                     "--config.model.sm_n_layer=8", "--config.seed=1"])
    assert w.startswith("cifar10-conditioned/") and "sm_n_layer=8" in w


# ------------------------------------------------------------------------------ rng / data
def test_rng_is_deterministic_and_splits_differ():
    from mulan_amd.rng import PRNGKey
    k = PRNGKey(1)
    a, b = k.split()
    assert a.v != b.v and PRNGKey(1).split()[0].v == a.v
    assert k.fold_in(3).v != k.fold_in(4).v
    assert 0 <= a.uniform() < 1


def test_synthetic_stream_shapes_and_rank_shards():
    from mulan_amd.config import ConfigDict
    from mulan_amd import data
    cfg = ConfigDict(dict(data=dict(dataset="synthetic"), training=dict(batch_size_train=8, batch_size_eval=4, substeps=3)))
    tr0, ev0 = data.create_dataset(cfg, "cpu", seed=0, rank=0, world=2)
    tr1, _ = data.create_dataset(cfg, "cpu", seed=0, rank=1, world=2)
    b0, b1 = next(tr0), next(tr1)
    assert b0["images"].shape == (3, 4, 32, 32, 3) and b0["images"].dtype == torch.uint8
    assert not torch.equal(b0["images"], b1["images"])
    assert ev0.next()["images"].shape == (2, 32, 32, 3)
    with pytest.raises(ValueError):
        data.BatchStream("synthetic", 7, train=True, device="cpu", world=2)


def test_npz_one_pass_eval_is_sharded_in_order(tmp_path):
    from mulan_amd.config import ConfigDict
    from mulan_amd import data
    imgs = np.arange(10, dtype=np.uint8)[:, None, None, None] * np.ones((1, 32, 32, 3), dtype=np.uint8)
    np.savez(tmp_path / "d.npz", images=imgs)
    cfg = ConfigDict(dict(data=dict(dataset=f"npz:{tmp_path / 'd.npz'}")))
    seen = []
    for rank in range(2):
        it = data.create_one_time_eval_dataset(cfg, 1, "cpu", rank=rank, world=2)
        seen.append([int(b["images"][0, 0, 0, 0]) for b in it])
    assert seen == [[0, 2, 4, 6, 8], [1, 3, 5, 7, 9]]
    # global batches of 4: whole batches go to the ranks round-robin, the remainder (8, 9) is dropped like the
    # reference's drop_remainder batching, so world size does not change which images are evaluated
    for world in (1, 2, 3):
        got = []
        for rank in range(world):
            it = data.create_one_time_eval_dataset(cfg, 4, "cpu", rank=rank, world=world)
            assert len(it) == len(list(data.create_one_time_eval_dataset(cfg, 4, "cpu", rank=rank, world=world)))
            for b in it:
                assert b["images"].shape == (4, 32, 32, 3)
                got.append([int(v) for v in b["images"][:, 0, 0, 0]])
        assert sorted(got) == [[0, 1, 2, 3], [4, 5, 6, 7]]


def test_train_stream_ranks_partition_every_epoch(tmp_path):
    """all ranks walk ONE permutation per epoch (rank-independent generator): the union over ranks of one epoch is the
    data set, without cross-rank duplicates; the next epoch is shuffled differently"""
    from mulan_amd import data
    N, world, bs = 1000, 4, 40
    imgs = np.zeros((N, 32, 32, 3), dtype=np.uint8)
    imgs[:, 0, 0, 0] = np.arange(N) % 256
    imgs[:, 0, 0, 1] = np.arange(N) // 256
    np.savez(tmp_path / "d.npz", images=imgs)
    streams = [data.BatchStream(f"npz:{tmp_path / 'd.npz'}", bs, train=True, device="cpu", seed=3, rank=r, world=world)
               for r in range(world)]
    ident = lambda b: (b["images"][:, 0, 0, 0].long() + 256 * b["images"][:, 0, 0, 1].long()).tolist()
    epochs = []
    for _ in range(2):
        seen = []
        for _ in range(N // bs):
            for st in streams:
                seen += ident(next(st))
        assert sorted(seen) == list(range(N))
        epochs.append(seen)
    assert epochs[0] != epochs[1]


# ------------------------------------------------------------------------------ train state / checkpoints
def _tiny_tree():
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    conv_in = r(3, 3, 16, 8)
    conv_in[:, :, 15, :] = 0          # the 16th input channel is alignment padding and stays zero
    return {"score_model": {"conv_in": {"kernel": conv_in, "bias": r(8)}, "GroupNorm_0": {"scale": r(8), "bias": r(8)}},
            "encoder_model": {"dense": {"kernel": r(5, 3), "bias": r(3)}},
            "gamma": {"dense_1": {"kernel": r(7, 2), "bias": r(2)}}}


def test_train_state_layout_decay_mask_and_views():
    from mulan_amd.train_state import TrainState
    tree = _tiny_tree()
    st = TrainState.create(apply_fn=None, variables={"params": tree}, device="cpu")
    paths = ["/".join(p) for p, _, _ in st.layout]
    n_bias = sum(1 for p in paths if p.endswith("/bias"))
    assert all(p.endswith("/bias") for p in paths[-n_bias:]) and not any(p.endswith("/bias") for p in paths[:-n_bias])
    assert "score_model/GroupNorm_0/scale" in paths[:-n_bias]          # GroupNorm scale IS decayed (experiment.py:139-146)
    first_bias_off = st.layout[len(paths) - n_bias][1]
    assert st.n_decay == first_bias_off and st.numel % 4 == 0
    assert torch.equal(st.params["gamma"]["dense_1"]["kernel"].detach(), tree["gamma"]["dense_1"]["kernel"])
    assert torch.equal(st.ema, st.flat)
    leaf = st.params["encoder_model"]["dense"]["kernel"]
    st.zero_grad()
    (leaf * 2).sum().backward()
    st.collect_grads()                      # ops without a gradient sink are copied into the flat buffer
    off = dict((("/".join(p)), o) for p, o, _ in st.layout)["encoder_model/dense/kernel"]
    assert torch.equal(st.grad[off:off + 15], torch.full((15,), 2.0))
    assert leaf.grad.data_ptr() == leaf._gview.data_ptr()
    st.zero_grad()
    assert leaf.grad is None and float(st.grad.abs().sum()) == 0


def test_checkpoint_roundtrip_pt_and_flax_msgpack(tmp_path):
    from mulan_amd import checkpoint as ck
    from mulan_amd.model import to_flax_layout
    from mulan_amd.train_state import TrainState
    st = TrainState.create(apply_fn=None, variables={"params": _tiny_tree()}, device="cpu")
    st.step = 41
    st.ema.mul_(0.5)
    d = tmp_path / "checkpoints"
    n = ck.save(str(d), st.state_dict())
    assert n == 1 and ck.checkpoint_numbers(str(d)) == [1]
    ck.save(str(d), st.state_dict())
    assert ck.latest_checkpoint(str(d)).endswith("ckpt-2.pt")
    st2 = TrainState.create(apply_fn=None, variables={"params": _tiny_tree()}, device="cpu")
    st2.flat.zero_()
    st2.load_state_dict(ck.restore_dict(os.path.join(str(d), "ckpt-2")))
    assert st2.step == 41 and torch.equal(st2.flat, st.flat) and torch.equal(st2.ema, st.ema)
    # reference layout: conv_in with 15 input channels, Flax msgpack, optimizer state as masked 2-tuple
    sd = st.state_dict()
    flax_sd = {"step": np.int32(7), "params": to_flax_layout(sd["params"]), "ema_params": to_flax_layout(sd["ema_params"]),
               "opt_state": {"0": {"inner_state": {"0": {"count": np.int32(7), "mu": to_flax_layout(sd["opt_state"]["mu"]),
                                                         "nu": to_flax_layout(sd["opt_state"]["nu"])}}}}}
    assert flax_sd["params"]["score_model"]["conv_in"]["kernel"].shape[2] == 15
    ck.save_flax(str(d / "ckpt-9.flax"), flax_sd)
    assert ck.checkpoint_numbers(str(d)) == [1, 2, 9]
    back = ck.restore_dict(os.path.join(str(d), "ckpt-9"))
    assert back["step"] == 7 and "mu" in back["opt_state"]
    st3 = TrainState.create(apply_fn=None, variables={"params": _tiny_tree()}, device="cpu")
    st3.flat.zero_()
    st3.load_state_dict(back)
    assert torch.equal(st3.flat, st.flat) and torch.equal(st3.ema, st.ema) and st3.step == 7


def test_verify_checkpoint_on_synthetic_flax_with_aliased_names(tmp_path):
    """tests/verify_checkpoint.py (the one-command oracle pin for whoever holds a released checkpoint) on a synthetic
    Flax msgpack written the way the reference might: gamma network under its attribute names l1 .. l3_c
    (ldm/model_mulan_epsilon.py:493-512), optax.chain(masked(adamw), masked(adamw)) optimizer state in CLU / Flax
    state-dict form with masked-out leaves as empty nodes (ldm/experiment.py:151-173), one array in Flax's chunked
    form.  The loader must resolve all of it; the verifier must report a matching tree and the oracle's BPD."""
    import copy
    from mulan_amd import checkpoint as ck
    from mulan_amd.config import load_config_file
    from mulan_amd.train_state import TrainState
    from oracle import torch_ref as tr
    from tests import verify_checkpoint as vc
    config = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    config.model.sm_n_embd = 32
    config.model.sm_n_layer = 1
    config.model.forward_n_layer = 1
    ocfg = vc.oracle_cfg(config)
    path = str(tmp_path / "ckpt-9.flax")
    sd, ref, trees = vc.write_synthetic_flax_checkpoint(path, config, seed=4, step=9)
    ema, mu, nu = trees["ema"], trees["mu"], trees["nu"]
    k = ema["score_model"]["dense0"]["kernel"]

    back = ck.restore_dict(path)
    assert set(back["ema_params"]["gamma"]) == set(ck.GAMMA_NET_ALIASES.values())
    assert np.array_equal(back["ema_params"]["score_model"]["dense0"]["kernel"], k)
    assert set(back["opt_state"]["mu"]) == {"score_model", "encoder_model", "gamma"}
    assert np.array_equal(back["opt_state"]["nu"]["gamma"]["dense_2"]["kernel"], nu["gamma"]["l2"]["kernel"])
    assert np.array_equal(back["opt_state"]["mu"]["score_model"]["conv_in"]["bias"], mu["score_model"]["conv_in"]["bias"])

    vdm, tmpl = vc.expected_tree(config)
    st = TrainState.create(apply_fn=None, variables={"params": vdm.init(__import__("mulan_amd.rng", fromlist=["x"]).PRNGKey(0))},
                           device="cpu")
    st.load_state_dict(back)                            # strict: every leaf of params / ema / mu / nu is present
    assert st.step == 9
    assert torch.equal(st.ema_params["gamma"]["dense_out_b"]["kernel"], torch.tensor(ema["gamma"]["l3_b"]["kernel"]))

    img = np.random.default_rng(1).integers(0, 256, (1, 32, 32, 3), dtype=np.uint8)
    lines = []
    res = vc.verify(path, config, img, n_timesteps=2, use_gpu=False, seed=3, log=lines.append)
    assert res["tree_ok"] and len(res["oracle"]) == 1 and np.isfinite(res["oracle"][0]) and res["hip"] == []
    rng = np.random.default_rng(3)
    t0, raw = float(rng.random()), rng.gamma(1.0 / 15, size=(10, 2, 50))
    e0, e = rng.standard_normal((2, 3072)), rng.standard_normal((2, 3072))
    half = tr.tree_map(lambda t: (t.detach() * 0.5).float().double(), ref)        # what the checkpoint's ema_params hold
    want = tr.mulan_forward(half, ocfg, torch.tensor(np.repeat(img, 2, axis=0)), t0, torch.tensor(raw),
                            torch.tensor(e0).view(2, 32, 32, 3), torch.tensor(e).view(2, 32, 32, 3))
    assert abs(res["oracle"][0] - float(want["bpd"])) < 1e-9 * abs(float(want["bpd"]))

    broken = copy.deepcopy(sd)
    del broken["ema_params"]["gamma"]["l3_c"]
    broken["ema_params"]["score_model"]["conv_in"]["bias"] = np.zeros(7, np.float32)
    ck.save_flax(str(tmp_path / "ckpt-10.flax"), broken)
    res2 = vc.verify(str(tmp_path / "ckpt-10.flax"), config, img, n_timesteps=2, use_gpu=False, log=lines.append)
    assert not res2["tree_ok"] and ("gamma", "dense_out_c", "kernel") in res2["missing"]
    assert any(p == ("score_model", "conv_in", "bias") for p, _, _ in res2["mismatched"])
    assert vc.main(["--ckpt", path, "--config", os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"),
                    "--set", "model.sm_n_embd=32", "--set", "model.sm_n_layer=1", "--set", "model.forward_n_layer=1",
                    "--data", "synthetic", "--n-images", "0", "--no-gpu"]) == 0


def test_reference_python_surface_is_importable():
    """every name SURVEY section 8(b) lists under the kept Python API resolves, with the reference's call signatures"""
    import inspect
    import ldm.experiment, ldm.experiment_vdm, ldm.train_state, ldm.model_vdm, ldm.model_mulan_velocity  # noqa: E401
    import ldm.model_mulan_epsilon, ldm.ldm_unet, ldm.notebook_utils, ldm.dataset, ldm.main, ldm.eval_bpd  # noqa: E401
    assert inspect.isclass(ldm.experiment.Experiment) and inspect.isclass(ldm.experiment_vdm.Experiment_VDM)
    assert inspect.isclass(ldm.train_state.TrainState)
    for name in ("train_step", "eval_step", "loss_fn", "sample_fn", "get_model_and_params", "train_and_evaluate", "evaluate"):
        assert hasattr(ldm.experiment_vdm.Experiment_VDM, name), name
    assert {"VDMConfig", "VDMOutput", "VDM", "ScoreUNet", "EncDec"} <= set(dir(ldm.model_vdm))
    assert list(inspect.signature(ldm.model_vdm.ScoreUNet.apply).parameters)[:6] == \
        ["self", "params", "z", "g_t", "conditioning", "deterministic"]          # ldm/model_vdm.py:314
    assert inspect.signature(ldm.model_vdm.ScoreUNet.apply).parameters["time"].default is False
    assert list(inspect.signature(ldm.ldm_unet.UNet.apply).parameters)[:6] == \
        ["self", "params", "z", "g_t", "conditioning", "deterministic"]          # ldm/ldm_unet.py:69
    assert list(inspect.signature(ldm.model_mulan_epsilon.UnetEncoder.apply).parameters)[:4] == \
        ["self", "params", "z", "deterministic"]                                 # ldm/model_mulan_epsilon.py:105
    assert list(inspect.signature(ldm.model_mulan_epsilon.NoiseSchedule_polynomial_fixedend.apply).parameters)[:4] == \
        ["self", "params", "embedding", "t"]                                     # ldm/model_mulan_epsilon.py:602
    assert ldm.model_mulan_epsilon.GAMMA_NETWORKS["poly_fixedend"] is ldm.model_mulan_epsilon.NoiseSchedule_polynomial_fixedend
    assert callable(ldm.model_mulan_velocity.VDM) and callable(ldm.model_mulan_epsilon.VDM)
    for name in ("Experiment_Colab", "eval_bpd_dense_sampling", "eval_bpd_sparse_sampling", "eval_bpd_ode"):
        assert hasattr(ldm.notebook_utils, name), name
    assert callable(ldm.dataset.create_dataset) and callable(ldm.dataset.create_one_time_eval_dataset)


def test_profiling_hooks_are_cheap_noops_without_a_profiler():
    """config.training.profile / StepTraceAnnotation counterparts (ldm/experiment.py:230-232,243): ranges nest, the
    Profile action opens a window of num_profile_steps calls after first_profile and emits phase ranges only inside it"""
    from mulan_amd import profiling
    with profiling.trace_range("train step 0"):
        with profiling.trace_range("forward"):
            pass
    p = profiling.Profile(num_profile_steps=2, first_profile=3)
    seen = []
    for step in range(1, 8):
        p(step)
        seen.append(p.active)
    assert seen == [False, False, True, True, False, False, False]
    import contextlib
    assert isinstance(p.phase("forward"), contextlib.nullcontext)


def test_partial_restore_overlays_only_present_keys():
    from mulan_amd.experiment import restore_partial
    from mulan_amd.train_state import TrainState
    st = TrainState.create(apply_fn=None, variables={"params": _tiny_tree()}, device="cpu")
    before = st.params["gamma"]["dense_1"]["kernel"].detach().clone()
    new = torch.ones(5, 3)
    restore_partial(st, {"params": {"encoder_model": {"dense": {"kernel": new}}}})
    assert torch.equal(st.params["encoder_model"]["dense"]["kernel"].detach(), new)
    assert torch.equal(st.params["gamma"]["dense_1"]["kernel"].detach(), before)


def test_model_init_matches_reference_parameter_counts():
    """CIFAR config 71.15 M parameters (score 30.60 + encoder 2.63 + gamma 37.92)."""
    from mulan_amd.config import load_config_file
    from mulan_amd.model import VDMConfig, make_vdm, to_flax_layout, tree_leaves
    from mulan_amd.rng import PRNGKey
    c = load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    vdm = make_vdm(c.vdm_type, VDMConfig(**c.model.to_dict()))
    tree = to_flax_layout(vdm.init(PRNGKey(1)))
    count = lambda t: sum(v.numel() for _, v in tree_leaves(t))
    assert count(tree["score_model"]) == 30_603_267
    assert count(tree["encoder_model"]) == 2_632_883
    assert count(tree["gamma"]) == 37_917_696
    assert tree["score_model"]["dense0"]["kernel"].shape == (178, 512)
    assert tree["score_model"]["up.block_3"]["conv1"]["kernel"].shape == (3, 3, 256, 128)
    assert float(tree["score_model"]["conv_out"]["kernel"].abs().sum()) == 0       # zero-init (model_vdm.py:382)
    assert float(tree["gamma"]["dense_out_a"]["kernel"].abs().sum()) == 0         # model_mulan_epsilon.py:495-500


def test_ode_evaluator_host_logic():
    """bits/dim offsets of the two dequantisations (ldm/notebook_utils.py:446-458) and the importance-weighted
    bound's logsumexp against the oracle / scipy; image grid layout of utils.generate_image_grids"""
    import math
    import numpy as np
    from scipy.special import logsumexp
    from mulan_amd import checkpoint as ck
    from mulan_amd import evaluators as ev
    from oracle import torch_ref as tr
    for deq, n in (("uniform", 1), ("tn", 1), ("tn", 20)):
        assert abs(ev._get_bpd_offset(deq, n) - tr.bpd_offset(deq, n)) < 1e-12
    assert ev._get_bpd_offset("uniform", 1) == 7.0
    a = np.random.default_rng(0).standard_normal((5, 7)) * 30
    assert np.allclose(ev._logsumexp0(a), logsumexp(a, axis=0), rtol=1e-13)
    assert abs(ev.TN_LOG_Z - math.log(0.9974613)) < 1e-15
    imgs = np.arange(5 * 2 * 2 * 3, dtype=np.uint8).reshape(5, 2, 2, 3)           # 5 images -> 2 x 2 grid
    g = ck.generate_image_grids(imgs)
    assert g.shape == (4, 4, 3)
    assert np.array_equal(g[:2, :2], imgs[1]) and np.array_equal(g[:2, 2:], imgs[0])   # rows run right to left
    assert np.array_equal(g[2:, :2], imgs[3]) and np.array_equal(g[2:, 2:], imgs[2])


def test_imagenet32_pickle_reader(tmp_path, monkeypatch):
    """downsampled-ImageNet 32x32 archives: channel-major rows -> [N, 32, 32, 3], labels zeroed, unshuffled
    validation order for create_one_time_eval_dataset"""
    import pickle
    import numpy as np
    from mulan_amd import data
    rng = np.random.default_rng(0)
    val = rng.integers(0, 256, (6, 3072), dtype=np.uint8)
    tr1 = rng.integers(0, 256, (5, 3072), dtype=np.uint8)
    (tmp_path / "Imagenet32_val").mkdir()
    (tmp_path / "Imagenet32_train").mkdir()
    with open(tmp_path / "Imagenet32_val" / "val_data", "wb") as f:
        pickle.dump({"data": val, "labels": [1, 2, 3, 4, 5, 1000]}, f)
    with open(tmp_path / "Imagenet32_train" / "train_data_batch_1", "wb") as f:
        pickle.dump({"data": tr1, "labels": [7] * 5}, f)
    monkeypatch.setenv("MULAN_DATA_DIR", str(tmp_path))
    x, y = data.load_arrays("imagenet32", train=False)
    assert x.shape == (6, 32, 32, 3) and list(y) == [0] * 6          # labels dropped like the reference (label_key=None)
    assert np.array_equal(x[2, 5, 7], val[2].reshape(3, 32, 32)[:, 5, 7])
    xt, yt = data.load_arrays("imagenet32", train=True)          # only the first training shard is present
    assert xt.shape == (5, 32, 32, 3) and set(yt) == {0}
    stream = data.BatchStream("imagenet32", 4, train=False, device="cpu", one_pass=True)
    first = next(iter(stream))
    assert np.array_equal(first["images"].numpy(), x[:4])


def test_cifar10_aug_stream_flags_augmented_images(tmp_path, monkeypatch):
    """cifar10_aug (ldm/dataset.py:358-376): flips / 90-degree rotations on the train stream only, conditioning = 1 for
    augmented images, which are then a flip / rotation of an archive image; the eval stream is untouched"""
    import pickle
    import numpy as np
    from mulan_amd import data
    rng = np.random.default_rng(1)
    d = tmp_path / "cifar-10-batches-py"
    d.mkdir()
    raw = rng.integers(0, 256, (10, 3072), dtype=np.uint8)
    for name in [f"data_batch_{i}" for i in range(1, 6)] + ["test_batch"]:
        with open(d / name, "wb") as f:
            pickle.dump({b"data": raw, b"labels": list(range(10))}, f)
    monkeypatch.setenv("MULAN_DATA_DIR", str(tmp_path))
    imgs = raw.reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    variants = {}
    for i, im in enumerate(imgs):
        for fl in (False, True):
            a = im[:, ::-1] if fl else im
            for k in range(4):
                variants.setdefault(np.rot90(a, k=k, axes=(0, 1)).tobytes(), []).append((i, fl, k))
    st = data.BatchStream("cifar10_aug", 8, train=True, device="cpu", seed=3, substeps=2)
    b = next(st)
    assert b["images"].shape == (2, 8, 32, 32, 3) and b["conditioning"].dtype == torch.uint8
    x, c = b["images"].reshape(-1, 32, 32, 3).numpy(), b["conditioning"].reshape(-1).numpy()
    assert 0 < c.sum() < len(c)
    for im, flag in zip(x, c):
        hits = variants[im.tobytes()]
        assert any(bool(fl or k) == bool(flag) for _, fl, k in hits)
    ev = data.BatchStream("cifar10_aug", 4, train=False, device="cpu")
    e = next(ev)
    assert int(e["conditioning"].sum()) == 0 and np.array_equal(e["images"].numpy(), imgs[:4])


def test_plane_hand_over_gating_and_backward_scope(monkeypatch):
    """host logic of the two round-2 schedulers, no GPU needed: which (GroupNorm, convolution) pairs take the plane
    hand-over, and that the backward scope of the weight-gradient stream arms / disarms itself (also when the body
    raises); the shared-chip block count of the weight-gradient launches is an ARGUMENT derived from that scope
    (ops._share_chip), no library-global switch is touched."""
    from mulan_amd import ops
    monkeypatch.setattr(ops, "CONV_MODE", "f16x3")
    monkeypatch.setattr(ops, "GN_CONV_PLANES", True)
    ok = ops.gn_conv_ok
    assert ok(128, 0, 128, 32) and ok(128, 128, 128, 32) and ok(256, 256, 256, 32) and ok(256, 0, 512, 32)
    assert not ok(64, 0, 128, 32)            # the plane-fed weight-gradient kernel wants 128-multiples of channels
    assert not ok(128, 0, 3, 32)             # conv_out
    assert not ok(512, 512, 128, 32)         # more than 16 maxima slots
    assert not ok(128, 0, 128, 3)            # groups that do not divide the channels
    monkeypatch.setattr(ops, "CONV_MODE", "f32")
    assert not ok(128, 0, 128, 32)
    calls = []
    monkeypatch.setattr(ops, "call", lambda name, *a: calls.append((name,) + a))
    monkeypatch.setattr(ops, "SIDE_STREAM", True)
    monkeypatch.setattr(ops, "SIDE_WGRAD_SHARE", True)
    assert not ops._side_ok(None)
    assert ops._share_chip() == 0
    with ops.weight_gradient_stream():
        assert ops._SIDE["active"] and ops._share_chip() == 1
        assert not ops._side_ok(None)        # no sink in the flat gradient buffer: stays on the current stream
    assert not ops._SIDE["active"] and ops._share_chip() == 0
    with pytest.raises(RuntimeError):
        with ops.weight_gradient_stream():
            raise RuntimeError("backward failed")
    assert not ops._SIDE["active"] and ops._share_chip() == 0
    monkeypatch.setattr(ops, "SIDE_WGRAD_SHARE", False)
    with ops.weight_gradient_stream():
        assert ops._SIDE["active"] and ops._share_chip() == 0
    monkeypatch.setattr(ops, "SIDE_STREAM", False)
    assert ops._alone() == 1                 # forward passes / evaluators: small convolution launches may take whole CUs
    with ops.weight_gradient_stream():
        assert not ops._SIDE["active"] and ops._share_chip() == 0
        assert ops._alone() == 0             # the backward pass of a train step, with or without the second stream
    assert ops._alone() == 1
    monkeypatch.setattr(ops, "SIDE_STREAM", True)
    with ops.weight_gradient_stream():
        assert ops._alone() == 0
    with pytest.raises(RuntimeError):
        with ops.weight_gradient_stream():
            raise RuntimeError("backward failed")
    assert ops._alone() == 1
    # round 5: the streaming GroupNorm forward runs in the window of images per launch where the step measured faster
    monkeypatch.setattr(ops, "GN_FWD_STREAM", True)
    monkeypatch.setattr(ops, "GN_FWD_STREAM_B", (32, 96))
    assert [ops._gn_fwd_stream_on(b) for b in (16, 31, 32, 64, 96, 97, 128)] == [False, False, True, True, True, False, False]
    monkeypatch.setattr(ops, "GN_FWD_STREAM", False)
    assert not ops._gn_fwd_stream_on(64)
    assert not any(c[0] == "mulan_set_tuning" for c in calls)       # the product path never touches the dev switches


# ------------------------------------------------------------------------------ round 3: schedule, stream sync, hazards
def test_lr_schedule_matches_oracle_and_reference_edge_cases():
    """H1: the product's Experiment.get_lr_schedule against the oracle's restatement of ldm/experiment.py:106-129 for
    every step of a short run, with and without lr_decay, including the reference's own edge cases (optax's
    linear_schedule with transition_steps <= 0 is the constant init_value: no warm-up and no decay -> lr 0.0)."""
    import types
    from mulan_amd.config import ConfigDict
    from mulan_amd.experiment import Experiment
    from oracle import mulan_np as onp
    for warm, decay, total in [(100, False, 1000), (2, False, 10), (5, True, 40), (0, True, 20), (0, False, 20),
                               (10, True, 10)]:
        cfg = ConfigDict(dict(optimizer=dict(learning_rate=2e-4, lr_decay=decay),
                              training=dict(num_steps_lr_warmup=warm, num_steps_train=total)))
        sched = Experiment.get_lr_schedule(types.SimpleNamespace(config=cfg))
        for step in list(range(0, total + 5)) + [10 * total]:
            want = onp.lr_schedule(step, 2e-4, warm, decay, total)
            assert abs(sched(step) - want) <= 1e-18 + 1e-12 * abs(want), (warm, decay, total, step, sched(step), want)
    cfg = ConfigDict(dict(optimizer=dict(learning_rate=2e-4, lr_decay=False),
                          training=dict(num_steps_lr_warmup=100, num_steps_train=1000)))
    sched = Experiment.get_lr_schedule(types.SimpleNamespace(config=cfg))
    assert sched(0) == 0.0 and abs(sched(50) - 1e-4) < 1e-15 and sched(100) == 2e-4 and sched(99999) == 2e-4


def test_train_stream_ranks_stay_on_one_permutation_for_uneven_sizes(tmp_path):
    """len(dataset) % world != 0: every rank ends its epoch at the same position (the tail is dropped), so after many
    epochs the ranks still partition ONE permutation per epoch -- no cross-rank duplicates in any epoch -- and seek()
    puts a fresh stream where a resumed run left off."""
    from mulan_amd import data
    N, world, bs = 103, 4, 20                       # 5 samples per rank and draw, 25 per rank and epoch, 3 dropped
    imgs = np.zeros((N, 32, 32, 3), dtype=np.uint8)
    imgs[:, 0, 0, 0] = np.arange(N)
    np.savez(tmp_path / "d.npz", images=imgs)
    name = f"npz:{tmp_path / 'd.npz'}"
    streams = [data.BatchStream(name, bs, train=True, device="cpu", seed=3, rank=r, world=world) for r in range(world)]
    ident = lambda b: b["images"][:, 0, 0, 0].long().tolist()
    for epoch in range(7):
        seen = []
        for _ in range(5):
            for st in streams:
                seen += ident(next(st))
        assert len(seen) == 100 and len(set(seen)) == 100, epoch          # disjoint shards of one permutation
        assert {st.epoch for st in streams} == {epoch} or {st.pos for st in streams} == {25}
    assert len({(st.epoch, st.pos) for st in streams}) == 1
    # resume: a fresh stream positioned after 7 epochs + 2 draws yields what the running one yields next
    for st in streams:
        next(st), next(st)
    fresh = data.BatchStream(name, bs, train=True, device="cpu", seed=3, rank=1, world=world)
    fresh.seek((7 * 5 + 2) * 5)
    assert ident(next(fresh)) == ident(next(streams[1]))
    # ... including the cifar10_aug flips / rotations (they came from a generator seek() did not restore);
    # the pictures get an asymmetric mark so that every flip / rotation shows
    imgs[:, 1, 2, 1] = 200
    np.savez(tmp_path / "e.npz", images=imgs)
    name2 = f"npz:{tmp_path / 'e.npz'}"
    run = data.BatchStream(name2, bs, train=True, device="cpu", seed=3, rank=2, world=world)
    run.augment = True
    for _ in range(9):
        next(run)
    res = data.BatchStream(name2, bs, train=True, device="cpu", seed=3, rank=2, world=world)
    res.augment = True
    res.seek(9 * 5)
    a, b = next(run), next(res)
    assert torch.equal(a["images"], b["images"]) and torch.equal(a["conditioning"], b["conditioning"])
    assert 0 < int(a["conditioning"].sum()) <= 5 or int(next(run)["conditioning"].sum()) > 0     # (augmentations do happen)


def test_no_mfma_data_hazard_in_the_built_objects(built_lib):
    """The inline-asm matrix instructions of conv3x3_f16x3_v3.hip are invisible to hipcc's hazard recognizer: a VALU
    write of an A / B operand right in front of one, or an accumulator read too soon after one, would corrupt results
    silently.  tools/mfma_hazard_scan.py walks the gfx950 disassembly of every object of this build (so a compiler
    bump or an edit that moves a register copy is caught here, on the CPU); its two rules are checked on a synthetic
    listing first."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import mfma_hazard_scan as scan
    finally:
        sys.path.pop(0)
    bad = """
0000000000000000 <kernel>:
\tv_mov_b32_e32 v2, v9                                       // 000000000000: 7E040309
\tv_mfma_f32_16x16x32_f16 a[0:3], v[2:5], v[6:9], a[0:3]     // 000000000004: 00000000
\ts_nop 3                                                    // 000000000008: 00000000
\tv_accvgpr_read_b32 v1, a2                                  // 00000000000c: 00000000
"""
    assert sorted(f[1] for f in scan.scan_text(bad)) == ["R1", "R2"]
    good = bad.replace("v_mov_b32_e32 v2, v9 ", "v_mov_b32_e32 v2, v9\n\ts_nop 1 ").replace("s_nop 3", "s_nop 5")
    assert scan.scan_text(good) == []
    if not os.path.exists(scan.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    found = scan.scan_objects()
    assert found == [], found[:5]


def test_bench_self_launches_for_several_gpus(monkeypatch):
    """`python bench.py --gpus 8` without WORLD_SIZE in the environment starts `python -m torch.distributed.run
    --nproc-per-node 8 ... bench.py <same arguments>` as a CHILD process (the parent never initialises the GPU) and
    exits with the child's code; with WORLD_SIZE set (it is one of the ranks) it does not."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    # the rendezvous port is torchrun's own choice (--standalone: no bind / close / reuse gap), on the loop-back address
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch.cuda" not in sys.modules or not __import__("torch").cuda.is_initialized()


def test_load_flax_reads_bytes_it_did_not_write():
    """tests/golden/tiny_state.flax was assembled byte by byte from the msgpack specification by
    tests/golden/make_flax_fixture.py (struct.pack only: neither `checkpoint.save_flax` nor the msgpack library wrote
    it), the way flax 0.7.0 serialises the reference's TrainState (ldm/experiment.py:210-214, notebook_utils.py:31-37):
    ext-1 ndarrays, an ext-3 numpy scalar, a chunked array, the gamma network under its attribute names, one wrapping
    'params' level, optax.chain(masked(adamw), masked(adamw)) with masked-out leaves as empty maps."""
    import msgpack
    from mulan_amd import checkpoint as ck
    from tests.golden import make_flax_fixture as fx
    path = os.path.join(ROOT, "tests", "golden", "tiny_state.flax")
    raw = open(path, "rb").read()
    assert raw == fx.build_bytes() and len(raw) < 2048
    # the hand-assembled headers are the ones msgpack itself emits: decode (extension payloads kept opaque) and
    # re-encode with the library -> the same bytes
    generic = msgpack.unpackb(raw, raw=False, strict_map_key=False)
    assert msgpack.packb(generic, use_bin_type=True) == raw
    assert isinstance(generic["step"], msgpack.ExtType) and generic["step"].code == 3
    assert generic["ema_params"]["score_model"]["norm_out"]["scale"]["__msgpack_chunked_array__"] is True
    # ldm/experiment.py:160-170: masked AdamW number 0 holds the score_model leaves, number 1 the rest; a masked-out leaf
    # is an empty map; each adamw's add_decayed_weights(mask=...) state is itself a MaskedState: {'inner_state': {}}
    adam0, adam1 = generic["opt_state"]["0"]["inner_state"], generic["opt_state"]["1"]["inner_state"]
    assert adam0["0"]["mu"]["gamma"]["l1"]["bias"] == {} and adam0["0"]["mu"]["gamma"]["l1"]["kernel"] == {}
    assert adam1["0"]["mu"]["score_model"]["conv_out"]["kernel"] == {} and adam1["0"]["nu"]["score_model"]["norm_out"]["scale"] == {}
    assert isinstance(adam0["0"]["mu"]["score_model"]["conv_out"]["kernel"], msgpack.ExtType)
    assert isinstance(adam1["0"]["mu"]["gamma"]["l1"]["kernel"], msgpack.ExtType)
    assert adam0["1"] == {"inner_state": {}} and adam1["1"] == {"inner_state": {}} and adam0["2"] == {} and adam1["2"] == {}
    got = ck.load_flax(path)
    want = fx.expected_tree()
    assert got["step"] == 223 and isinstance(got["step"], int)

    def same(a, b, where=""):
        assert isinstance(a, dict) == isinstance(b, dict), where
        if isinstance(b, dict):
            assert sorted(a) == sorted(b), (where, sorted(a), sorted(b))
            for k in b:
                same(a[k], b[k], where + "/" + k)
        else:
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), where
    for key in ("params", "ema_params"):
        same(got[key], want[key], key)
    same(got["opt_state"]["mu"], want["opt_state"]["mu"], "mu")
    same(got["opt_state"]["nu"], want["opt_state"]["nu"], "nu")


def test_committed_bench_record_keeps_the_driver_contract():
    """profiles/r06_bench_n1.json.log is the line `python bench.py` printed on an MI355X with the round-6 code: the keys
    the driver and the judge read (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
    scaling / vs_baseline / dtype / data / config.workload, the `roofline` and `cpu_baseline` objects) are all there and
    consistent with each other, and so are the round-6 additions (dtype names the split scheme,
    step_roofline_frac, the as-run fraction and a non-null share next to the kernel's own figures, the precision probe, the
    chip's clock / power / allocator record, config #1's GPU twin)."""
    import json
    rec = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_n1.json.log")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["metric"] == "train images/sec" and rec["unit"] == "images/s" and rec["higher_is_better"] is True
    assert rec["n_gpus"] == 1 and rec["scaling"] == "weak" and rec["vs_baseline"] is None and rec["data"] == "synthetic"
    assert rec["dtype"].startswith("f32 (f16x3 split: 3 fp16 MFMA passes") and "workload" in rec["config"] and "model" not in rec["config"]
    B = rec["config"]["global_batch"]
    assert abs(rec["value"] - B / (rec["ms_per_step"] * 1e-3)) < 0.01 * rec["value"]
    roof = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_as_run", "share_of_step", "as_run"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["traffic"] is None or roof["traffic"] > 1e8
    assert roof["frac_as_run"] == roof["as_run"]["frac"] < roof["frac"] and 0 < roof["share_of_step"] < roof["as_run"]["share_of_step"] < 1
    assert abs(rec["step_roofline_frac"] - rec["model_roofline_frac"]) < 1e-9 and 0.2 < rec["step_roofline_frac"] < 1
    pr = rec["precision"]
    assert 0 < pr["split_error_vs_fp64"] <= 1.5 * pr["f32_error_vs_fp64"] < 1e-5       # the split scheme is not narrower than fp32
    chip = rec["chip"]
    assert chip["allocator_peak_gb"] > 0 and "sclk_mhz" in chip and "socket_power_w" in chip and "power_cap_w" in chip
    cpu = rec["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] == "port" and cpu["cores"] >= 1
    assert set(rec["configs"]) >= {"1", "3", "4", "5", "sampler", "ode"}
    assert rec["configs"]["1"]["global_batch"] == 2 and rec["configs"]["1"]["steps"] == 10
    assert all("chip" in rec["configs"][k] for k in ("3", "4"))
    assert rec["multi_gpu"] is None                              # (N = 1; the N > 1 fields: tests/test_gpu_model.py, bench_ranks)


def test_step_form_choice_only_explicit_requests_make_a_failed_capture_fatal():
    """the multi-rank default heuristic may pick the replayed step, but only an explicit request
    (config.training.hip_graph=True or MULAN_HIP_GRAPH=1) turns a failed capture into an error -- a default-chosen
    replay must fall back to the eager step with a warning, or one rank raising would hang the others' collectives."""
    from mulan_amd.experiment import choose_step_form as f
    # one rank, nothing said: replay, not required
    assert f(None, "", 1, 128, 128, False) == (True, False)
    # several ranks, small local batch: the heuristic picks the replay -- still not required
    assert f(None, "", 8, 512, 128, False) == (True, False)          # 64 per GPU < 96
    assert f(None, "", 8, 1024, 128, False) == (False, False)        # 128 per GPU: eager overlapped step
    assert f(None, "", 8, 1024, 256, False) == (False, False)        # E = 256 counts four-fold
    assert f(None, "", 8, 128, 256, False) == (True, False)          # 16 per GPU x 4 = 64 < 96
    # explicit requests
    assert f(True, "", 8, 1024, 128, False) == (True, True)
    assert f(None, "1", 8, 1024, 128, False) == (True, True)
    assert f(False, "1", 8, 1024, 128, False) == (False, False)      # the config's False wins over the environment
    assert f(None, "0", 1, 128, 128, False) == (False, False)
    assert f(True, "0", 1, 128, 128, False) == (False, True)         # asked for by config, switched off by the environment
    # the opt-in overlap forms keep the replay as their default
    assert f(None, "", 8, 1024, 128, True) == (True, False)
