#!/usr/bin/env python3
"""MuLAN training-throughput bench on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) it is one of the ranks; started plainly
(`python bench.py --gpus 8`) it launches that command itself as a child process and relays rank 0's JSON line -- the
reference needs no launcher either (ldm/experiment.py:89-95: pmap inside one process).

One "step" = one full train step of the hot path (forward ELBO + backward + RCCL gradient all-reduce + AdamW/EMA)
on one synthetic CIFAR-shaped uint8 batch that is already resident in HBM.  Default workload at every N: BASELINE.json
configs[1] -- MuLAN-epsilon, ldm/configs/cifar10-conditioned.py (E=128, 32+2+33 ResBlocks), batch 128 per GPU
(weak scaling: global batch 128*N), fp32 (the reference forces fp32 matmuls, ldm/main.py:39).
`--global-batch G` fixes the GLOBAL batch instead (strong scaling, G/N images per GPU): BASELINE configs[2] is
`--vdm-type mulan_velocity --global-batch 512`, configs[3] `--config ldm/configs/imagenet32.py --vdm-type mulan_velocity
--vfe --global-batch 1024`.
Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel: the 3x3-conv implicit GEMM) and
`cpu_baseline` (oracle port on the host cores, bounded samples) objects.  At N = 1 the line also carries, under
"configs", at every N (`--no-also-configs` skips them), a short timing of the other BASELINE configurations -- #3
MuLAN-velocity at GLOBAL batch 512 (strong scaling), #4 ImageNet-32 velocity_from_epsilon at 128 images per GPU (global
1024 at N = 8), #5 the dense VLB evaluator (T = 1000) with the test images sharded over the ranks -- each training entry
with the convolution kernel's own roofline fraction, and of the SURVEY 8(f) widenings "sampler" (ancestral sampler) and
"ode" (exact-likelihood evaluator: function evaluations/s, adaptive RK45 at rtol = atol = 1e-5, n_is = 1).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 MFMA
PEAK_BF16_MFMA_TFLOPS = 2500.0        # same guide: dense bf16 MFMA (~2.5 PF)
FWD_GFLOP_PER_IMAGE = 57.78           # score 53.37 + encoder 4.33 + gamma 0.076 (CIFAR config)
FWD_GFLOP_BY_WIDTH = {128: 57.78, 256: 228.58}   # CIFAR config / ImageNet-32 config
SCORE_FWD_GFLOP_BY_WIDTH = {128: 53.37, 256: 212.32}   # the score U-Net alone (sampler / ODE: no encoder pass per step)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--per-gpu-batch", type=int, default=128)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: fixed global batch split over the ranks (BASELINE configs[2]: 512)")
    ap.add_argument("--vfe", action="store_true", help="model.velocity_from_epsilon=True (BASELINE configs[3])")
    ap.add_argument("--vdm-type", default="mulan_epsilon")
    ap.add_argument("--config", default=os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--traffic", type=float, default=None, help="HBM bytes per dominant-kernel launch from a PMC pass")
    ap.add_argument("--cpu-timeout", type=int, default=420)
    ap.add_argument("--no-f32-mode", action="store_true", help="skip the reference measurement with exact-fp32 MFMA convs")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--also-configs", dest="also_configs", action="store_true", default=None,
                    help="time BASELINE configs #3, #4, #5 after the headline (default: on at N = 1)")
    ap.add_argument("--no-also-configs", dest="also_configs", action="store_false")
    ap.add_argument("--also-steps", type=int, default=6, help="timed steps per extra training configuration")
    ap.add_argument("--configs-small", action="store_true",
                    help="test-sized extra configurations (global batch 32 / 8 per GPU, T = 64, tiny sampler / ODE batches): the "
                         "control flow of \"configs\" in seconds; the numbers it prints are not the configurations' numbers")
    ap.add_argument("--small-depth", type=int, default=0, help=argparse.SUPPRESS)   # tests only: ResBlocks per U-Net stage
    a = ap.parse_args()
    global TEST_DEPTH
    TEST_DEPTH = a.small_depth
    return a


TEST_DEPTH = 0


ALSO_WARMUP = 8          # untimed steps in front of the timed ones of every extra training configuration


def _test_depth(config):
    """(tests only, --small-depth N) N ResnetBlocks per U-Net stage instead of the configuration's 32 / 4: the control flow
    of a multi-rank run in seconds.  Never set by the driver; the JSON line says so (`test_depth`) when it is."""
    if TEST_DEPTH > 0:
        config.model.sm_n_layer = TEST_DEPTH
        config.model.forward_n_layer = min(TEST_DEPTH, int(config.model.forward_n_layer))
    return config


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run` with the same arguments
    as a CHILD process (this parent never touches the GPU), pass its output through and exit with its code"""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # --standalone: torchrun picks the rendezvous port itself (no bind / close / reuse gap on a shared box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(a.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching", " ".join(cmd), file=sys.stderr, flush=True)
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def cpu_baseline(cfg_path, vdm_type, batch, steps):
    """The oracle's torch port of the train step (fwd + bwd + AdamW) in fp32 on all host cores: the two samples
    BASELINE.md section 3 names -- the metric's configuration (MuLAN, `steps` steps of batch `batch`; default 3 x 8)
    and BASELINE configs[0] (plain model_vdm.VDM, 10 steps of batch 2).  NOT the JAX reference (not installable here)."""
    import numpy as np
    import torch
    from mulan_amd.config import load_config_file
    from oracle import torch_ref as tr
    config = load_config_file(cfg_path)
    m = config.model
    ocfg = dict(vdm_type=vdm_type, n_embd=m.sm_n_embd, n_layer=m.sm_n_layer, forward_n_layer=m.forward_n_layer,
                latent_k=m.get("latent_k", 15), unet_type=m.unet_type)
    try:
        cores = len(os.sched_getaffinity(0))      # cores this job may actually use (cgroup/affinity aware)
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("MULAN_CPU_THREADS", "64"))))
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    keep = float(np.float32(1.0 - m.sm_pdrop))
    enc_names = [f"down.block_{i}" for i in range(m.forward_n_layer)] + ["mid.block_1", "mid.block_2"]
    sc_names = ([f"down.block_{i}" for i in range(m.sm_n_layer)] + ["mid.block_1", "mid.block_2"]
                + [f"up.block_{i}" for i in range(m.sm_n_layer + 1)])

    def timed(step_fn, n, bsz):
        step_fn()                                # untimed warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        for _ in range(n):
            step_fn()
        dt = time.perf_counter() - t0
        return bsz * n / dt, dt

    def adamw(params):
        leaves = [l.requires_grad_(True) for _, l in tr.tree_leaves(params)]
        return torch.optim.AdamW(leaves, lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)

    # ---- sample A: the metric's configuration
    params = tr.init_params(ocfg, seed=0, dtype=torch.float32)
    opt = adamw(params)

    def masks(names, bsz):
        return {n: torch.from_numpy(rng.random((bsz, 32, 32, m.sm_n_embd)) < keep) for n in names}

    def draw(bsz):
        x = torch.from_numpy(rng.integers(0, 256, (bsz, 32, 32, 3)).astype(np.uint8))
        e0 = torch.from_numpy(rng.standard_normal((bsz, 32, 32, 3)).astype(np.float32))
        e = torch.from_numpy(rng.standard_normal((bsz, 32, 32, 3)).astype(np.float32))
        return x, e0, e

    def step_a():
        x, e0, e = draw(batch)
        raw = torch.from_numpy(rng.gamma(1.0 / 15, size=(10, batch, 50)).astype(np.float32))
        out = tr.mulan_forward(params, ocfg, x, float(rng.random()), raw, e0, e, enc_masks=masks(enc_names, batch),
                               score_masks=masks(sc_names, batch), keep=keep, dtype=torch.float32)
        opt.zero_grad(set_to_none=True)
        out["bpd"].backward()
        opt.step()
    ips_a, dt_a = timed(step_a, steps, batch)
    del params, opt

    # ---- sample B: BASELINE configs[0], plain VDM (model_vdm.py) with the learnable monotone schedule, batch 2 x 10 steps
    full = tr.init_params(dict(ocfg, vdm_type="mulan_velocity"), seed=1, dtype=torch.float32)
    pv = {"score_model": full["score_model"]}
    pv["score_model"]["dense0"]["kernel"] = pv["score_model"]["dense0"]["kernel"][:m.sm_n_embd + 1].clone()
    gg = torch.Generator().manual_seed(8)
    pv["gamma"] = {"l1": {"kernel": torch.tensor([[18.3]]), "bias": torch.tensor([-13.3])},
                   "l2": {"kernel": torch.randn(1, 1024, generator=gg), "bias": torch.randn(1024, generator=gg)},
                   "l3": {"kernel": torch.randn(1024, 1, generator=gg)}}
    del full
    optb = adamw(pv)
    bsz_b, steps_b = 2, 10

    def step_b():
        x, e0, e = draw(bsz_b)
        out = tr.plain_vdm_forward(pv, dict(ocfg, n_timesteps=0), x, float(rng.random()), e0, e, dtype=torch.float32)
        optb.zero_grad(set_to_none=True)
        out["bpd"].backward()
        optb.step()
    ips_b, dt_b = timed(step_b, steps_b, bsz_b)
    return {"value": ips_a, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} train steps (fwd+bwd+AdamW) of batch {batch}, full {vdm_type} CIFAR config, fp32, "
                      f"oracle/torch_ref.py on {cores} threads, {dt_a:.1f} s of CPU work; NOT the JAX reference "
                      "(JAX / Flax are not installable on this image)",
            "config0_plain_vdm": {"value": ips_b, "unit": "images/s",
                                  "sample": f"{steps_b} train steps of batch {bsz_b}, plain model_vdm.VDM (gamma_type="
                                            f"learnable_nnet, no dropout), same U-Net, fp32, {dt_b:.1f} s of CPU work"}}


class Telemetry:
    """What the chip held while a timed region ran (config #4 read 542 -> 536 -> 510 images/s over three
    driver runs with nothing on the line to tell a slower box from a slower build): shader clock, socket power, power cap
    and temperatures of THIS process's GPU, sampled from the amdgpu hwmon files of its PCI device every 10 ms by a host
    thread (reads of sysfs: nothing is launched), plus the caching allocator's high-water mark.  Every field is None where
    the file is not readable."""

    def __init__(self, device_index=0):
        self.dir = None
        try:
            import glob
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for d in glob.glob("/sys/class/drm/card*/device"):
                if want in os.path.realpath(d).lower():
                    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
                    if hw:
                        self.dir = hw[0]
                    break
        except Exception:       # noqa: BLE001  telemetry never fails a run
            self.dir = None
        self.samples, self._stop, self._thread = [], None, None

    def _read(self, name, scale):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip()) * scale
        except Exception:       # noqa: BLE001
            return None

    def sample(self):
        if self.dir is None:
            return None
        return (self._read("freq1_input", 1e-6), self._read("power1_input", 1e-6), self._read("temp2_input", 1e-3),
                self._read("temp3_input", 1e-3))

    def __enter__(self):
        import threading
        import torch
        torch.cuda.reset_peak_memory_stats()
        self.samples = []
        if self.dir is not None:
            self._stop = threading.Event()

            def run():
                while not self._stop.is_set():
                    self.samples.append(self.sample())
                    self._stop.wait(0.01)
            self._thread = threading.Thread(target=run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
        return False

    def summary(self):
        import torch
        col = lambda i: [x[i] for x in self.samples if x is not None and x[i] is not None]
        stat = lambda v, nd=0: None if not v else {"min": round(min(v), nd), "mean": round(sum(v) / len(v), nd), "max": round(max(v), nd)}
        return {"sclk_mhz": stat(col(0)), "socket_power_w": stat(col(1)),
                "power_cap_w": None if self.dir is None else self._read("power1_cap", 1e-6),
                "temp_c": {"junction": stat(col(2)), "hbm": stat(col(3))} if self.dir is not None else None,
                "samples": len(self.samples), "source": (self.dir + "/{freq1_input,power1_input,power1_cap,temp2_input,"
                                                         "temp3_input}, one read per 10 ms of the timed region") if self.dir else None,
                "allocator_peak_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}


def _latest_pmc():
    """the newest committed PMC summary of the convolution kernel (profiles/rNN_pmc_conv3x3_f16x3.json)"""
    d = os.path.join(ROOT, "profiles")
    names = sorted(f for f in os.listdir(d) if f.endswith("_pmc_conv3x3_f16x3.json")) if os.path.isdir(d) else []
    return (os.path.join(d, names[-1]), "profiles/" + names[-1]) if names else (None, None)


def conv_roofline(exp, state, batch, a, rank, world, B, E, step_s):
    """HIP events (on the launch stream) around every convolution launch of one more step as shipped ("as_run") and --
    on one rank -- of one further step with the weight-gradient stream off, so that every launch owns the chip: the
    kernel's own roofline.  Every rank runs the steps (they contain the gradient all-reduce); rank 0 records."""
    import torch
    import torch.distributed as dist
    from mulan_amd import ops
    roof = None
    # (every rank times: with the timer on a step runs eagerly, and all ranks must run the same kind of step -- the
    # eager step launches its bucket all-reduces as the backward pass completes them, a replayed one behind the graph)
    ops.KERNEL_TIMER = []
    state, _ = exp.train_step(exp._train_rng, state, batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    recs = ops.KERNEL_TIMER
    ops.KERNEL_TIMER = None
    if rank == 0:
        # an event pair around nothing still reads a few microseconds (the two records themselves): measured here and
        # taken off every launch, so the average below is the kernel's own duration (it then agrees with rocprofv3)
        ov = []
        for _ in range(64):
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            e0.record()
            ov.append((s0, e0))
        torch.cuda.synchronize()
        ev_overhead = sorted(x.elapsed_time(y) for x, y in ov)[len(ov) // 2] * 1e-3
        per = {}
        for (name, s, e, fl) in recs:
            d = per.setdefault(name, [0.0, 0.0, 0])
            d[0] += max(1e-7, s.elapsed_time(e) * 1e-3 - ev_overhead)
            d[1] += fl
            d[2] += 1
        if per:
            # The dominant kernel: one kernel source can be launched as several instantiations (the f16x3 convolution:
            # "conv3x3_f16x3_kernel" = conv3x3_f16x3_v3_kernel<0, false> in a rocprofv3 trace -- fp32 input -- and
            # "conv3x3_f16x3_kernel<planes_in>" = <0, true>, the plane-fed launches); they are summed for the choice
            # and for the roofline, and listed one by one under "symbols".
            fam = {}
            for lab, (t_, f_, n_) in per.items():
                d = fam.setdefault(lab.split("<")[0], [0.0, 0.0, 0])
                d[0] += t_; d[1] += f_; d[2] += n_
            name, (tot_t, tot_f, n) = max(fam.items(), key=lambda kv: kv[1][0])
            labels = [lab for lab in per if lab.split("<")[0] == name]
            ach = tot_f / tot_t / 1e12
            passes = 6 if "bf16x6" in name else (3 if "f16x3" in name else 0)
            bf = passes > 0
            # split modes: `passes` 16-bit MFMA passes per algorithmic product -> the scheme's ceiling in algorithmic
            # FLOP/s is the dense bf16/fp16 MFMA peak / passes; exact-fp32 MFMA mode: the fp32 MFMA peak.
            peak = PEAK_BF16_MFMA_TFLOPS / passes if bf else PEAK_F32_MFMA_TFLOPS
            # HBM bytes per launch: from the committed rocprofv3 PMC passes of this kernel (separate FETCH_SIZE /
            # WRITE_SIZE runs per launch shape, gfx950 corrections applied; tools/pmc_conv.py), not re-measured inside
            # the bench: every launch of the step is priced with the bytes of its shape (instantiation, and 128->128
            # vs the 256 <-> 128 shapes by its FLOP count) and `traffic` is the mean over the step's launches
            traffic, pmc = a.traffic, None
            pmc_file, pmc_name = _latest_pmc()
            if name == "conv3x3_f16x3_kernel" and pmc_file and E == 128:
                with open(pmc_file) as f:
                    pj = json.load(f)["shapes"]
                scale = B / 128.0
                fl_small = 2.0 * B * 1024 * 9 * 128 * 128
                hb = lambda k: pj[k]["hbm_bytes_per_launch"]
                pin_dg = ("pin_dgrad_128_128", "pin_dgrad_128_256") if "pin_dgrad_128_128" in pj else ("dgrad_128_128", "dgrad_128_256")
                table = {"conv3x3_f16x3_kernel<planes_in>": (0.5 * (hb("pin_fwd_128_128_res") + hb("pin_fwd_128_128_film")),
                                                             hb("pin_fwd_256_128_film")),
                         "conv3x3_f16x3_kernel<planes_in,dgrad>": (hb(pin_dg[0]), hb(pin_dg[1])),
                         "conv3x3_f16x3_kernel": (hb("dgrad_128_128"), hb("dgrad_128_256"))}
                byts = [scale * table[nm][0 if fl < 1.5 * fl_small else 1] for (nm, _, _, fl) in recs if nm in table]
                used = tuple(dict.fromkeys(("dgrad_128_128", "dgrad_128_256") + pin_dg +
                                           ("pin_fwd_128_128_res", "pin_fwd_128_128_film", "pin_fwd_256_128_film")))
                pmc = {"source": f"{pmc_name} (B = 128, one entry per instantiation and launch shape; scaled by batch / 128)",
                       "hbm_bytes_per_launch_by_shape": {k: pj[k]["hbm_bytes_per_launch"] for k in used},
                       "mfma_util_by_shape": {k: round(pj[k]["mfma_util"], 4) for k in used},
                       "traffic_over_algorithmic_by_shape": {k: round(pj[k]["traffic_over_algorithmic"], 3) for k in used}}
                if traffic is None and byts:
                    traffic = sum(byts) / len(byts)
            symbols = {lab: {"launches_per_step": per[lab][2], "avg_launch_us": round(per[lab][0] / per[lab][2] * 1e6, 1),
                             "tflops": round(per[lab][1] / per[lab][0] / 1e12, 1)} for lab in labels}
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "pmc": pmc,
                    "peak_note": (f"dense 16-bit MFMA 2500 TFLOP/s / {passes} passes (split operands, fp32-equivalent "
                                  "products)" if bf else "dense fp32 MFMA"),
                    "executed_mfma_tflops": round(passes * ach, 1) if bf else None,
                    "launches_per_step": n, "avg_launch_us": round(tot_t / n * 1e6, 1),
                    "event_pair_overhead_us": round(ev_overhead * 1e6, 2),
                    "avg_gflop_per_launch": round(tot_f / n / 1e9, 3),
                    "share_of_step": round(tot_t / step_s, 3),
                    "symbols": symbols,
                    "note": ("as run: the weight-gradient kernels share the chip with these launches (second stream, "
                             "balanced to run side by side), so a launch's duration includes the time it shares the "
                             "chip") if ops.SIDE_STREAM and world == 1 else None,
                    "other_kernels": {k: {"tflops": round(v[1] / v[0] / 1e12, 1), "ms_per_step": round(v[0] * 1e3, 2)}
                                      for k, v in per.items() if k not in labels}}
    # ---- the dominant kernel without a neighbour: in the step as shipped the weight-gradient kernels run on a second
    # stream beside it (ops.weight_gradient_stream) and share the chip with it, which stretches its launches; one more
    # step with that stream off gives the kernel's own duration (profiles/*_serial_kernel_stats.csv is this mode)
    if rank == 0 and world == 1 and roof is not None and ops.SIDE_STREAM:
        ops.SIDE_STREAM = False
        ops.KERNEL_TIMER = []
        state, _ = exp.train_step(exp._train_rng, state, batch)
        torch.cuda.synchronize()
        recs_alone = ops.KERNEL_TIMER
        mine = [(s_, e_, fl) for (nm, s_, e_, fl) in recs_alone if nm.split("<")[0] == roof["kernel"]]
        ops.KERNEL_TIMER = None
        ops.SIDE_STREAM = True
        if mine:
            t_alone = sum(max(1e-7, s_.elapsed_time(e_) * 1e-3 - roof["event_pair_overhead_us"] * 1e-6) for s_, e_, _ in mine)
            ach_alone = sum(fl for _, _, fl in mine) / t_alone / 1e12
            # The roofline of a kernel is about the kernel: the primary figures are those of the launches that own the
            # chip (this extra step; profiles/*_serial_kernel_stats.csv is the rocprofv3 summary of the same mode).  The
            # figures of the step as shipped, where the launches share the chip with the weight-gradient stream, move to
            # "as_run" (profiles/*_kernel_stats.csv).
            roof["as_run"] = {k: roof[k] for k in ("achieved", "frac", "executed_mfma_tflops", "avg_launch_us",
                                                   "share_of_step", "symbols")}
            roof["as_run"]["note"] = roof.pop("note")
            per_lab = {}
            for (nm, s_, e_, fl) in [r_ for r_ in recs_alone if r_[0].split("<")[0] == roof["kernel"]]:
                d = per_lab.setdefault(nm, [0.0, 0.0, 0])
                d[0] += max(1e-7, s_.elapsed_time(e_) * 1e-3 - roof["event_pair_overhead_us"] * 1e-6)
                d[1] += fl
                d[2] += 1
            roof.update({"achieved": round(ach_alone, 2), "frac": round(ach_alone / roof["peak"], 4),
                         "executed_mfma_tflops": round(3 * ach_alone, 1) if roof["executed_mfma_tflops"] else None,
                         "avg_launch_us": round(t_alone / len(mine) * 1e6, 1),
                         # this kernel's launches, each alone on the chip, as a share of the timed step (as_run.share_of_step:
                         # the same launches as they ran, stretched by the weight-gradient stream beside them)
                         "share_of_step": round(t_alone / step_s, 3),
                         "frac_as_run": roof["as_run"]["frac"],
                         "symbols": {k: {"launches_per_step": v[2], "avg_launch_us": round(v[0] / v[2] * 1e6, 1),
                                         "tflops": round(v[1] / v[0] / 1e12, 1)} for k, v in per_lab.items()},
                         "measured": "one extra train step with the weight-gradient stream off (MULAN_SIDE_STREAM=0): no "
                                     "other kernel on the chip while a launch of this kernel runs"})
    return state, roof


def multi_rank_report(exp, state, batches, B, steps, warmup, world, dev, elapsed, elapsed_local, barrier):
    """What a --gpus N run must say about itself (the driver's one SCALE run is the only RCCL evidence
    this build gets: ldm/experiment.py:89-95,341).  After the K timed steps, on every rank:
      * replicas_in_sync: max |difference| over the ranks of the flat parameter buffer and of the EMA buffer (element-wise
        MAX minus MIN all-reduce): data parallelism keeps replicas bit-identical -- every rank adds the same all-reduced
        gradient with the same optimizer kernel -- so anything but 0.0 is a bug and the run exits non-zero;
      * ms_per_step of every rank (its own clock around the same timed region);
      * allreduce_exposed_ms: the timed step minus the same step with the collectives left out
        (GradReducer.skip_collectives), i.e. what of the gradient exchange is NOT hidden under the backward pass;
      * n1_equivalent: the throughput one GPU of this run reaches on its per-GPU batch with no exchange at all, and the
        scaling efficiency of the line against N times that (the driver computes its own from the per-N lines)."""
    import torch
    import torch.distributed as dist
    tl = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
    allt = [torch.zeros_like(tl) for _ in range(world)]
    dist.all_gather(allt, tl)
    per_rank = [round(float(x[0]) / steps * 1e3, 3) for x in allt]

    def spread(buf):
        mx, mn = buf.detach().clone(), buf.detach().clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        return float((mx - mn).abs().max())
    d_params, d_ema = spread(state.flat), spread(state.ema)
    rep = {"replicas_in_sync": d_params == 0.0 and d_ema == 0.0, "max_abs_param_difference_over_ranks": d_params,
           "max_abs_ema_difference_over_ranks": d_ema, "checked": "flat parameter buffer and EMA buffer after the timed steps",
           "ms_per_step_by_rank": per_rank, "ms_per_step_min": min(per_rank), "ms_per_step_max": max(per_rank)}
    red = exp.reducer
    captured = bool(exp._graphed is not None and getattr(exp._graphed, "captured_collectives", False))
    if red.enabled and not captured:
        red.skip_collectives = True
        try:
            for i in range(2):
                state, _ = exp.train_step(exp._train_rng, state, batches[i % len(batches)])
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            n = max(2, min(steps, 10))
            for i in range(n):
                state, _ = exp.train_step(exp._train_rng, state, batches[i % len(batches)])
            torch.cuda.synchronize()
            barrier()
            t = torch.tensor([(time.perf_counter() - t0) / n], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        finally:
            red.skip_collectives = False
        no_coll_ms = float(t[0]) * 1e3
        ms = elapsed / steps * 1e3
        rep.update({"ms_per_step_without_collectives": round(no_coll_ms, 3),
                    "allreduce_exposed_ms": round(ms - no_coll_ms, 3),
                    "n1_equivalent": {"images_per_sec_per_gpu": round(B / no_coll_ms * 1e3, 2),
                                      "what": "this run's per-GPU batch with the gradient exchange left out (max over ranks): what "
                                              "one GPU does alone",
                                      "scaling_efficiency": round((B * world / ms) / (world * B / no_coll_ms), 4)},
                    "note": "the steps without collectives ran AFTER the replica check (the replicas drift apart from there on)"})
    else:
        rep.update({"allreduce_exposed_ms": None, "n1_equivalent": None,
                    "note": "collectives captured into the step's graph: cannot be left out of a replay"})
    return rep


def train_workload(a, rank, world, cfg_path, vdm_type, vfe, B, steps, warmup, f32_reference):
    """W untimed + K timed train steps of one configuration (barrier + synchronize on both sides, MAX over ranks), then
    the convolution kernel's roofline steps.  Returns a dict of raw results."""
    import torch
    import torch.distributed as dist
    from mulan_amd import ops
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    config = _test_depth(load_config_file(cfg_path))
    config.vdm_type = vdm_type
    config.data.dataset = "synthetic"
    if vfe:
        config.model.velocity_from_epsilon = True
    config.training.batch_size_train = B * world
    config.training.batch_size_eval = B * world
    config.training.substeps = 1
    exp = Experiment_VDM(config)
    dev = exp.device

    g = torch.Generator().manual_seed(rank)
    nb = min(steps + warmup, 24)                 # distinct resident batches (cycled: 24 x 393 KB)
    batches = [{"images": torch.randint(0, 256, (B, 32, 32, 3), generator=g, dtype=torch.uint8).to(dev),
                "labels": torch.zeros(B, dtype=torch.int32, device=dev),
                "conditioning": torch.zeros(B, dtype=torch.uint8, device=dev)} for _ in range(max(1, nb))]

    def barrier():
        if world > 1:
            dist.barrier()

    state = exp.state
    # One-time set-up outside the W + K steps (the counterpart of a compile step): the first call runs eagerly
    # (kernel attributes, allocator), the second captures the HIP graph that every later call replays.  Without this a
    # run with --warmup 0 or 1 would time the capture.
    prime = 0
    while exp.hip_graph and exp._graphed is None and prime < 2:
        state, _ = exp.train_step(exp._train_rng, state, batches[0])
        prime += 1
    for i in range(warmup):
        state, _ = exp.train_step(exp._train_rng, state, batches[i % len(batches)])
    torch.cuda.synchronize()
    barrier()
    tele = Telemetry(torch.cuda.current_device())
    with tele:
        t0 = time.perf_counter()
        for i in range(steps):
            state, m = exp.train_step(exp._train_rng, state, batches[(warmup + i) % len(batches)])
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    multi = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        multi = multi_rank_report(exp, state, batches, B, steps, warmup, world, dev, elapsed, elapsed_local, barrier)
    res = {"elapsed": elapsed, "multi": multi, "last_bpd": float(m["scalars"]["train_bpd"]), "prime": prime,
           "graph_used": bool(exp.hip_graph and exp._graphed is not None), "E": int(config.model.sm_n_embd),
           "n_layer": int(config.model.sm_n_layer), "conv_mode": ops.CONV_MODE, "f32_mode": None,
           "graph_error": exp.graph_capture_error, "chip": tele.summary(), "elapsed_local": elapsed_local}
    red = exp.reducer
    if world > 1 and red.capture is not None:
        # how the replayed step hands its gradient buckets to the collectives (parallel.GradReducer): per bucket, how long
        # before the end of the graph the collective stream was released in the trial replay
        res["overlap"] = {"handoff": "signal" if red.capture.get("signals") is not None else "event",
                          "buckets": len(red.buckets), "marked": list(red.capture.get("order", [])),
                          "released_ms_before_graph_end": getattr(red, "bucket_leads", None),
                          "trial_replays_ms": getattr(red, "calibration", None)}
    state, res["roof"] = conv_roofline(exp, state, batches[-1], a, rank, world, B, res["E"], elapsed / steps)
    # ---- the same step with the exact-fp32 MFMA convolution kernels (MULAN_CONV_MODE=f32), for reference
    if f32_reference and world == 1 and ops.CONV_MODE != "f32":
        saved = ops.CONV_MODE
        ops.CONV_MODE = "f32"
        exp.hip_graph = False               # (the captured graph holds the split-operand kernels)
        state, _ = exp.train_step(exp._train_rng, state, batches[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nref = max(2, min(steps // 2, 10))
        for i in range(nref):
            state, _ = exp.train_step(exp._train_rng, state, batches[i % len(batches)])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / nref
        ops.CONV_MODE = saved
        res["f32_mode"] = {"value": round(B / dt, 2), "unit": "images/s", "ms_per_step": round(dt * 1e3, 2), "steps": nref,
                           "note": "convolutions on v_mfma_f32_32x32x2_f32 (exact fp32 MFMA) instead of the "
                                   "split-operand kernels"}
    exp._graphed = None
    del exp, state, batches
    _release_device_memory()
    return res


def _release_device_memory():
    """between two workloads: the experiment just dropped may still be held by reference cycles (autograd contexts of its last
    eager step); collect them before the caching allocator gives its blocks back, so that every workload starts from an
    empty pool.  (Config #4 reads 236-237 ms on most boxes and 248-253 ms on some, inside this run and alone alike: that
    spread is the box's, not this function's -- same-box sequences in profiles/r04_config4_box_spread.log.)"""
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()


def _max_over_ranks(seconds, world, dev):
    import torch
    import torch.distributed as dist
    if world > 1:
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t[0])
    return seconds


def dense_eval_workload(images, T, rank=0, world=1):
    """BASELINE configs[4]: eval_bpd --bpd_eval_method=dense on the ImageNet-32 configuration, the test images sharded
    over the ranks by index -- the evaluator itself (mulan_amd.evaluators.eval_bpd_dense_sampling = ldm/notebook_utils.py:
    176-191: every test image a batch of T copies through loss_fn(is_train=False) under one key, image i on rank i % world,
    one all-reduce of (sum bpd, count) at the end).  `images` per rank after one warm-up image per rank; barrier +
    synchronize on both sides, MAX over ranks."""
    import torch
    import torch.distributed as dist
    from mulan_amd import evaluators
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    config = _test_depth(load_config_file(os.path.join(ROOT, "ldm", "configs", "imagenet32.py")))
    config.data.dataset = "synthetic"
    config.vdm_type = "mulan_velocity"
    config.model.velocity_from_epsilon = True
    config.training.batch_size_train = 8 * world
    config.training.batch_size_eval = 8 * world
    exp = Experiment_VDM(config)
    exp.orig_params = exp.state.ema_params               # what Experiment_Colab evaluates (notebook_utils.py:36)
    quiet = open(os.devnull, "w")
    saved = sys.stdout

    def run(n_images):
        sys.stdout = quiet                                # (the evaluator prints its running mean like the reference)
        try:
            return evaluators.eval_bpd_dense_sampling(exp, config, n_timesteps=T, max_images=n_images * world)
        finally:
            sys.stdout = saved
    run(1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    bpd = run(images)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = _max_over_ranks(time.perf_counter() - t0, world, exp.device)
    dt = elapsed / images                                 # seconds per test image on one GPU
    del exp
    _release_device_memory()
    return {"workload": f"eval_bpd dense VLB (evaluators.eval_bpd_dense_sampling), ldm/configs/imagenet32.py (E=256, "
                        f"velocity_from_epsilon), T={T} copies per image, {images} images per rank after one warm-up "
                        f"image, test images sharded over {world} rank(s) by index, one (sum, count) all-reduce, forward "
                        "only, EMA weights",
            "metric": "dense-eval seconds per test image", "value": round(dt, 4), "unit": "s/image",
            "higher_is_better": False, "test_images_per_s_all_ranks": round(images * world / elapsed, 3),
            "forward_images_per_s": round(T * world / dt, 1),
            "model_tflops_per_gpu": round(T / dt * FWD_GFLOP_BY_WIDTH[256] / 1e3, 1), "bpd_random_init": round(float(bpd), 4)}


def sampler_workload(B, T, steps):
    """Experiment_VDM.sample_fn's inner loop (ldm/experiment_vdm.py:80-110) at the flagship
    configuration -- `steps` timed reverse steps of VDM.sample out of a T-step schedule, then generate_x"""
    import torch
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd.rng import PRNGKey
    config = _test_depth(load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py")))
    config.data.dataset = "synthetic"
    from mulan_amd import parallel
    config.training.batch_size_train = B * parallel.world_size()      # (the configuration's batch sizes are global ones:
    config.training.batch_size_eval = B * parallel.world_size()       # B images per rank, as the reference samples)
    exp = Experiment_VDM(config)
    st, model = exp.state, exp.model
    cond = torch.zeros(B, dtype=torch.uint8, device=exp.device)
    key = PRNGKey(0)
    packer = st.param_packer("ema")
    with torch.no_grad():
        if packer is not None:
            packer.refresh()
        emb = model.deterministic_embedding(B, exp.device)
        coeffs = model.sample_coefficients(st.ema_params, emb)
        step = model.reverse_stepper(st.ema_params, B, exp.device, emb, cond, coeffs, T)     # the product path: replayed graph
        z = float(config.model.sigma_prior) * key.normal((B, 3072), exp.device)
        for i in range(3):
            z = step(i, z, key)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + steps):
            z = step(i, z, key)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        x = model.generate_x(st.ema_params, z, coeffs, rng=key.fold_in(T))
        torch.cuda.synchronize()
        if packer is not None:
            packer.invalidate()
    ok = tuple(x.shape) == (B, 32, 32, 3) and bool(torch.isfinite(z).all())
    E = int(config.model.sm_n_embd)
    del exp
    _release_device_memory()
    gf = SCORE_FWD_GFLOP_BY_WIDTH[E]
    return {"workload": f"ancestral sampler (Experiment_VDM.sample_fn loop: VDM.sample + generate_x), cifar10-conditioned "
                        f"(E={E}), batch {B}, {steps} timed reverse steps of a T={T} schedule, EMA weights, 1 GPU",
            "metric": "sampled images per second at T=1000", "value": round(B / (dt * T), 3), "unit": "images/s",
            "hip_graph": type(getattr(step, "__self__", None)).__name__ == "GraphedReverseStep",
            "ms_per_reverse_step": round(dt * 1e3, 3), "reverse_steps_per_s": round(1.0 / dt, 2),
            "image_steps_per_s": round(B / dt, 1), "seconds_per_grid_of_B_images": round(dt * T, 2),
            "model_tflops": round(B / dt * gf / 1e3, 1),
            "model_roofline_frac": round(B / dt * gf / 1e3 / (PEAK_BF16_MFMA_TFLOPS / 3), 4), "finite": ok}


def ode_workload(B):
    """one exact-likelihood solve as eval_bpd --bpd_eval_method=ode runs it (ldm/notebook_utils.py:
    264-373) -- a batch of B test images, truncated-normal dequantisation, Rademacher Hutchinson probe per function
    evaluation, adaptive Dormand-Prince at rtol = atol = 1e-5, one importance sample (n_is = 1)"""
    import torch
    from mulan_amd import evaluators
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    from mulan_amd.rng import PRNGKey
    config = _test_depth(load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py")))
    config.data.dataset = "synthetic"
    from mulan_amd import parallel
    config.training.batch_size_train = B * parallel.world_size()      # (global batch sizes: B images per rank)
    config.training.batch_size_eval = B * parallel.world_size()
    exp = Experiment_VDM(config)
    exp.orig_params = exp.state.ema_params
    like = evaluators.get_ode_likelihood_fn(exp, hutchinson_type="Rademacher", rtol=1e-5, atol=1e-5, dequantization="tn")
    img = torch.randint(0, 256, (B, 32, 32, 3), dtype=torch.uint8, device=exp.device)
    like(PRNGKey(1), img, t_grid=[0.0, 0.01])            # warm-up: one Dormand-Prince step (7 function evaluations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    log_p, log_q, aux, info = like(PRNGKey(2), img)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    offset = evaluators._get_bpd_offset("tn", 1)
    bpd = float((-(log_p) + aux.double()).mean() / (3072 * math.log(2.0)) + offset)
    E = int(config.model.sm_n_embd)
    del exp
    _release_device_memory()
    per = dt / info["nfev"]
    gf = 3 * SCORE_FWD_GFLOP_BY_WIDTH[E] * 2.0 / 3.0       # forward + input-gradient pass (no weight gradients)
    return {"workload": f"exact-likelihood ODE evaluator (evaluators.get_ode_likelihood_fn), cifar10-conditioned (E={E}), "
                        f"batch {B}, dequantization tn, Rademacher probes, adaptive RK45 rtol=atol=1e-5, n_is=1, EMA "
                        "weights at their random initialisation, 1 GPU",
            "metric": "ODE function evaluations per second (batch)", "value": round(1.0 / per, 2), "unit": "nfe/s",
            "nfev": int(info["nfev"]), "rk45_steps": int(info["steps"]), "rk45_rejected": int(info["rejected"]),
            "ms_per_function_evaluation": round(per * 1e3, 3), "image_evaluations_per_s": round(B / per, 1),
            "seconds_per_image_this_solve": round(dt / B, 4),
            "seconds_per_image_at_nfev_300_n_is_20": round(300 * 20 * per / B, 3),
            "model_tflops": round(B / per * gf / 1e3, 1), "bpd_random_init": round(bpd, 4),
            "finite": bool(torch.isfinite(log_p).all())}


def split_precision_probe():
    """The precision claim as observed in THIS process: one 128 -> 128 3x3 convolution launch (8 images,
    standard-normal data, weights x 0.05) through the f16x3 kernel and through the exact-fp32 MFMA kernel, each against
    torch's float64 convolution of the same operands; error = max |y - y64| / max(sum |x| |w|) -- the unit of DESIGN.md's
    table (K = 1152 products per output)."""
    import torch
    import torch.nn.functional as F
    from mulan_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    Bp, C, N = 8, 128, 128
    x = torch.randn(Bp, 1024, C, device="cuda", generator=g)
    w = torch.randn(3, 3, C, N, device="cuda", generator=g) * 0.05
    x64 = x.double().view(Bp, 32, 32, C).permute(0, 3, 1, 2)
    w64 = w.double().permute(3, 2, 0, 1)
    y64 = F.conv2d(x64, w64, padding=1).permute(0, 2, 3, 1).reshape(Bp, 1024, N)
    scale = float(F.conv2d(x64.abs(), w64.abs(), padding=1).max())
    out, saved = {}, ops.CONV_MODE
    try:
        for mode, key in (("f16x3", "split_error_vs_fp64"), ("f32", "f32_error_vs_fp64")):
            ops.CONV_MODE = mode
            y = ops.conv3x3_raw(x, w, None, None, None)
            y = y[0] if isinstance(y, tuple) else y
            out[key] = float((y.double() - y64).abs().max()) / scale
    finally:
        ops.CONV_MODE = saved
    out["what"] = ("max |y - y_fp64| / max sum|x||w| of one 128 -> 128 3x3 convolution (8 images, K = 1152), measured in this "
                   "process: the f16x3 split kernel and the exact-fp32 MFMA kernel against torch's float64 convolution")
    return out


def plain_vdm_workload(steps=10, batch=2):
    """BASELINE configs[0] on the HIP path (the GPU twin of cpu_baseline.config0_plain_vdm): model_vdm.VDM with the
    learnable monotone schedule (--config.vdm_type=vdm --config.model.gamma_type=learnable_nnet), batch 2, 10 train steps
    through Experiment.train_step (eager: ~1100 launches per step, host-bound at this batch)."""
    import torch
    from mulan_amd.config import load_config_file
    from mulan_amd.experiment import Experiment_VDM
    config = _test_depth(load_config_file(os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py")))
    config.vdm_type = "vdm"
    config.model.gamma_type = "learnable_nnet"
    config.data.dataset = "synthetic"
    config.training.batch_size_train = batch
    config.training.batch_size_eval = batch
    config.training.substeps = 1
    exp = Experiment_VDM(config)
    g = torch.Generator().manual_seed(0)
    mk = lambda: {"images": torch.randint(0, 256, (batch, 32, 32, 3), generator=g, dtype=torch.uint8).cuda(),
                  "labels": torch.zeros(batch, dtype=torch.int32).cuda(),
                  "conditioning": torch.zeros(batch, dtype=torch.uint8).cuda()}
    batches = [mk() for _ in range(steps)]
    state = exp.state
    for i in range(2):
        state, _ = exp.train_step(exp._train_rng, state, batches[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        state, m = exp.train_step(exp._train_rng, state, batches[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"workload": f"BASELINE configs[0]: model_vdm.VDM (gamma_type learnable_nnet), CIFAR-10 32x32, batch {batch}, "
                       f"{steps} train steps through Experiment.train_step on the HIP path (eager step; 2 untimed steps first)",
           "value": round(batch * steps / dt, 2), "unit": "images/s", "ms_per_step": round(dt / steps * 1e3, 2),
           "steps": steps, "global_batch": batch, "last_train_bpd": round(float(m["scalars"]["train_bpd"]), 4),
           "cpu_twin": "cpu_baseline.config0_plain_vdm"}
    del exp, state, batches
    _release_device_memory()
    return res


def main():
    a = parse()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(a.config, a.vdm_type, a.cpu_batch, a.cpu_steps)), flush=True)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    import torch
    import torch.distributed as dist
    from mulan_amd import ops, parallel

    rank, world, local = parallel.init_distributed()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    backend = dist.get_backend() if world > 1 else None
    rccl_ranks = 0
    if world > 1:
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        if backend == "nccl":
            # every rank of the RCCL communicator answers one all-reduce: the count must be --gpus
            one = torch.ones(1, device="cuda")
            dist.all_reduce(one)
            rccl_ranks = int(one.item())
            assert rccl_ranks == a.gpus, f"RCCL communicator spans {rccl_ranks} ranks, --gpus {a.gpus}"
    strong = a.global_batch > 0
    if strong and a.global_batch % world != 0:
        raise SystemExit(f"--global-batch {a.global_batch} is not divisible by {world} ranks")
    B = a.global_batch // world if strong else a.per_gpu_batch
    head = train_workload(a, rank, world, a.config, a.vdm_type, a.vfe, B, a.steps, a.warmup, not a.no_f32_mode)
    elapsed, roof = head["elapsed"], head["roof"]

    # ---- the other BASELINE configurations, driver-timed with the headline ("configs"), at every N:
    #   "3": configs[2] as worded -- MuLAN-velocity CIFAR-10 at GLOBAL batch 512 (strong scaling: 512 / N images per GPU;
    #        at N = 1 also "3_at_64_per_gpu", the per-GPU size of the 8-GPU configuration, as in rounds 1-3);
    #   "4": configs[3] -- MuLAN-velocity ImageNet-32, velocity_from_epsilon, 128 images per GPU (global 128 N: the
    #        configuration's global batch of 1024 at N = 8);
    #   "5": configs[4] -- the dense evaluator itself, test images sharded over the ranks;
    #   "sampler", "ode": SURVEY 8(f) ranks 3 and 2 (rank 0's GPU only).
    also = a.also_configs if a.also_configs is not None else True
    extra = None
    out_of_sync = ["headline"] if (head["multi"] is not None and not head["multi"]["replicas_in_sync"]) else []
    if also:
        extra = {}
        cif = os.path.join(ROOT, "ldm", "configs", "cifar10-conditioned.py")
        inet = os.path.join(ROOT, "ldm", "configs", "imagenet32.py")
        g3, b4, T5, n5, bs, bo = (32, 8, 64, 1, 8, 4) if a.configs_small else (512, 128, 1000, 2, 64, 64)
        if a.configs_small:
            b3, l3 = g3 // world, f"(--configs-small) MuLAN-velocity CIFAR-10, global batch {g3}"
        elif 512 % world != 0:
            b3, l3 = 64, "BASELINE configs[2]: MuLAN-velocity CIFAR-10 at its 8-GPU per-GPU size of 64 images"
        else:
            b3, l3 = 512 // world, (f"BASELINE configs[2]: MuLAN-velocity CIFAR-10, GLOBAL batch 512 (strong scaling: "
                                    f"{512 // world} images per GPU)")
        table = {"3": (cif, "mulan_velocity", False, b3, l3),
                 "4": (inet, "mulan_velocity", True, b4, "BASELINE configs[3]: MuLAN-velocity ImageNet-32 (E=256), "
                                                         f"velocity_from_epsilon, {b4} per GPU = global batch {b4 * world} "
                                                         "(the configuration's 1024 at 8 GPUs x 128)")}
        if world == 1 and not a.configs_small:
            table["3_at_64_per_gpu"] = (cif, "mulan_velocity", False, 64, "BASELINE configs[2] at its 8-GPU per-GPU size: "
                                        "MuLAN-velocity CIFAR-10, 64 images per GPU (512 / 8)")
        for key, (cfgp, vt, vfe, bsz, label) in table.items():
            # (8 untimed replays: after the set-up of a new workload the chip has idled for seconds and its clock ramps over
            # the first steps -- with 3 the `chip` record of config #4 showed 1.3-1.5 GHz in the first timed step and the entry
            # read 522-536 images/s instead of 550-568, profiles/r06_box_spread.log)
            r = train_workload(a, rank, world, cfgp, vt, vfe, bsz, a.also_steps, ALSO_WARMUP, False)
            ips = bsz * world * a.also_steps / r["elapsed"]
            gf = FWD_GFLOP_BY_WIDTH[r["E"]]
            rr = r["roof"] or {}
            extra[key] = {"workload": label + f"; full train step, {world} GPU(s) x {bsz}", "value": round(ips, 2),
                          "unit": "images/s", "global_batch": bsz * world, "ms_per_step": round(r["elapsed"] / a.also_steps * 1e3, 2),
                          "steps": a.also_steps, "warmup": ALSO_WARMUP, "hip_graph": r["graph_used"],
                          "model_tflops_per_gpu": round(ips / world * 3 * gf / 1e3, 2),
                          "conv_kernel": {k: rr.get(k) for k in ("kernel", "achieved", "peak", "frac", "avg_launch_us",
                                                                 "launches_per_step", "measured")},
                          "conv_kernel_as_run_frac": (rr.get("as_run") or {}).get("frac"),
                          "last_train_bpd": round(r["last_bpd"], 4), "chip": r["chip"], "multi_gpu": r["multi"]}
            if r["multi"] is not None and not r["multi"]["replicas_in_sync"]:
                out_of_sync.append(key)
        extra["5"] = dense_eval_workload(n5, T5, rank, world)
        extra["5"]["workload"] = "BASELINE configs[4]: " + extra["5"]["workload"]
        if world > 1:
            dist.barrier()
        if rank == 0:
            if world == 1:      # (a one-device configuration; on several ranks its train step would wait for the others' collectives)
                extra["1"] = plain_vdm_workload()
            extra["sampler"] = sampler_workload(bs, 1000, 3 if a.configs_small else 20)
            extra["ode"] = ode_workload(bo)
            if world == 1 and not a.configs_small:
                # the reference samples / evaluates with the per-device batch: 16 images per GPU on 8 GPUs
                # (ldm/experiment.py:96-102) -- short convolution tiles + replayed function evaluations (round 5)
                extra["sampler_16"] = sampler_workload(16, 1000, 20)
                extra["ode_16"] = ode_workload(16)
    if rank != 0:
        if world > 1:
            dist.barrier()
        if out_of_sync:
            raise SystemExit(3)
        return
    precision = split_precision_probe() if (ops.CONV_MODE == "f16x3" and not a.configs_small) else None
    cpu = None
    if not a.no_cpu_baseline and world == 1:
        # host leg in a child process (no GPU touched there) with a hard wall-clock bound
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--config", a.config, "--vdm-type",
               a.vdm_type, "--cpu-batch", str(a.cpu_batch), "--cpu-steps", str(a.cpu_steps)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=a.cpu_timeout,
                               env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            cpu = json.loads(line[-1]) if line else {"value": None, "error": (r.stderr or "no output")[-300:]}
        except subprocess.TimeoutExpired:
            cpu = {"value": None, "error": f"cpu baseline exceeded {a.cpu_timeout}s"}
    ms = elapsed / a.steps * 1e3
    value = B * world * a.steps / elapsed
    fwd_gflop = FWD_GFLOP_BY_WIDTH.get(head["E"], FWD_GFLOP_PER_IMAGE)
    out = {
        "metric": "train images/sec", "value": round(value, 2), "unit": "images/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "setup_steps": head["prime"], "ms_per_step": round(ms, 2),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": {"f16x3": "f32 (f16x3 split: 3 fp16 MFMA passes, fp32 accumulate)",
                  "bf16x6": "f32 (bf16x6 split: 6 bf16 MFMA passes, fp32 accumulate)"}.get(ops.CONV_MODE, "f32"),
        "data": "synthetic",
        "conv_mode": ops.CONV_MODE + {
            "bf16x6": ": fp32 operands split into 3 bf16 pieces, 6 bf16 MFMA passes, fp32 accumulate",
            "f16x3": ": fp32 operands scaled by a power of two and split into 2 fp16 pieces, 3 fp16 MFMA passes, fp32 "
                     "accumulate (error vs fp64 equal to the fp32 MFMA kernel's; the float32 matmul precision the "
                     "reference requests, ldm/main.py:39)",
            "f32": ": exact fp32 MFMA"}.get(ops.CONV_MODE, ""),
        "config": {"workload": f"MuLAN ({a.vdm_type}) config ldm/configs/{os.path.basename(a.config)} "
                               f"(E={head['E']}, {head['n_layer']}+2+{head['n_layer'] + 1} "
                               f"ResnetBlocks{', velocity_from_epsilon' if a.vfe else ''}), full train step "
                               f"(fwd+bwd+all-reduce+AdamW/EMA), " +
                               (f"global batch {B * world} fixed (strong scaling), {B}/GPU" if strong else
                                f"batch {B}/GPU (weak scaling)"),
                   "global_batch": B * world, "parallelism": f"dp{world}", "image": "32x32x3 uint8"},
        "collective": ({"backend": backend, "rccl_ranks": rccl_ranks,
                        "schedule": ("replayed HIP graph, bucket all-reduces behind their signal words (opt-in)"
                                     if head.get("overlap") else
                                     ("replayed HIP graph, collectives behind it" if head["graph_used"] else
                                      "eager step, bucketed all-reduce overlapping the backward pass")),
                        "replay_overlap": head.get("overlap"),
                        "note": "torch.distributed backend 'nccl' is RCCL on ROCm; bucketed gradient all-reduce on a "
                                "side stream, overlapped with the (replayed) backward pass"} if world > 1 else None),
        "model_tflops_per_gpu": round(value / world * 3 * fwd_gflop / 1e3, 2),
        # the whole step against the arithmetic ceiling of its scheme (173.3 GFLOP per image fwd + bwd): the figure the
        # headline moves with; roofline.frac below is the dominant kernel alone on the chip, roofline.frac_as_run the same
        # launches inside the step as timed
        "step_roofline_frac": round(value / world * 3 * fwd_gflop / 1e3 /
                                    {"bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "f16x3": PEAK_BF16_MFMA_TFLOPS / 3}.get(
                                        ops.CONV_MODE, PEAK_F32_MFMA_TFLOPS), 4),
        "model_roofline_frac": round(value / world * 3 * fwd_gflop / 1e3 /
                                     {"bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "f16x3": PEAK_BF16_MFMA_TFLOPS / 3}.get(
                                         ops.CONV_MODE, PEAK_F32_MFMA_TFLOPS), 4),
        "last_train_bpd": round(head["last_bpd"], 4),
        "hip_graph": head["graph_used"], "hip_graph_error": head["graph_error"],
        "oracle_pin": "unpinned: the reference has no golden vectors, JAX / Flax are not installable here and no "
                      "released checkpoint is in the image (tools/verify_checkpoint.py pins it in one command)",
        "roofline": roof, "f32_mfma_mode": head["f32_mode"], "precision": precision, "chip": head["chip"],
        "multi_gpu": head["multi"], "cpu_baseline": cpu, "configs": extra,
    }
    if out_of_sync:
        out["replicas_out_of_sync"] = out_of_sync
    if TEST_DEPTH > 0:        # (--small-depth: a test run; none of its numbers is the configuration's)
        out["test_depth"] = TEST_DEPTH
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if out_of_sync:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
