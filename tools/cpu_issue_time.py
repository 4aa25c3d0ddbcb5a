"""How long does the host need to issue one train step (no synchronisation inside the loop)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mulan_amd.config import load_config_file
from mulan_amd.experiment import Experiment_VDM
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config = load_config_file(os.path.join(root, "ldm", "configs", "cifar10-conditioned.py"))
config.vdm_type = "mulan_epsilon"; config.data.dataset = "synthetic"
config.training.batch_size_train = 128; config.training.batch_size_eval = 128; config.training.substeps = 1
exp = Experiment_VDM(config)
B = 128
batch = {"images": torch.randint(0, 256, (B, 32, 32, 3), dtype=torch.uint8).cuda(),
         "labels": torch.zeros(B, dtype=torch.int32).cuda(), "conditioning": torch.zeros(B, dtype=torch.uint8).cuda()}
state = exp.state
for _ in range(3):
    state, _ = exp.train_step(exp._train_rng, state, batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 6
for _ in range(n):
    state, m = exp.train_step(exp._train_rng, state, batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / n:.1f} ms/step, until GPU done {1e3 * (t2 - t0) / n:.1f} ms/step")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        state, m = exp.train_step(exp._train_rng, state, batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
