"""world_size-2 / 4 / 8 gloo tests (CPU) of the N>1 path: the bucketed gradient reducer over the flat buffer equals the
big-batch gradient, scalar metrics are averaged like lax.pmean, and the sharded dense-eval reduction is
partition invariant, the sampler's all-gather concatenates in rank order.  (RCCL is the same torch.distributed API with backend 'nccl'.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from mulan_amd import parallel
    from mulan_amd.train_state import TrainState, tree_leaves_in_layout
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    tree = {"score_model": {"a": {"kernel": torch.randn(40, 30, generator=g), "bias": torch.randn(30, generator=g)},
                            "b": {"kernel": torch.randn(30, 7, generator=g), "bias": torch.randn(7, generator=g)}},
            "gamma": {"c": {"kernel": torch.randn(7, 5, generator=g)}}}
    st = TrainState.create(apply_fn=None, variables={"params": tree}, device="cpu")
    leaves = [(leaf, off, leaf.numel()) for (path, off, shape), (_, leaf) in
              zip(st.layout, tree_leaves_in_layout(st.params, st.layout))]
    red = parallel.GradReducer(st.grad, leaves, bucket_bytes=4 * 400)   # several buckets
    assert len(red.buckets) > 1
    # the buckets tile the flat gradient buffer: contiguous, no gap, no overlap, every leaf inside exactly one
    spans = sorted((lo, hi) for lo, hi, _ in red.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == st.grad.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for _, off, n in leaves:
        assert sum(1 for lo, hi in spans if lo <= off and off + n <= hi) == 1

    def loss(params, x):
        h = torch.tanh(x @ params["score_model"]["a"]["kernel"] + params["score_model"]["a"]["bias"])
        h = torch.tanh(h @ params["score_model"]["b"]["kernel"] + params["score_model"]["b"]["bias"])
        return (h @ params["gamma"]["c"]["kernel"]).pow(2).mean()

    xg = torch.randn(8, 40, generator=torch.Generator().manual_seed(1))
    per = 8 // world                          # the global batch of 8 split over the ranks (512 // world at full size)
    shard = xg[rank * per:(rank + 1) * per]
    ok = True
    for _ in range(2):                       # two steps: hooks / buckets re-arm correctly
        st.zero_grad()
        red.prepare()
        loss(st.params, shard).backward()
        st.collect_grads()
        red.finish()
        mean_grad = st.grad / world          # the optimizer kernel applies 1/world as grad_scale
        ref = TrainState.create(apply_fn=None, variables={"params": tree}, device="cpu")
        ref.zero_grad()
        loss(ref.params, xg).backward()      # big-batch gradient on one process
        ref.collect_grads()
        ok = ok and torch.allclose(mean_grad, ref.grad, atol=1e-6)
    # the replayed-step entry points on a reducer without device streams (CPU tensors): begin_capture / end_capture are
    # no-ops and allreduce_captured reduces every bucket behind the whole backward pass -- the same sum
    st.zero_grad()
    red.paused = True
    red.begin_capture()
    loss(st.params, shard).backward()
    red.end_capture()
    st.collect_grads()
    red.paused = False
    ok = ok and red.capture is None
    red.allreduce_captured()
    ok = ok and torch.allclose(st.grad / world, ref.grad, atol=1e-6) and sorted(red.ready_order) == list(range(len(red.buckets)))
    m = parallel.allreduce_mean_scalars({"bpd": torch.tensor(float(rank + 1)), "var": 2.0 * (rank + 1)}, "cpu")
    ok = ok and abs(float(m["bpd"]) - (world + 1) / 2) < 1e-6 and abs(float(m["var"]) - (world + 1.0)) < 1e-6
    # sharded evaluator reduction: per-image values split by index, (sum, count) all-reduced
    from mulan_amd.evaluators import _reduce_mean
    vals = [float(i) for i in range(10)]
    mine = vals[rank::world]
    mean, n = _reduce_mean(sum(mine), len(mine), "cpu")
    ok = ok and n == 10 and abs(mean - 4.5) < 1e-12
    # sample grids: every rank ends up with the samples of all ranks in rank order (jax.lax.all_gather of the reference)
    mine_u8 = torch.full((3, 2, 2, 3), rank + 1, dtype=torch.uint8)
    allg = parallel.all_gather_tensor(mine_u8)
    ok = ok and allg.shape == (3 * world, 2, 2, 3) and all(int(allg[3 * r].max()) == r + 1 for r in range(world))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(420)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_gradient_reduction_over_n_ranks_matches_big_batch(world):
    """the reference runs on however many local devices there are (ldm/experiment.py:86-102, dataset.py:256-266): the
    reducer, the scalar mean, the evaluator's (sum, count) reduction with images i mod world, and the sample all-gather
    at 2, 4 and 8 ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=360) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def test_single_process_reducer_is_a_noop():
    from mulan_amd import parallel
    flat = torch.zeros(16)
    t = flat[:8].view(2, 4).detach().requires_grad_(True)
    red = parallel.GradReducer(flat, [(t, 0, 8)])
    red.prepare()
    red.finish()
    assert not red.enabled and parallel.world_size() == 1
