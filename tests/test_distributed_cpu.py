"""N > 1 path on CPU: world_size-2 gloo.  Each rank takes its shard of the batch (whole graphs, greedy
balance), produces the shard's [logdet, sum z^2, num_nodes] (here with the CPU oracle standing in
for the device kernels - the product has no CPU compute path), the ONE collective of the path
(all-reduce of 3 x fp64) runs over gloo, and every rank must assemble the same batch log-prob as a
single-process run over the whole batch (run_grevnet.py:290-302)."""
import os
import socket

import numpy as np
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


def _worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gnf_oracle as O
        from gnf_amd.flow import log_prob_from_sums
        from gnf_amd.sharding import all_reduce_shard_sums, shard_graph_ids
        d = np.load(os.path.join(ROOT, "data", "community_medium.npz"))
        rng = np.random.default_rng(12345)                      # identical on every rank
        ids = rng.choice(168, size=10, replace=True)
        nn_all, ne_all = d["n_node"][ids], d["n_edge"][ids]
        dim, t = 8, 2
        x_all = rng.standard_normal((int(nn_all.sum()), dim)).astype(np.float32)
        p = O.make_grevnet_params(4, dim // 2, 16, 3, t, final_scale=0.5)
        noff = np.concatenate([[0], np.cumsum(nn_all)])
        mine = shard_graph_ids(nn_all, ne_all, world)[rank]
        rows = np.concatenate([np.arange(noff[i], noff[i + 1]) for i in mine])
        nn, ne, s, r = O.batch_graphs(d["n_node"], d["n_edge"], d["senders"], d["receivers"], ids[mine])
        res = O.Fp64Dense(s, r, int(nn.sum())).log_prob(x_all[rows], p, t)
        sums = torch.tensor([res["log_det_jacobian"], float((res["z"] ** 2).sum()), float(nn.sum())],
                            dtype=torch.float64)
        # the pipelined form bench.py uses: two batches' reductions in flight, read one step later
        from gnf_amd.sharding import all_reduce_shard_sums_async
        pair = [sums.clone(), sums.clone() * 2.0]
        works = [all_reduce_shard_sums_async(t_) for t_ in pair]
        all_reduce_shard_sums(sums)                             # the single collective
        for w_ in works:
            w_.wait()
        assert torch.equal(pair[0], sums) and torch.equal(pair[1], sums * 2.0)
        out = log_prob_from_sums(sums.tolist(), dim)
        ret[rank] = (out["log_prob_xs_per_node"], out["num_nodes"], len(mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_log_prob_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    # single-process reference over the whole batch
    from oracle import gnf_oracle as O
    d = np.load(os.path.join(ROOT, "data", "community_medium.npz"))
    rng = np.random.default_rng(12345)
    ids = rng.choice(168, size=10, replace=True)
    nn, ne, s, r = O.batch_graphs(d["n_node"], d["n_edge"], d["senders"], d["receivers"], ids)
    x = rng.standard_normal((int(nn.sum()), 8)).astype(np.float32)
    p = O.make_grevnet_params(4, 4, 16, 3, 2, final_scale=0.5)
    ref = O.Fp64Dense(s, r, int(nn.sum())).log_prob(x, p, 2)
    for rank in range(world):
        lp, n, cnt = ret[rank]
        assert n == float(nn.sum())
        assert abs(lp - ref["log_prob_xs_per_node"]) < 1e-9
    assert ret[0][2] + ret[1][2] == 10


def test_all_reduce_is_identity_without_process_group():
    from gnf_amd.sharding import all_reduce_shard_sums, assemble_from_sums
    s = torch.tensor([1.0, 2.0, 4.0], dtype=torch.float64)
    out = all_reduce_shard_sums(s.clone())
    assert torch.equal(out, s)
    from gnf_amd.sharding import all_reduce_shard_sums_async
    t = s.clone()
    assert all_reduce_shard_sums_async(t).wait() and torch.equal(t, s)
    asm = assemble_from_sums(s)
    assert float(asm["log_prob_xs"]) == 3.0 and float(asm["log_prob_xs_per_node"]) == 0.75


def _grad_worker(rank, world, port, ret):
    """Data-parallel training: total_loss is a sum over nodes, so the batch gradient is the SUM of the shard
    gradients (one flat all-reduce, GRevNetTrainer.all_reduce_gradients).  The shard gradients come from the
    oracle here (the product has no CPU compute path)."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gnf_oracle as O
        from gnf_amd.sharding import shard_graph_ids
        from gnf_amd.train import GRevNetTrainer
        d = np.load(os.path.join(ROOT, "data", "community_medium.npz"))
        rng = np.random.default_rng(4321)
        ids = rng.choice(168, size=6, replace=True)
        nn_all, ne_all = d["n_node"][ids], d["n_edge"][ids]
        dim, t = 4, 1
        x_all = rng.standard_normal((int(nn_all.sum()), dim))
        p = O.make_grevnet_params(9, dim // 2, 8, 2, t, final_scale=0.5)
        noff = np.concatenate([[0], np.cumsum(nn_all)])
        mine = shard_graph_ids(nn_all, ne_all, world)[rank]
        rows = np.concatenate([np.arange(noff[i], noff[i + 1]) for i in mine])
        nn, ne, s, r = O.batch_graphs(d["n_node"], d["n_edge"], d["senders"], d["receivers"], ids[mine])
        res = O.loss_and_grads(s, r, int(nn.sum()), x_all[rows], p, t)
        flat = np.concatenate([a.ravel() for kind in ("s", "t") for half in res["grads"][kind] for net in half
                               for wb in net for a in wb])
        tr = GRevNetTrainer(None)
        tr.grad = torch.tensor(flat)                            # the arena the all-reduce runs over
        tr.all_reduce_gradients()
        ret[rank] = tr.grad.numpy().copy()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_gradient_all_reduce_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grad_worker, args=(world, port, ret), nprocs=world, join=True)
    from oracle import gnf_oracle as O
    d = np.load(os.path.join(ROOT, "data", "community_medium.npz"))
    rng = np.random.default_rng(4321)
    ids = rng.choice(168, size=6, replace=True)
    nn, ne, s, r = O.batch_graphs(d["n_node"], d["n_edge"], d["senders"], d["receivers"], ids)
    x = rng.standard_normal((int(nn.sum()), 4))
    p = O.make_grevnet_params(9, 2, 8, 2, 1, final_scale=0.5)
    res = O.loss_and_grads(s, r, int(nn.sum()), x, p, 1)
    flat = np.concatenate([a.ravel() for kind in ("s", "t") for half in res["grads"][kind] for net in half
                           for wb in net for a in wb])
    for rank in range(world):
        np.testing.assert_allclose(ret[rank], flat, atol=1e-9)
