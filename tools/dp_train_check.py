"""Data-parallel training through the product path: two ranks on ONE GPU (gloo on 127.0.0.1), each holding the shard
shard_graph_ids gives it, run GRevNetTrainer.step(graph, all_reduce=True) three times; a single process runs the
same three steps on the whole batch.  total_loss is a sum over nodes (run_grevnet.py:295), so the all-reduced
gradient IS the batch gradient and the parameters must follow the same trajectory.  Prints 'dp-train-ok' from rank 0.
Used by tests/test_multirank_gpu.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp

HP = dict(D=16, latent=64, K=3, T=3, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu", weight_sharing=False)
STEPS = 3


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gnf_oracle as O
    from helpers import graph_from_arrays, make_product_grevnet
    from gnf_amd.sharding import all_reduce_shard_sums, shard_graph_ids
    from gnf_amd.train import GRevNetTrainer
    dev = "cuda:0"
    d = np.load(os.path.join(ROOT, "data", "community_medium.npz"))
    rng = np.random.default_rng(11)
    ids = rng.choice(168, size=24, replace=True)
    nn_all, ne_all = d["n_node"][ids], d["n_edge"][ids]
    off = np.concatenate([[0], np.cumsum(nn_all)])
    n = int(off[-1])
    x = rng.standard_normal((n, HP["D"])).astype(np.float32)
    p = O.make_grevnet_params(5, HP["D"] // 2, HP["latent"], HP["K"], HP["T"], final_scale=0.3)

    def batch(sel):
        nn, ne, s, r = O.batch_graphs(d["n_node"], d["n_edge"], d["senders"], d["receivers"], ids[sel])
        rows = np.concatenate([np.arange(off[g], off[g + 1]) for g in sel])
        return graph_from_arrays(nn, ne, s, r, x[rows], dev)

    # one process, whole batch
    full = make_product_grevnet(HP, p)
    tr_full = GRevNetTrainer(full, lr=1e-3, use_lr_decay=False)
    g_full = batch(np.arange(24))
    losses_full = [float(tr_full.step(g_full)["total_loss"]) for _ in range(STEPS)]
    theta_full = tr_full.theta.detach().cpu().numpy().copy()
    # this rank's shard, gradients all-reduced every step
    mine = shard_graph_ids(nn_all, ne_all, world)[rank]
    net = make_product_grevnet(HP, p)
    tr = GRevNetTrainer(net, lr=1e-3, use_lr_decay=False)
    g_mine = batch(mine)
    losses = []
    for _ in range(STEPS):
        out = tr.step(g_mine, all_reduce=True)
        t = torch.tensor([float(out["total_loss"])], dtype=torch.float64)
        dist.all_reduce(t)
        losses.append(float(t[0]))
    torch.cuda.synchronize()
    theta = tr.theta.detach().cpu().numpy()
    err_theta = float(np.abs(theta - theta_full).max())
    moved = float(np.abs(theta_full - np.concatenate([a.ravel() for k in "st" for half in p[k] for netp in half for wb in netp for a in wb])).max())
    err_loss = max(abs(a - b) / n for a, b in zip(losses, losses_full))
    ok = err_theta <= 2e-6 and err_loss <= 1e-5 and moved > 1e-4 and losses_full[-1] < losses_full[0]
    print(f"rank {rank}: |theta - theta_1proc| {err_theta:.2e} (parameters moved {moved:.2e}), loss/node err {err_loss:.2e}, "
          f"loss {losses_full[0] / n:.4f} -> {losses_full[-1] / n:.4f}", flush=True)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0 and float(flag[0]) == 1.0:
        print("dp-train-ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
