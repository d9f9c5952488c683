"""Data-parallel training through the product path: two ranks on ONE GPU (gloo on 127.0.0.1), each holding the shard
shard_graph_ids gives it, run GRevNetTrainer.step(graph, all_reduce=True) three times; a single process runs the
same three steps on the whole batch.  total_loss is a sum over nodes (run_grevnet.py:295), so the all-reduced
gradient IS the batch gradient: checked step by step AT THE SAME PARAMETERS (the single-process trainer is put on the
data-parallel trainer's parameters before every step), and the parameters after a step must then agree to rounding.
(Comparing two free-running trajectories is not a test of the product: parameters 8e-8 apart can put one node's
pre-activation on the other side of a leaky-relu kink, the gradient then differs by that node's contribution and Adam
turns it into half a learning-rate step - seen once the MLP-row stash changed the rounding of the backward pass.)
Prints 'dp-train-ok' from rank 0.  Used by tests/test_multirank_gpu.py."""
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

    full = make_product_grevnet(HP, p)
    tr_full = GRevNetTrainer(full, lr=1e-3, use_lr_decay=False)      # one process, whole batch
    g_full = batch(np.arange(24))
    mine = shard_graph_ids(nn_all, ne_all, world)[rank]
    net = make_product_grevnet(HP, p)
    tr = GRevNetTrainer(net, lr=1e-3, use_lr_decay=False)            # this rank's shard, gradients all-reduced every step
    g_mine = batch(mine)
    losses, losses_full, err_grad, err_theta = [], [], 0.0, 0.0
    theta0 = None
    for k in range(STEPS):
        if k > 0:                                                    # same parameters on both sides (see the header)
            tr_full.theta.copy_(tr.theta)
            full.repack()
        out_f = tr_full.loss_and_grads(g_full)
        out = tr.loss_and_grads(g_mine)
        if theta0 is None:
            theta0 = tr_full.theta.detach().cpu().numpy().copy()
        tr.all_reduce_gradients()
        gf = tr_full.grad.detach().cpu().numpy()
        err_grad = max(err_grad, float(np.abs(tr.grad.detach().cpu().numpy() - gf).max()) / float(np.abs(gf).max()))
        tr_full.apply_gradients()
        tr.apply_gradients()
        err_theta = max(err_theta, float((tr.theta - tr_full.theta).abs().max()))
        t = torch.tensor([float(out["total_loss"])], dtype=torch.float64)
        dist.all_reduce(t)
        losses.append(float(t[0]))
        losses_full.append(float(out_f["total_loss"]))
    torch.cuda.synchronize()
    moved = float(np.abs(tr_full.theta.detach().cpu().numpy() - theta0).max())
    err_loss = max(abs(a - b) / n for a, b in zip(losses, losses_full))
    ok = err_grad <= 2e-6 and err_theta <= 2e-6 and err_loss <= 1e-5 and moved > 1e-4 and losses_full[-1] < losses_full[0]
    print(f"rank {rank}: all-reduced gradient vs whole-batch gradient {err_grad:.2e} (relative), |theta - theta_1proc| after a step "
          f"{err_theta:.2e} (parameters moved {moved:.2e}), loss/node err {err_loss:.2e}, "
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
