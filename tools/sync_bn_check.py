"""Cross-rank batch-norm moments (GRevNet(sync_batch_norm=True), GnfFlow.bn_allreduce): two ranks on ONE GPU (gloo
rendezvous on 127.0.0.1), each holding half of the graphs of a batch, must reproduce what a single process computes
for the whole batch with the reference's single-device semantics: z, log-det, batch moments, and - after the
gradient all-reduce - every gradient.  Prints 'sync-bn-ok' from rank 0 on success.  Used by tests/test_train_gpu.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.distributed as dist
import torch.multiprocessing as mp


def make_graphs(rng, n_graphs):
    from gnf_amd.datasets import senders_receivers
    n_node = rng.integers(5, 12, size=n_graphs).astype(np.int32)
    return n_node


def sub_batch(n_node, ids):
    from gnf_amd.datasets import senders_receivers
    nn = n_node[ids]
    s, r, ne = senders_receivers(nn)
    return nn, ne, s, r


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gnf_oracle as O
    from helpers import graph_from_arrays, make_product_grevnet
    from gnf_amd.train import GRevNetTrainer
    dev = "cuda:0"
    D, L, K, T = 8, 32, 3, 2
    rng = np.random.default_rng(3)
    n_node = rng.integers(5, 12, size=10).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(n_node)])
    n = int(off[-1])
    x = (rng.standard_normal((n, D)) * 1.3 + 0.4).astype(np.float32)
    p = O.make_grevnet_params(8, D // 2, L, K, T, final_scale=0.3)
    p["bn"] = O.make_bn_params(9, D // 2, T)
    hp = dict(D=D, latent=L, K=K, T=T, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu", weight_sharing=False)

    # --- the whole batch in one process: the reference's semantics (per-process, no collective)
    all_ids = np.arange(10)
    nn, ne, s, r = sub_batch(n_node, all_ids)
    full = make_product_grevnet(hp, p)
    tr_full = GRevNetTrainer(full)
    out_full = tr_full.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, dev))
    z_full, _ = full(graph_from_arrays(nn, ne, s, r, x, dev), inverse=True)
    g_full = tr_full.grad.detach().cpu().numpy().copy()
    mom_full = [[(b.batch_mean.cpu().numpy().copy(), b.batch_variance.cpu().numpy().copy()) for b in half] for half in full.bns]

    # --- this rank's shard with cross-rank moments
    ids = all_ids[rank::world]
    rows = np.concatenate([np.arange(off[g], off[g + 1]) for g in ids])
    nn, ne, s, r = sub_batch(n_node, ids)
    net = make_product_grevnet(hp, p)
    net.sync_batch_norm = True
    tr = GRevNetTrainer(net)
    graph = graph_from_arrays(nn, ne, s, r, x[rows], dev)
    out = tr.loss_and_grads(graph)
    tr.all_reduce_gradients()
    z, _ = net(graph, inverse=True)
    torch.cuda.synchronize()
    loss = torch.tensor([float(out["total_loss"])], dtype=torch.float64)
    dist.all_reduce(loss)
    errs = {
        "z": float(np.abs(z.nodes.cpu().numpy() - z_full.nodes.cpu().numpy()[rows]).max()),
        "loss": abs(float(loss[0]) - float(out_full["total_loss"])) / n,
        "grad": float(np.abs(tr.grad.cpu().numpy() - g_full).max() / np.abs(g_full).max()),
        "moments": max(float(np.abs(b.batch_mean.cpu().numpy() - mom_full[h][i][0]).max() +
                             np.abs(b.batch_variance.cpu().numpy() - mom_full[h][i][1]).max())
                       for h, half in enumerate(net.bns) for i, b in enumerate(half)),
    }
    # per-shard moments (sync off) must NOT reproduce the whole-batch result: the check above is not vacuous
    local = make_product_grevnet(hp, p)
    z_loc, _ = local(graph, inverse=True)
    errs["z_without_sync"] = float(np.abs(z_loc.nodes.cpu().numpy() - z_full.nodes.cpu().numpy()[rows]).max())
    ok = errs["z"] <= 2e-5 and errs["loss"] <= 1e-5 and errs["grad"] <= 2e-5 and errs["moments"] <= 1e-5 \
        and errs["z_without_sync"] > 1e-3
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    print(f"rank {rank}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items()), flush=True)
    if rank == 0 and float(flag[0]) == 1.0:
        print("sync-bn-ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
