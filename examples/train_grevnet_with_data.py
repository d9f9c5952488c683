#!/usr/bin/env python3
"""The data-backed trainer's loop (/root/reference/train_grevnet_with_data.py:145-271, 336-416, 520-545) on the
MI355X kernels: node-embedding chunks on disk -> batches of complete graphs (transform_example) -> GRevNet with that
driver's default GNN (dm_attn: one head, kq = v = 64, C = 64, kq_dim_division; :40-46, 303-310) around its wide relu
MLPs (latent 2048 x 3 layers, D = 200, 10 coupling layers, batch norm) -> Adam (beta2 0.999, constant lr); then the
sampling pipeline z ~ N(0, I) -> grevnet(., inverse=False) -> pred_adj(scaled_hacky_sigmoid_l2) -> threshold 0.5.

There is no trained encoder here (run_gnn.py is out of scope), so --make_chunks writes embedding chunks whose
embeddings are synthetic (two well-separated clusters per graph, so that the decoder finds structure); point
--train_data_dir at real chunks written by generate_grevnet_training_data.py to use those instead.

    python examples/train_grevnet_with_data.py --make_chunks --num_train_iters 200 --clip_gradient_by_norm

(With these synthetic, nearly degenerate embeddings the un-clipped defaults diverge after ~50 iterations - the
gradients themselves are verified against the oracle at this width, tools/wide_grad_check.py - which is what the
driver's clip_gradient_by_norm / clip_gradient_by_value flags are for.)
"""
import argparse
import os
import random
import sys
import tempfile
import time
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnf_amd import datasets as D, gnn                                # noqa: E402
from gnf_amd.flow import pred_adj, sample                             # noqa: E402
from gnf_amd.train import GRevNetTrainer                              # noqa: E402


def make_chunks(path, files, graphs_per_file, dim, rng):
    for k in range(files):
        n_node = rng.integers(8, 20, size=graphs_per_file).astype(np.int32)
        rows = []
        for n in n_node:
            centres = rng.standard_normal((2, dim)) * 0.6
            lab = rng.integers(0, 2, size=n)
            rows.append(centres[lab] + 0.1 * rng.standard_normal((n, dim)))
        D.write_embedding_chunk(os.path.join(path, f"grevnet_train_{k}.pkl"), np.concatenate(rows), n_node)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train_data_dir", default=None)
    ap.add_argument("--make_chunks", action="store_true")
    ap.add_argument("--node_embedding_dim", type=int, default=200)
    ap.add_argument("--latent_dim", type=int, default=2048)
    ap.add_argument("--num_layers", type=int, default=3)
    ap.add_argument("--num_coupling_layers", type=int, default=10)
    ap.add_argument("--bias_init_stddev", type=float, default=0.3)
    ap.add_argument("--no_batch_norm", action="store_true")
    ap.add_argument("--weight_sharing", action="store_true")
    # the reference's defaults (train_grevnet_with_data.py:40-46): dm_attn, one head, kq = v = 64, C = 64
    # ("avg_then_mlp" is not a choice of the reference's ATTN_MAP; kept here as the message-passing alternative)
    ap.add_argument("--attn_type", default="dm_attn", choices=["dm_attn", "avg_then_mlp"])
    ap.add_argument("--use_layer_norm", action="store_true")
    ap.add_argument("--attn_kq_dim", type=int, default=64)
    ap.add_argument("--attn_v_dim", type=int, default=64)
    ap.add_argument("--attn_num_heads", type=int, default=1)
    ap.add_argument("--attn_concat_heads_output_dim", type=int, default=64)
    ap.add_argument("--train_batch_size", type=int, default=32)
    ap.add_argument("--train_epochs", type=int, default=20)
    ap.add_argument("--num_train_iters", type=int, default=60)
    ap.add_argument("--log_every_n_steps", type=int, default=10)
    ap.add_argument("--sample_size", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-4)          # lr_type 'constant' (train_grevnet_with_data.py:75-78)
    ap.add_argument("--adam_beta1", type=float, default=0.9)
    ap.add_argument("--adam_beta2", type=float, default=0.999)  # (:86; run_grevnet.py's default is 0.9)
    ap.add_argument("--adam_epsilon", type=float, default=1e-8)
    ap.add_argument("--clip_gradient_by_value", action="store_true")
    ap.add_argument("--clip_gradient_by_norm", action="store_true")
    ap.add_argument("--clip_gradient_norm", type=float, default=10.0)
    ap.add_argument("--random_seed", type=int, default=12345)
    F = ap.parse_args()
    random.seed(F.random_seed)
    np.random.seed(F.random_seed)
    torch.manual_seed(F.random_seed)
    gnn.set_random_seed(F.random_seed)
    rng = np.random.default_rng(F.random_seed)
    dev = torch.device("cuda", 0)
    tmp = None
    if F.train_data_dir is None:
        if not F.make_chunks:
            raise SystemExit("give --train_data_dir or --make_chunks")
        tmp = tempfile.TemporaryDirectory()
        F.train_data_dir = tmp.name
        make_chunks(F.train_data_dir, 3, 10 * F.train_batch_size, F.node_embedding_dim, rng)
    # sort_files: the reference consumes the chunks in os.listdir order, which differs from one temporary directory
    # to the next (and with it the whole loss curve); sorted here so that a run of this demo is reproducible
    data = D.GrevnetDatasetFixed(F.train_data_dir, F.train_batch_size, F.train_epochs, sort_files=True)

    make_mlp_fn = partial(gnn.make_mlp_model, F.latent_dim, F.node_embedding_dim / 2, F.num_layers,
                          activation=gnn.relu, l2_regularizer_weight=0.000001, bias_init_stddev=F.bias_init_stddev)
    make_gnn_fn = {
        "avg_then_mlp": partial(gnn.avg_then_mlp_gnn, make_mlp_fn, 1.0),
        "dm_attn": partial(gnn.dm_self_attn_gnn, kq_dim=F.attn_kq_dim, v_dim=F.attn_v_dim, make_mlp_fn=make_mlp_fn,
                           num_heads=F.attn_num_heads, concat_heads_output_dim=F.attn_concat_heads_output_dim,
                           kq_dim_division=True, layer_norm=F.use_layer_norm),
    }[F.attn_type]
    grevnet = gnn.GRevNet(make_gnn_fn, F.num_coupling_layers, F.node_embedding_dim,
                          use_batch_norm=not F.no_batch_norm, weight_sharing=F.weight_sharing)
    trainer = GRevNetTrainer(grevnet, lr=F.lr, adam_beta1=F.adam_beta1, adam_beta2=F.adam_beta2, adam_epsilon=F.adam_epsilon,
                             use_lr_decay=False, clip_gradient_by_value=F.clip_gradient_by_value,
                             clip_gradient_by_norm=F.clip_gradient_by_norm, clip_gradient_norm=F.clip_gradient_norm)
    t0 = time.perf_counter()
    for iteration in range(F.num_train_iters + 1):
        z, n_node = data.train_batch()
        graph = D.transform_example(z, n_node, dev)
        v = trainer.step(graph)
        if iteration % F.log_every_n_steps == 0:
            print(f"iteration {iteration:5d} ({time.perf_counter() - t0:6.1f} s): total_loss {float(v['total_loss']):12.3f} "
                  f"per_node_loss {float(v['loss_per_node']):9.4f} log_det_jacobian {float(v['log_det_jacobian']):12.3f} "
                  f"batch nodes {graph.nodes.shape[0]}")
            if not np.isfinite(float(v["total_loss"])):
                raise SystemExit("loss is not finite")
    # ---- sampling pipeline (train_grevnet_with_data.py:397-416, 526-540) ---------------------------------------
    n_node = np.asarray(random.sample(list(data.n_node) * F.sample_size, F.sample_size), np.int32)
    shell = D.transform_example(np.zeros((int(n_node.sum()), F.node_embedding_dim), np.float32), n_node, dev)
    out = sample(grevnet, shell)
    blocks = pred_adj(out["grevnet_top"])
    for i, b in enumerate(blocks):
        adj = (b > 0.5)
        print(f"sampled graph {i}: {int(n_node[i])} nodes, {int(adj.sum().item()) // 2} edges, "
              f"mean sample log-prob {float(out['sample_log_prob'][sum(n_node[:i]):sum(n_node[:i + 1])].mean()):.2f}")
    if tmp is not None:
        tmp.cleanup()


if __name__ == "__main__":
    main()
