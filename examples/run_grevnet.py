#!/usr/bin/env python3
"""The reference's GRevNet driver loop (/root/reference/run_grevnet.py:269-311, 340-377, 440-447) on the MI355X
kernels, flag for flag where the flag concerns the hot path: build the flow from `--make_gnn_fn`, train it on a
synthetic 2-D dataset with Adam, log the scalars of values_map, then sample.  Defaults are the reference's
(dm_self_attn GNN, use_batch_norm, 12 coupling layers, moons_100, batch 32, lr 1e-4) except --num_train_iters.

    python examples/run_grevnet.py --dataset moons_100 --num_train_iters 300 --log_every_n_steps 50
"""
import argparse
import os
import random
import sys
import time
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnf_amd import gnn                                              # noqa: E402
from gnf_amd.flow import sample                                      # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from driver_utils import get_learning_rate                           # noqa: E402
from gnf_amd.grevnet_synthetic_data import DATASETS_MAP              # noqa: E402
from gnf_amd.train import GRevNetTrainer                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--node_embedding_dim", type=int, default=2)
    ap.add_argument("--num_coupling_layers", type=int, default=12)
    ap.add_argument("--weight_sharing", action="store_true")
    ap.add_argument("--make_gnn_fn", default="dm_self_attn",
                    choices=["dm_self_attn", "avg_then_mlp", "avg_concat_then_mlp", "sum_concat_then_mlp"])
    ap.add_argument("--gnn_num_layers", type=int, default=5)
    ap.add_argument("--gnn_latent_dim", type=int, default=256)
    ap.add_argument("--gnn_bias_init_stddev", type=float, default=0.1)
    ap.add_argument("--gnn_avg_then_mlp_epsilon", type=float, default=1.0)
    ap.add_argument("--attn_kq_dim", type=int, default=10)
    ap.add_argument("--attn_v_dim", type=int, default=10)
    ap.add_argument("--attn_num_heads", type=int, default=8)
    ap.add_argument("--attn_concat_heads_output_dim", type=int, default=80)
    ap.add_argument("--no_attn_concat", action="store_true")
    ap.add_argument("--attn_residual", action="store_true")
    ap.add_argument("--attn_layer_norm", action="store_true")
    ap.add_argument("--no_batch_norm", action="store_true")
    ap.add_argument("--dataset", default="moons_100", choices=sorted(DATASETS_MAP))
    ap.add_argument("--train_batch_size", type=int, default=32)
    ap.add_argument("--num_train_iters", type=int, default=300)
    ap.add_argument("--log_every_n_steps", type=int, default=50)
    ap.add_argument("--random_seed", type=int, default=12345)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no_lr_decay", action="store_true")
    ap.add_argument("--lr_decay_steps", type=int, default=1000)
    ap.add_argument("--lr_decay_rate", type=float, default=0.96)
    ap.add_argument("--use_lr_schedule", action="store_true")
    ap.add_argument("--lr_schedule_ramp_up", type=int, default=1000)
    ap.add_argument("--lr_schedule_hold", type=int, default=2000)
    ap.add_argument("--adam_beta1", type=float, default=0.9)
    ap.add_argument("--adam_beta2", type=float, default=0.9)
    ap.add_argument("--adam_epsilon", type=float, default=1e-8)
    ap.add_argument("--clip_gradient_by_value", action="store_true")
    ap.add_argument("--clip_gradient_by_norm", action="store_true")
    ap.add_argument("--clip_gradient_norm", type=float, default=10.0)
    F = ap.parse_args()

    random.seed(F.random_seed)
    np.random.seed(F.random_seed)
    torch.manual_seed(F.random_seed)
    gnn.set_random_seed(F.random_seed)
    dev = torch.device("cuda", 0)
    dataset = DATASETS_MAP[F.dataset]
    h = F.node_embedding_dim / 2

    def mlp(act):                                    # run_grevnet.py:154-180 / 199-211
        return partial(gnn.make_mlp_model, F.gnn_latent_dim, h, F.gnn_num_layers, act, 0.1, F.gnn_bias_init_stddev)

    make_gnn_fn = {
        "dm_self_attn": partial(gnn.dm_self_attn_gnn, kq_dim=F.attn_kq_dim, v_dim=F.attn_v_dim, make_mlp_fn=mlp(gnn.relu),
                                num_heads=F.attn_num_heads, concat_heads_output_dim=F.attn_concat_heads_output_dim,
                                concat=not F.no_attn_concat, residual=F.attn_residual, layer_norm=F.attn_layer_norm),
        "avg_then_mlp": partial(gnn.avg_then_mlp_gnn, mlp(gnn.leaky_relu), F.gnn_avg_then_mlp_epsilon),
        "avg_concat_then_mlp": partial(gnn.avg_concat_then_mlp_gnn, mlp(gnn.leaky_relu)),
        "sum_concat_then_mlp": partial(gnn.sum_concat_then_mlp_gnn, mlp(gnn.leaky_relu)),
    }[F.make_gnn_fn]
    grevnet = gnn.GRevNet(make_gnn_fn, F.num_coupling_layers, F.node_embedding_dim,
                          use_batch_norm=not F.no_batch_norm, weight_sharing=F.weight_sharing)
    trainer = GRevNetTrainer(grevnet, lr=F.lr, adam_beta1=F.adam_beta1, adam_beta2=F.adam_beta2,
                             adam_epsilon=F.adam_epsilon, use_lr_decay=not F.no_lr_decay,
                             lr_decay_steps=F.lr_decay_steps, lr_decay_rate=F.lr_decay_rate,
                             clip_gradient_by_value=F.clip_gradient_by_value,
                             clip_gradient_by_norm=F.clip_gradient_by_norm, clip_gradient_norm=F.clip_gradient_norm)
    t0 = time.perf_counter()
    for iteration in range(F.num_train_iters + 1):
        graph = dataset.get_next_batch(F.train_batch_size, dev)
        lr = None
        if F.use_lr_schedule:   # --use_lr_schedule (utils.py:93-105 through run_grevnet.py:444)
            lr = get_learning_rate(iteration, F.lr, F.lr_schedule_ramp_up, F.lr_schedule_hold)
        v = trainer.step(graph, learning_rate=lr)
        if iteration % F.log_every_n_steps == 0:
            z = v["z_graph"].nodes
            print("*" * 50)
            print(f"iteration num: {iteration}   ({time.perf_counter() - t0:.1f} s)")
            print(f"total_loss: {float(v['total_loss']):.4f}")
            print(f"loss per node: {float(v['loss_per_node']):.5f}")
            print(f"log det jacobian: {float(v['log_det_jacobian']):.4f}")
            print(f"original mean {graph.nodes.mean(0).cpu().numpy()} std dev {graph.nodes.std(0).cpu().numpy()}")
            print(f"transformed mean {z.mean(0).cpu().numpy()} std dev {z.std(0).cpu().numpy()}")
            print(f"device memory: {torch.cuda.memory_allocated(dev) / 2**20:.1f} MiB allocated, "
                  f"{torch.cuda.max_memory_allocated(dev) / 2**20:.1f} MiB peak")
            if not np.isfinite(float(v["total_loss"])):
                raise SystemExit("loss is not finite")
    out = sample(grevnet, dataset.get_next_batch(F.train_batch_size, dev))     # run_grevnet.py:304-311
    x = out["grevnet_top_nodes"]
    print("*" * 50)
    print(f"samples: mean {x.mean(0).cpu().numpy()} std dev {x.std(0).cpu().numpy()} finite {bool(torch.isfinite(x).all())}")


if __name__ == "__main__":
    main()
