#!/usr/bin/env python3
"""One trainer step of a bench workload with and without the auxiliary stream for the weight-gradient GEMMs
(GRevNetTrainer.overlap_weight_grads): ms per step, on the GPU box.
    python tools/probe_train_overlap.py wide_fc_train"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "wide_fc_train"
    bench.WORKLOAD = bench.WORKLOADS[wl]
    bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
    bench.HP.update(bench.WORKLOAD["hp"])
    from gnf_amd.factories import make_product_grevnet
    from gnf_amd.graphs import data_dicts_to_graphs_tuple
    from gnf_amd.train import GRevNetTrainer
    dev = torch.device("cuda", 0)
    dicts, n, e = bench.make_batch(1, 0)
    graph = data_dicts_to_graphs_tuple(dicts, dev)
    params = bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE)
    for overlap in (True, False):
        net = make_product_grevnet(bench.HP, params)
        tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
        tr.overlap_weight_grads = overlap
        for _ in range(3):
            tr.step(graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step(graph)
        torch.cuda.synchronize()
        print(f"{wl} overlap_weight_grads={overlap}: {1e2 * (time.perf_counter() - t0):.3f} ms per step", flush=True)


if __name__ == "__main__":
    main()
