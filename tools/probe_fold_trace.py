#!/usr/bin/env python3
"""Developer probe: cycle stamps (s_memtime, workgroup 0 / thread 0) of the fused half-step kernel's attention instance
(k_half_fused<1, 2, false, true>: front-end phases, then the MLP layers) on the config2_attn bench batch.
Needs a -DGNF_FOLD_TRACE build:  tools/build_variants.sh foldtrace "-DGNF_FOLD_TRACE"
  python tools/probe_fold_trace.py [variant] [train]      ("train": the training forward's stash instance, default flags)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else "foldtrace"
train = len(sys.argv) > 2 and sys.argv[2] == "train"
os.environ["GNF_LIB_PATH"] = os.path.join(ROOT, "graph-normalizing-flows_amd", "variants", f"libgnf_{variant}.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import make_product_grevnet  # noqa: E402
from gnf_amd.graphs import data_dicts_to_graphs_tuple  # noqa: E402

dev = torch.device("cuda:0")
bench.WORKLOAD = bench.WORKLOADS["default_flags_train" if train else "config2_attn"]
bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
bench.HP.update(bench.WORKLOAD["hp"])
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
if train:
    from gnf_amd.train import GRevNetTrainer
    tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
    for _ in range(3):
        tr.loss_and_grads(graph)
else:
    for _ in range(5):
        net(graph, inverse=True)
torch.cuda.synchronize()
raw = C.CDLL(os.environ["GNF_LIB_PATH"])
buf = (C.c_ulonglong * 64)()
assert raw.gnf_debug_read_fold_trace(buf) == 0
t = list(buf)
names = {0: "start", 1: "P0 own rows staged", 2: "P1 k done", 3: "c0 chunk top", 5: "c0 window rows staged", 6: "c0 q|v projected",
         7: "c0 attention", 8: "c1 chunk top", 10: "c1 window rows staged", 11: "c1 q|v projected", 12: "c1 attention",
         28: "normalised + barrier", 30: "MLP prefetches issued", 31: "output projection done", 29: "h0 rows in LDS", 32: "biases in LDS", 33: "barrier"}
K = bench.HP["K"]
for j in range(K):
    names[34 + j] = f"layer {j} done"
names[48] = "end"
last = t[0]
for k in sorted(names, key=lambda q: t[q]):
    if t[k] == 0:
        continue
    print(f"{names[k]:26s} {t[k] - t[0]:8d}  (+{t[k] - last})")
    last = t[k]
