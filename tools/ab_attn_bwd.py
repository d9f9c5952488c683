"""Developer tool: the attention backward's edge passes on the default-flags training step (config-2 batch, dm_self_attn +
batch norm), one process, interleaved arms: receiver + sender pass in one launch (default on sparse batches) against the
two-launch form (attn_bwd_split=1), for each row-tile size (attn_bwd_rows)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch
import bench
from helpers import make_product_grevnet
from gnf_amd import _abi
from gnf_amd.graphs import data_dicts_to_graphs_tuple
from gnf_amd.train import GRevNetTrainer

bench.WORKLOAD = bench.WORKLOADS["default_flags_train"]
bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
bench.HP.update(bench.WORKLOAD["hp"])
dev = torch.device("cuda:0")
dicts, _, _ = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
ARMS = [("one launch, auto", 0, 0), ("two launches, auto", 1, 0), ("one launch, 32 / 32", 0, 32), ("one launch, 32 / 64", 0, 3264),
        ("one launch, 64 / 64", 0, 64)]
for _ in range(60):
    tr.step(graph)
torch.cuda.synchronize()
res = {a[0]: [] for a in ARMS}
for rnd in range(6):
    for name, split, rows in ARMS:
        _abi.set_option("attn_bwd_split", split)
        _abi.set_option("attn_bwd_rows", rows)
        for _ in range(5):
            tr.step(graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            tr.step(graph)
        torch.cuda.synchronize()
        res[name].append(1e3 * (time.perf_counter() - t0) / 40)
_abi.set_option("attn_bwd_split", 0)
_abi.set_option("attn_bwd_rows", 0)
for name, v in res.items():
    v.sort()
    print(f"{name:22s} median {v[len(v) // 2]:.3f} ms/step  (min {v[0]:.3f}, max {v[-1]:.3f})")
