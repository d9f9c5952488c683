#!/usr/bin/env python3
"""Developer probe: s_memtime stamps of workgroup 0 of k_attn_front (needs tools/build_variants.sh attn_trace "-DGNF_ATTN_TRACE")."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GNF_LIB_PATH"] = os.path.join(ROOT, "graph-normalizing-flows_amd", "variants", "libgnf_attn_trace.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
bench.WORKLOAD = bench.WORKLOADS["config2_attn"]; bench.HP.update(bench.WORKLOAD["hp"])
from helpers import make_product_grevnet
from gnf_amd.flow import forward_shard_sums
from gnf_amd.graphs import data_dicts_to_graphs_tuple
dev = torch.device("cuda:0")
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
for _ in range(3):
    forward_shard_sums(net, graph)
torch.cuda.synchronize()
raw = C.CDLL(os.environ["GNF_LIB_PATH"])
out = (C.c_ulonglong * 32)()
assert raw.gnf_debug_read_front_trace(out) == 0
t = np.array(list(out), dtype=np.int64)
names = {0: "start", 1: "P0 staged", 2: "P1 k done", 27: "chunks done", 28: "normalised", 29: "end"}
names = {0: "start", 1: "P0 staged", 2: "P1 k done", 28: "agg in LDS", 29: "end"}
for c in range(5):
    for j, nm in enumerate(["chunk top", "col+erow", "gathered", "q|v proj", "attention"]):
        names[3 + 5 * c + j] = f"c{c} {nm}"
prev = t[0]
for i in sorted(names):
    if t[i] >= t[0]:
        print(f"{names[i]:16s} {t[i] - t[0]:8d}  (+{t[i] - prev})")
        prev = t[i]
