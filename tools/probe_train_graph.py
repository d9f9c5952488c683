#!/usr/bin/env python3
"""Developer probe: is the training step host-bound?  Times trainer.step() eagerly and as a replayed HIP graph
(torch.cuda.graph capture of loss_and_grads + apply_gradients; the Adam step size is frozen in the capture, fine for timing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "config2_train"
bench.WORKLOAD = bench.WORKLOADS[wl]; bench.HP.update(bench.WORKLOAD["hp"])
from helpers import make_product_grevnet
from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple
from gnf_amd.train import GRevNetTrainer
dev = torch.device("cuda:0")
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
for _ in range(5):
    tr.step(graph)
torch.cuda.synchronize()
def timeit(fn, k=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / k
print("eager step        %.4f ms" % timeit(lambda: tr.step(graph)))
t0 = time.perf_counter()
for _ in range(50): tr.step(graph)
print("eager host-only   %.4f ms per step (launch calls return)" % (1e3 * (time.perf_counter() - t0) / 50))
torch.cuda.synchronize()
cg = torch.cuda.CUDAGraph()
with torch.cuda.graph(cg):
    tr.step(graph)
print("graph replay step %.4f ms" % timeit(cg.replay))
