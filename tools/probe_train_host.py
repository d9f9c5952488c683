"""Developer tool: where does the HOST spend a training step?  cProfile over K steps of the config-2 trainer (device work
is asynchronous, so what shows up is launch / ctypes / torch overhead), plus enqueue vs completion time per step."""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch
import bench
from helpers import make_product_grevnet
from gnf_amd.graphs import data_dicts_to_graphs_tuple
from gnf_amd.train import GRevNetTrainer

bench.WORKLOAD = bench.WORKLOADS["config2_train"]
bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
dev = torch.device("cuda:0")
dicts, n_global, e_global = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
for _ in range(10):
    tr.step(graph)
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for _ in range(K):
    tr.step(graph)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{K} steps: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, complete {1e3 * (t2 - t0) / K:.3f} ms/step")
enq, tot = [], []
for _ in range(30):           # one step at a time on an empty queue: the host's own launch cost, no back-pressure
    torch.cuda.synchronize()
    a = time.perf_counter()
    tr.step(graph)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    enq.append(1e3 * (b - a)), tot.append(1e3 * (c - a))
enq.sort(), tot.sort()
print(f"single step on an empty queue: host returns after {enq[len(enq) // 2]:.3f} ms (median), done after {tot[len(tot) // 2]:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    tr.step(graph)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
