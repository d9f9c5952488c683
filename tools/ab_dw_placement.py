"""Developer tool: where the weight-gradient GEMMs of a half-step run - inside the next half-step's backward launch (default),
on the auxiliary stream beside the walk (dw_unmerged=1), or on the walk's own stream right behind the half-step
(dw_unmerged=1, no auxiliary stream) - ms per training step, interleaved arms, both training workloads."""
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

dev = torch.device("cuda:0")
base_hp = dict(bench.HP)
for wl in ("config2_train", "default_flags_train"):
    bench.WORKLOAD = bench.WORKLOADS[wl]
    bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
    bench.HP.clear(); bench.HP.update(base_hp); bench.HP.update(bench.WORKLOAD["hp"])
    dicts, _, _ = bench.make_batch(1, 0)
    graph = data_dicts_to_graphs_tuple(dicts, dev)
    net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
    tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
    ARMS = [("merged launch", 0, True, 0), ("auxiliary stream", 1, True, 0), ("same stream, behind the half-step", 1, False, 0),
            ("same stream, 256 wide units", 1, False, 256), ("same stream, 512 wide units", 1, False, 512)]
    for _ in range(60):
        tr.step(graph)
    torch.cuda.synchronize()
    res = {a[0]: [] for a in ARMS}
    for rnd in range(5):
        for name, unmerged, aux, units in ARMS:
            _abi.set_option("dw_unmerged", unmerged)
            _abi.set_option("dw_wide_units", units)
            tr.overlap_weight_grads = aux
            for _ in range(5):
                tr.step(graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                tr.step(graph)
            torch.cuda.synchronize()
            res[name].append(1e3 * (time.perf_counter() - t0) / 40)
    _abi.set_option("dw_unmerged", 0)
    _abi.set_option("dw_wide_units", 0)
    for name, v in res.items():
        v.sort()
        print(f"{wl:20s} {name:36s} median {v[len(v) // 2]:.3f} ms/step")
