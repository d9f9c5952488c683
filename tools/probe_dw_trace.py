"""Developer probe: where a step of k_gemm_dw_wide goes (s_memtime ticks, 100 MHz, workgroup 0 / thread 0).
Build first:  tools/build_variants.sh dw_trace "-DGNF_DW_TRACE";  run with
GNF_LIB_PATH=graph-normalizing-flows_amd/variants/libgnf_dw_trace.so GNF_DW_WIDE_CHUNKS=2 GNF_TRAIN_NO_OVERLAP=1"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from helpers import make_product_grevnet
from gnf_amd import _abi
from gnf_amd.graphs import data_dicts_to_graphs_tuple
from gnf_amd.train import GRevNetTrainer
bench.WORKLOAD = bench.WORKLOADS["config2_train"]; bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
bench.HP.update(bench.WORKLOAD["hp"])
dev = torch.device("cuda:0")
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
for _ in range(3):
    tr.step(graph)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
lib = _abi.lib()
lib.gnf_debug_read_dw_trace(buf, 1)
tr.step(graph)
torch.cuda.synchronize()
lib.gnf_debug_read_dw_trace(buf, 0)
t = list(buf)
steps = max(1, t[7])
names = ["fetch issue", "colsum", "compute (MFMA)", "stash (wait + ds_write)", "barrier"]
print(f"{steps} steps traced")
for i, nm in enumerate(names):
    print(f"{nm:26s} {t[i] / 100.0 / steps:8.3f} us per step")
print(f"{'total':26s} {sum(t[:5]) / 100.0 / steps:8.3f} us per step")
launches = 16
print(f"prologue {t[5] / launches:.0f} ticks per launch, epilogue {t[6] / launches:.0f} ticks per launch, loop {sum(t[:5]) / launches:.0f} ticks per launch")
