"""Developer probe: where a k-step of the generic GEMM tile goes (s_memtime ticks of workgroup (0,0,0) / thread 0) on the
wide_fc forward (2718 x 2048 x 2048 layers through the layered path).
Build first: tools/build_variants.sh dw_trace "-DGNF_DW_TRACE"; run with GNF_LIB_PATH=.../variants/libgnf_dw_trace.so"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from helpers import make_product_grevnet
from gnf_amd import _abi
from gnf_amd.graphs import data_dicts_to_graphs_tuple
bench.WORKLOAD = bench.WORKLOADS["wide_fc"]; bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
bench.HP.update(bench.WORKLOAD["hp"])
dev = torch.device("cuda:0")
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
for _ in range(2):
    net(graph, inverse=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
lib = _abi.lib()
lib.gnf_debug_read_dw_trace(buf, 1)
net(graph, inverse=True)
torch.cuda.synchronize()
lib.gnf_debug_read_dw_trace(buf, 0)
t = list(buf)
steps = max(1, t[7])
print(f"{steps} k-steps traced (all generic GEMM launches of one forward)")
for i, nm in ((0, "fetch issue"), (2, "compute (LDS frags + MFMA)"), (3, "stash (wait + ds_write)"), (4, "barrier")):
    print(f"{nm:28s} {t[i] / steps:8.0f} ticks per step")
print(f"{'total':28s} {sum(t[:5]) / steps:8.0f} ticks per step  (MFMA floor at two waves per SIMD: 2048 cycles)")
