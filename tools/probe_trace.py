#!/usr/bin/env python3
"""Developer probe: per-wave s_memtime stamps of workgroup 0 of the fused half-step kernel
(needs a -DGNF_TRACE build: tools/build_variants.sh trace "-DGNF_TRACE")."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GNF_LIB_PATH"] = os.path.join(ROOT, "graph-normalizing-flows_amd", "variants", "libgnf_" + (sys.argv[1] if len(sys.argv) > 1 else "trace") + ".so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import make_product_grevnet  # noqa: E402
from gnf_amd import _abi  # noqa: E402
from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple  # noqa: E402

dev = torch.device("cuda:0")
HP = bench.HP
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(HP, bench.make_params(bench.WEIGHT_SEED, HP, bench.FINAL_SCALE))
lib = _abi.lib()
raw = C.CDLL(os.environ["GNF_LIB_PATH"])
h = HP["D"] // 2
flow = net._flow(h, dev)
csr = csr_of(graph)
ws_bytes = lib.gnf_workspace_bytes(n, HP["D"], C.byref(flow))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
buf = graph.nodes.clone()
st = _abi.stream_ptr(dev)
for rep in range(3):
    for q in (0, 1):
        lib.gnf_coupling_half_f32(C.byref(csr.desc), C.byref(flow.s_nets[q]), C.byref(flow.t_nets[q]),
                                  C.byref(flow.gnn), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 4 * h),
                                  buf.stride(0), h, 0, None, _abi.ptr(ws), ws_bytes, st)
torch.cuda.synchronize()
out = (C.c_ulonglong * 128)()
assert raw.gnf_debug_read_trace(out) == 0
t = np.array(list(out), dtype=np.int64).reshape(8, 16)
t0 = t[:, 0].min()
names = ["start", "agg done", "bar0", "L0 done", "bar", "L1 done", "bar", "L2 done", "bar", "L3 done", "bar", "L4 done",
         "bar", "-", "-", "end"]
print("s_memtime ticks (100 MHz constant clock on gfx9 => 10 ns per tick) relative to first wave start")
print("slot            " + " ".join(f"w{w:<7d}" for w in range(8)))
for sl in range(16):
    if names[sl] == "-":
        continue
    print(f"{names[sl]:14s} " + " ".join(f"{int(t[w, sl] - t0):8d}" for w in range(8)))

st = (C.c_ulonglong * 320)()
if hasattr(raw, "gnf_debug_read_stages") and raw.gnf_debug_read_stages(st) == 0:
    g = np.array(list(st), dtype=np.int64).reshape(8, 40)
    base = t[:, 4]  # barrier before layer 1
    print("layer-1 stage stamps relative to the layer's opening barrier (slot 0 = chunk entry, 2+kg = after stage kg, 38 = MFMAs done)")
    for sl in [0] + list(range(2, 18)) + [38]:
        print(f"{sl:3d} " + " ".join(f"{int(g[w, sl] - base[w]):8d}" for w in range(8)))

    print("prologue stamps relative to kernel start (0 prefetch issued, 1 table filled, 2 bias copied, 3 rowptr staged+barrier, 4 col staged+barrier; then 'agg done' above)")
    for sl in range(5):
        print(f"P{sl}  " + " ".join(f"{int(g[w, 30 + sl] - t[w, 0]):8d}" for w in range(8)))
