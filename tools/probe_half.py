#!/usr/bin/env python3
"""Developer probe: time the fused half-step kernel (a) cycling through the 16 different nets of a
flow (cold weights: every launch streams 1.7 MB never seen by the XCD L2s) and (b) re-running ONE net
(weights L2-resident).  One event pair around a burst of launches, so host overhead is amortised."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import make_product_grevnet  # noqa: E402
from gnf_amd import _abi  # noqa: E402
from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple  # noqa: E402

dev = torch.device("cuda:0")
HP = bench.HP
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bench.GRAPHS_PER_GPU = G
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
print(f"graphs={G} nodes={n} tiles16={(n + 15) // 16}", flush=True)
net = make_product_grevnet(HP, bench.make_params(bench.WEIGHT_SEED, HP, bench.FINAL_SCALE))
lib = _abi.lib()
h = HP["D"] // 2
flow = net._flow(h, dev)
csr = csr_of(graph)
ws_bytes = lib.gnf_workspace_bytes(n, HP["D"], C.byref(flow))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
buf = graph.nodes.clone()
st = _abi.stream_ptr(dev)


def burst(qs, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        for q in qs:
            half = q // HP["T"]
            cond = buf.data_ptr() + (0 if half == 0 else 4 * h)
            upd = buf.data_ptr() + (4 * h if half == 0 else 0)
            # INVERSE then FORWARD alternate would keep values bounded; values do not affect timing
            lib.gnf_coupling_half_f32(C.byref(csr.desc), C.byref(flow.s_nets[q]), C.byref(flow.t_nets[q]),
                                      C.byref(flow.gnn), C.c_void_p(cond), C.c_void_p(upd), buf.stride(0), h, 0,
                                      None, _abi.ptr(ws), ws_bytes, st)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (reps * len(qs))


for name, qs in (("cold (16 nets cycled)", list(range(16))), ("hot (net 0 repeated)", [0] * 16)):
    buf.copy_(graph.nodes * 0)
    burst(qs, 2)
    print(f"{name:28s} {burst(qs, 10):8.2f} us per launch", flush=True)
