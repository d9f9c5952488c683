#!/usr/bin/env python3
"""Developer probe: per-wave s_memtime stamps of workgroup 0 of the TRAINING forward's fused half-step kernel
(k_half_fused<1, 2, STASH>, the last half-step of a flow call with GnfFlow.mlp_stash set), next to the inference
instance's on the same batch.  Needs a -DGNF_TRACE build: tools/build_variants.sh trace "-DGNF_TRACE"."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GNF_LIB_PATH"] = os.path.join(ROOT, "graph-normalizing-flows_amd", "variants", "libgnf_trace.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import make_product_grevnet  # noqa: E402
from gnf_amd.graphs import data_dicts_to_graphs_tuple  # noqa: E402
from gnf_amd.train import GRevNetTrainer  # noqa: E402

dev = torch.device("cuda:0")
HP = bench.HP
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(HP, bench.make_params(bench.WEIGHT_SEED, HP, bench.FINAL_SCALE))
raw = C.CDLL(os.environ["GNF_LIB_PATH"])
names = ["start", "agg done", "bar0", "L0 done", "bar", "L1 done", "bar", "L2 done", "bar", "L3 done", "bar", "L4 done",
         "bar", "-", "-", "end"]


def read(label):
    out = (C.c_ulonglong * 128)()
    assert raw.gnf_debug_read_trace(out) == 0
    t = np.array(list(out), dtype=np.int64).reshape(8, 16)
    t0 = t[:, 0].min()
    print(f"== {label}: s_memtime ticks (10 ns) relative to the first wave's start")
    print("slot            " + " ".join(f"w{w:<7d}" for w in range(8)))
    for sl in range(16):
        if names[sl] != "-":
            print(f"{names[sl]:14s} " + " ".join(f"{int(t[w, sl] - t0):8d}" for w in range(8)))
    st = (C.c_ulonglong * 320)()
    if hasattr(raw, "gnf_debug_read_stages") and raw.gnf_debug_read_stages(st) == 0:
        g = np.array(list(st), dtype=np.int64).reshape(8, 40)
        base = t[:, 4]  # barrier before layer 1
        print("layer-1 stage stamps relative to the layer's opening barrier (0 = chunk entry, 2+kg = after stage kg, 38 = MFMAs done)")
        for sl in [0] + list(range(2, 18)) + [38]:
            print(f"{sl:3d} " + " ".join(f"{int(g[w, sl] - base[w]):8d}" for w in range(8)))


for stash in (False, True):
    tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
    tr.stash_mlp_rows = stash
    for _ in range(3):
        out = tr.loss_and_grads(graph)   # forward (the traced kernel) + backward (other kernels)
    torch.cuda.synchronize()
    # the forward's last fused launch is what the stamps hold only if nothing traced ran after it: run the forward alone
    net(graph, inverse=True) if not stash else None
    torch.cuda.synchronize()
    read("training forward with the MLP-row stash" if stash else "inference instance")
