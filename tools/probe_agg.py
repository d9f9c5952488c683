#!/usr/bin/env python3
"""Developer probe: stand-alone kernel A (gnf_aggregate_f32) bandwidth on small and large batches."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnf_amd import _abi  # noqa: E402
from gnf_amd import datasets as D  # noqa: E402
from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple  # noqa: E402

dev = torch.device("cuda:0")
lib = _abi.lib()
for name, pool, h in (("community x64", None, 32), ("ego x128", D.synthetic_ego(128), 128), ("protein x256", D.synthetic_protein(256), 32),
                      ("protein x1024", D.synthetic_protein(1024), 128)):
    if pool is None:
        ds = D.GraphDataset("graph_rnn_community_medium", h)
        pool, ids = ds.all, ds.sample_ids(64)
    else:
        ids = np.arange(len(pool))
    rng = np.random.default_rng(0)
    g = data_dicts_to_graphs_tuple(pool.data_dicts(ids, lambda n: rng.standard_normal((n, h)).astype(np.float32)), dev)
    csr = csr_of(g)
    n, e = g.nodes.shape[0], g.senders.shape[0]
    out = torch.empty(n, h, device=dev)
    st = _abi.stream_ptr(dev)
    for agg in (0, 1):
        def run():
            lib.gnf_aggregate_f32(C.byref(csr.desc), _abi.ptr(g.nodes), h, h, agg, _abi.ptr(out), h, st)
        for _ in range(5):
            run()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            run()
        b.record()
        torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 50
        alg = 8 * n * h + 4 * e + 4 * n
        gath = 4 * e * h + 4 * n * h + 4 * e + 4 * n
        print(f"{name:14s} N={n:6d} E={e:7d} H={h:3d} agg={'mean' if agg else 'sum '}: {us:8.2f} us  "
              f"algorithmic {alg / us / 1e3:7.1f} GB/s  (gathered bytes {gath / us / 1e3:7.1f} GB/s)", flush=True)
