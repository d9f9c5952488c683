#!/usr/bin/env python3
"""Within-process interleaved A/B of library options (gnf_set_option) on the config-2 forward step: every
configuration is timed ROUNDS times in turn (STEPS steps each, results to pinned host memory, one sync per round) and
the median / min per configuration are printed.  Separate bench.py invocations land on different boxes, clocks and
process states; deltas under a few per cent are only readable like this (cdna_hip_programming.md, methodology 24).

    python tools/ab_options.py "base:" "late_xu:fused_late_xu=1" "old_tail:flow_separate_tail=1,flow_no_oop=1"
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from helpers import make_product_grevnet
from gnf_amd import _abi
from gnf_amd.flow import forward_shard_sums
from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple

ROUNDS, STEPS = int(os.environ.get("AB_ROUNDS", "15")), int(os.environ.get("AB_STEPS", "50"))
configs = []
for spec in sys.argv[1:] or ["base:"]:
    name, _, opts = spec.partition(":")
    configs.append((name, [(k, int(v)) for k, _, v in (o.partition("=") for o in opts.split(",") if o)]))
all_opts = sorted({k for _, o in configs for k, _ in o})
dev = torch.device("cuda:0")
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(bench.HP, bench.make_params(bench.WEIGHT_SEED, bench.HP, bench.FINAL_SCALE))
csr_of(graph)
host = torch.zeros(STEPS, 3, dtype=torch.float64).pin_memory()
times = {name: [] for name, _ in configs}
vals = {}
for rnd in range(ROUNDS + 1):
    for name, opts in configs:
        for k in all_opts:
            _abi.set_option(k, 0)
        for k, v in opts:
            _abi.set_option(k, v)
        forward_shard_sums(net, graph, host[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(STEPS):
            forward_shard_sums(net, graph, host[i])
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0) / STEPS
        if rnd:      # round 0 is warm-up
            times[name].append(dt)
        vals[name] = host[STEPS - 1, :2].tolist()
for name, _ in configs:
    a = np.array(times[name])
    print(f"{name:20s} median {np.median(a):.4f} ms  min {a.min():.4f}  max {a.max():.4f}   sums {vals[name]}")
