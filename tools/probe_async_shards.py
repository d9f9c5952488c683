#!/usr/bin/env python3
"""Developer probe (round 6, VERDICT r5 "Next 1"): ONE config-2 batch cut into S whole-graph shards, each shard's forward
(gnf_grevnet_from_f32 + its log-prob sums) on its own HIP stream, the whole step captured in ONE hipGraph (fork / join by
events), for the launch shapes force_shape in {0 = automatic (16 rows x both nets), 11 = one net per workgroup, 10 / 20 =
the large-batch 4-wave kernel capped at 1 / 2 row tiles}.  Graphs are independent through all 2T half-steps
(/root/reference/gnn.py:304-341), so the shards' sums add up to the batch's.

Prints one table: ms per batch (median of R replays-bursts), per-node log-prob delta vs the single-launch step.

    python tools/probe_async_shards.py [--workload config2] [--reps 7] [--burst 50]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gnf_amd import _abi  # noqa: E402
from gnf_amd.factories import make_product_grevnet  # noqa: E402
from gnf_amd.flow import forward_shard_sums, log_prob_from_sums  # noqa: E402
from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple  # noqa: E402
from gnf_amd.sharding import shard_graph_ids  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--burst", type=int, default=50)
ap.add_argument("--shards", default="1,2,3,4,6,8")
ap.add_argument("--shapes", default="0,11,10,20")
args = ap.parse_args()

dev = torch.device("cuda:0")
bench.WORKLOAD = bench.WORKLOADS[args.workload]
bench.GRAPHS_PER_GPU = bench.WORKLOAD["graphs"]
bench.HP.update(bench.WORKLOAD["hp"])
HP = bench.HP
dicts, n_total, e_total = bench.make_batch(1, 0)
net = make_product_grevnet(HP, bench.make_params(bench.WEIGHT_SEED, HP, bench.FINAL_SCALE))
_abi.lib()
nn = np.array([d["nodes"].shape[0] for d in dicts])
ne = np.array([len(d["senders"]) for d in dicts])
print(f"workload {args.workload}: {len(dicts)} graphs, {n_total} nodes ({(n_total + 15) // 16} 16-row tiles), {e_total} edges", flush=True)


def shard_graphs(s):
    ids = shard_graph_ids(nn, ne, s)
    gs = [data_dicts_to_graphs_tuple([dicts[i] for i in idx], dev) for idx in ids]
    for g in gs:
        csr_of(g)
    return gs


def build_step(gs, streams):
    """Returns (run, sums): run() enqueues every shard's forward on its stream, forked from / joined to the current stream."""
    sums = []
    for g in gs:
        t = torch.zeros(3, dtype=torch.float64, device=dev)
        t[2] = float(g.nodes.shape[0])
        sums.append(t)

    def run():
        cur = torch.cuda.current_stream()
        if streams is None:
            for g, s3 in zip(gs, sums):
                forward_shard_sums(net, g, s3)
            return
        fork = torch.cuda.Event()
        fork.record(cur)
        for g, s3, st in zip(gs, sums, streams):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                forward_shard_sums(net, g, s3)
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)
    return run, sums


def time_graph(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(cap):
        with torch.cuda.graph(cg, stream=cap):
            run()
    for _ in range(20):
        cg.replay()
    torch.cuda.synchronize()
    ms = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.burst):
            cg.replay()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b) / args.burst)
    return float(np.median(ms)), float(np.min(ms)), cg


pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
ref_lp = None
rows = []
for shape in [int(v) for v in args.shapes.split(",")]:
    for s in [int(v) for v in args.shards.split(",")]:
        gs = shard_graphs(s)
        for multi in ((False,) if s == 1 else (True, False)):
            _abi.set_option("force_shape", shape)
            try:
                run, sums = build_step(gs, pool[:s] if multi else None)
                med, mn, cg = time_graph(run)
            except Exception as exc:   # a shape the planner refuses for this workload
                print(f"shape {shape:2d} shards {s} streams {s if multi else 1}: {exc}", flush=True)
                _abi.set_option("force_shape", 0)
                continue
            _abi.set_option("force_shape", 0)
            tot = torch.stack(sums).sum(dim=0).tolist()
            lp = log_prob_from_sums(tot, HP["D"])["log_prob_xs_per_node"]
            if ref_lp is None:
                ref_lp = lp
            rows.append((shape, s, s if multi else 1, med, mn, abs(lp - ref_lp)))
            print(f"shape {shape:2d}  shards {s}  streams {s if multi else 1}  {med:.4f} ms per batch (min {mn:.4f})  "
                  f"log-prob/node {lp:.9f}  |delta vs first arm| {abs(lp - ref_lp):.2e}", flush=True)
            del cg
base = rows[0][3]
print("\nshape shards streams  ms/batch   vs single launch")
for shape, s, k, med, mn, dlt in rows:
    print(f"{shape:5d} {s:6d} {k:7d}  {med:8.4f}   {base / med:5.3f}x   dlogp {dlt:.1e}")
