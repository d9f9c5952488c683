#!/usr/bin/env python3
"""bench.py - GRevNet forward + log-det throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or - when no
    launcher set WORLD_SIZE - bench.py starts its N ranks itself: self_launch())

One "step" = one pass of the hot path over one batch: GRevNet.f (gnn.py:304-341) + the log-prob
reductions (run_grevnet.py:290-302) on a synthetic community_medium batch, inputs (node features,
edge list -> CSR, weights) already resident in HBM, the two batch scalars landing in host memory
before the timed region closes.  Workload = BASELINE.json configs[1]: community_medium batch=64
graphs per GPU, 8-step GRevNet; hyper-parameters frozen in BASELINE.md section 4 (D=64, L=256, K=5,
avg_then_mlp eps=1, leaky_relu 0.2, sparse dataset topology + self loops, no batch norm).  With N
GPUs the batch is 64*N graphs sharded N ways (weak scaling, configs[2] at N=8) and every step ends
with the path's single collective: one all-reduce of [logdet, sum z^2, num_nodes] (3 x fp64).

Prints ONE JSON line (rank 0).  `value` = node-updates/s over all GPUs, one node-update = one node
through one GRevNet timestep (both half couplings), SURVEY.md 8d.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HP = dict(D=64, latent=256, K=5, T=8, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
          weight_sharing=False)
GRAPHS_PER_GPU = 64
# --workload: the default is BASELINE.json configs[1]; the others are the remaining configs (parity /
# shape coverage runs, not the headline bench line).  "dataset": a GraphDataset name or a generator.
WORKLOADS = {
    "config2": dict(desc="community_medium", dataset="graph_rnn_community_medium", graphs=64, hp={}, inverse=False, fc=False),
    "config2_fc": dict(desc="community_medium, fully connected topology (utils.py:164-183)", dataset="graph_rnn_community_medium",
                       graphs=64, hp={}, inverse=False, fc=True),
    "config4": dict(desc="protein stand-in (synthetic k-NN, n~U{100..500}), inverse pass", dataset="synthetic_protein",
                    graphs=256, hp={}, inverse=True, fc=False),
    "config5": dict(desc="citeseer/ego stand-in (synthetic, n~U{50..399}), node-dim 256", dataset="synthetic_ego",
                    graphs=128, hp=dict(D=256, T=16), inverse=False, fc=False),
    # the reference drivers' DEFAULT make_gnn_fn (run_grevnet.py:56,199-211): edge-list self-attention GNN
    "config2_attn": dict(desc="community_medium, dm_self_attn GNN (8 heads, kq=v=10, C=80, relu: run_grevnet.py:74-80)",
                         dataset="graph_rnn_community_medium", graphs=64,
                         hp=dict(activation="relu", attn=dict(num_heads=8, kq_dim=10, v_dim=10, out_dim=80, concat=True,
                                                              kq_dim_division=False, residual=False)),
                         inverse=False, fc=False),
    # the drivers' DEFAULT flags together at inference: dm_self_attn GNN + use_batch_norm=True (training-mode batch moments in f)
    "default_flags": dict(desc="community_medium, the drivers' default GNN (dm_self_attn) and use_batch_norm=True, forward + log-prob",
                          dataset="graph_rnn_community_medium", graphs=64,
                          hp=dict(activation="relu", use_batch_norm=True,
                                  attn=dict(num_heads=8, kq_dim=10, v_dim=10, out_dim=80, concat=True,
                                            kq_dim_division=False, residual=False)),
                          inverse=False, fc=False),
    # train_grevnet_with_data.py:104-117 defaults (10 coupling steps, 3 x 2048 MLPs, D = 200, fully connected
    # topology): too wide for the LDS-resident kernel, runs the layered path's matrix-core GEMM
    "wide_fc": dict(desc="community_medium, fully connected, latent 2048 x 3 layers, D = 200, T = 10 (layered GEMM path)",
                    dataset="graph_rnn_community_medium", graphs=64, hp=dict(D=200, latent=2048, K=3, T=10),
                    inverse=False, fc=True),
    # train_grevnet_with_data.py's LITERAL defaults (:40-46, 100-117, 303-310): dm_attn with ONE head, kq = v = 64, C = 64,
    # kq_dim_division, concat; relu MLPs 2048 x 3 (bias_init_stddev 0.3); D = 200; 10 coupling layers; use_batch_norm=True;
    # complete graphs (transform_example).  Forward + log-prob (training-mode batch moments in f)
    "data_default_flags": dict(desc="community_medium, fully connected, the DATA driver's default flags: dm_attn (1 head, kq=v=64, C=64, "
                                    "kq_dim_division) + batch norm around relu MLPs 2048 x 3, D = 200, T = 10",
                               dataset="graph_rnn_community_medium", graphs=64,
                               hp=dict(D=200, latent=2048, K=3, T=10, activation="relu", use_batch_norm=True, bias_init_stddev=0.3,
                                       attn=dict(num_heads=1, kq_dim=64, v_dim=64, out_dim=64, concat=True,
                                                 kq_dim_division=True, residual=False)),
                               inverse=False, fc=True),
    # ... and one iteration of that driver's training loop per step (train_grevnet_with_data.py:380,478-554)
    "data_default_flags_train": dict(desc="community_medium, fully connected, TRAINING step with the DATA driver's default flags: dm_attn "
                                          "(1 head, kq=v=64, C=64) + batch norm around relu MLPs 2048 x 3, D = 200, T = 10",
                                     dataset="graph_rnn_community_medium", graphs=64,
                                     hp=dict(D=200, latent=2048, K=3, T=10, activation="relu", use_batch_norm=True, bias_init_stddev=0.3,
                                             attn=dict(num_heads=1, kq_dim=64, v_dim=64, out_dim=64, concat=True,
                                                       kq_dim_division=True, residual=False)),
                                     inverse=False, fc=True, train=True),
    # the same trainer step with the message-passing GNN in place of the attention block (the wide MLPs' backward alone)
    "wide_fc_train": dict(desc="community_medium, fully connected, TRAINING step, latent 2048 x 3 layers, D = 200, T = 10 (avg_then_mlp)",
                          dataset="graph_rnn_community_medium", graphs=64, hp=dict(D=200, latent=2048, K=3, T=10),
                          inverse=False, fc=True, train=True),
    # one training iteration of run_grevnet.py:440-447 per step: forward + reversible backward + Adam + re-pack
    "config2_train": dict(desc="community_medium, TRAINING step (fwd + reversible backward + Adam)",
                          dataset="graph_rnn_community_medium", graphs=64, hp={}, inverse=False, fc=False, train=True),
    # the drivers' DEFAULT flags together: dm_self_attn GNN + use_batch_norm=True, one training iteration per step
    "default_flags_train": dict(desc="community_medium, TRAINING step with the drivers' default GNN (dm_self_attn) and use_batch_norm=True",
                                dataset="graph_rnn_community_medium", graphs=64,
                                hp=dict(activation="relu", use_batch_norm=True,
                                        attn=dict(num_heads=8, kq_dim=10, v_dim=10, out_dim=80, concat=True,
                                                  kq_dim_division=False, residual=False)),
                                inverse=False, fc=False, train=True),
}
WORKLOAD = WORKLOADS["config2"]
WEIGHT_SEED = 99
FINAL_SCALE = 0.25     # last Linear layer of every net scaled by this so |s| stays O(1) over 16 half-steps
PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak
PEAK_HBM_GBS = 8000.0


# Sources of the kernels each workload's launches live in: what a PMC pass under profiles/ is evidence FOR.  An entry of
# profiles/pmc_traffic.json is quoted only while the sha256 over its workload's set is the build's (there is no .git on the
# GPU box, so the stamp is content-based).
_SRC_FUSED = ("gnf_fused.hip", "gnf_fused_dev.h", "gnf_attn_front_dev.h")
_SRC_BIG = ("gnf_fused_big.hip", "gnf_fused_dev.h", "gnf_layered.hip")
_SRC_WIDE = ("gnf_linear_big.hip", "gnf_train.hip", "gnf_layered.hip", "gnf_fused_dev.h")
_SRC_ATTN = ("gnf_attn.hip", "gnf_attn_dev.h", "gnf_attn_core.hip", "gnf_bn.hip")
_SRC_BWD = ("gnf_train.hip", "gnf_fused_bwd.hip", "gnf_fused_bwd_dev.h", "gnf_optim.hip")
_SRC_ATTN_BWD = ("gnf_attn_bwd.hip", "gnf_attn_core_bwd.hip", "gnf_bn_bwd.hip")
WORKLOAD_SOURCES = {
    "config2": _SRC_FUSED, "config2_fc": _SRC_FUSED, "config2_attn": _SRC_FUSED, "default_flags": _SRC_FUSED,
    "config4": _SRC_BIG, "config5": _SRC_BIG,
    "wide_fc": _SRC_WIDE,
    "data_default_flags": _SRC_WIDE + _SRC_ATTN,
    "config2_train": _SRC_FUSED + _SRC_BWD,
    "default_flags_train": _SRC_FUSED + _SRC_BWD + _SRC_ATTN_BWD,
    "wide_fc_train": _SRC_WIDE + _SRC_BWD,
    "data_default_flags_train": _SRC_WIDE + _SRC_ATTN + _SRC_BWD + _SRC_ATTN_BWD,
}


def kernel_source_stamp(workload="config2"):
    """sha256 (16 hex digits) over the kernel sources of `workload` (WORKLOAD_SOURCES)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "graph-normalizing-flows_amd", "csrc")
    for name in sorted(set(WORKLOAD_SOURCES[workload])):
        h.update(name.encode())
        h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def line_consistency_errors(d):
    """Cross-checks a bench line must pass (tests/test_bench_helpers_cpu.py runs them over the lines committed under
    profiles/; main() attaches the result as `consistency`): the half-step time times the launches of a step cannot
    exceed the step, value is the step rate, the spread brackets the median."""
    errs = []
    rf = d.get("roofline") or {}
    if rf.get("kernel_us") and rf.get("launches_per_step") and "train" not in d.get("config", {}).get("workload", ""):
        if rf["kernel_us"] * rf["launches_per_step"] > 1.03 * 1e3 * d["ms_per_step"]:
            errs.append(f"kernel_us x launches_per_step = {rf['kernel_us'] * rf['launches_per_step']:.1f} us > ms_per_step "
                        f"= {1e3 * d['ms_per_step']:.1f} us")
    sp = d.get("spread")
    if sp and not (sp["ms_per_step_min"] <= d["ms_per_step"] <= sp["ms_per_step_max"]):
        errs.append("ms_per_step outside its spread")
    if rf.get("frac") is not None and rf.get("achieved") and rf.get("peak"):
        if abs(rf["frac"] - rf["achieved"] / rf["peak"]) > 2e-3:
            errs.append("roofline.frac != achieved / peak")
    return errs


def secondary_entry(d):
    """What the default run keeps of a child workload's line (secondary_workloads[wl]); training children also carry
    train_roofline's fraction / flops and the counter traffic of the trainer step's dominant kernel."""
    e = {"value": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "spread": d["spread"],
         "frac": d["roofline"]["frac"], "kernel_us": d["roofline"]["kernel_us"],
         "launches_per_step": d["roofline"]["launches_per_step"], "traffic": d["roofline"]["traffic"],
         "algorithmic_flops_per_launch": d["roofline"]["algorithmic_flops_per_launch"],
         "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"],
         "nodes": d["config"]["nodes_total"], "edges": d["config"]["edges_total"], "workload": d["config"]["workload"],
         "log_prob_xs_per_node": d.get("log_prob_xs_per_node"),
         "round_trip_max_abs_err": d.get("round_trip_max_abs_err"),
         "consistency": d.get("consistency")}
    if d.get("kernel_a") is not None:          # (absent where the flow never launches k_aggregate: attention GNNs)
        e["kernel_a"] = d["kernel_a"]
    tr = d.get("train_roofline")
    if tr:
        e["train_roofline"] = {k: tr.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_flops_per_step",
                                                      "traffic", "traffic_kernel")}
        e["train_frac"] = tr["frac"]
        e["frac_note"] = "`frac` / `kernel_us` are the forward flow alone inside the trainer step; `train_frac` is the whole step"
    return e


def percentiles(ms):
    a = np.sort(np.asarray(ms, np.float64))
    return {"p50_ms": round(float(np.percentile(a, 50)), 4), "p95_ms": round(float(np.percentile(a, 95)), 4),
            "mean_ms": round(float(a.mean()), 4), "min_ms": round(float(a[0]), 4), "max_ms": round(float(a[-1]), 4),
            "iterations": int(a.size)}


def algorithmic_half_step(n, e, hp):
    """flops and minimum HBM bytes of ONE fused coupling half-step launch."""
    h, l, k = hp["D"] // 2, hp["latent"], hp["K"]
    in0 = 2 * h if hp["combine"] == "concat" else h
    att = hp.get("attn")
    extra_f = extra_b = 0
    if att:   # per net: q/k/v projection, logits (two passes), weighted values, output projection
        nh, kq, vd, c = att["num_heads"], att["kq_dim"], att["v_dim"], att["out_dim"]
        in0 = (h if att["concat"] else 0) + c
        pw_att = h * (2 * nh * kq + vd) + nh * vd * c
        extra_f = 2 * (2 * n * pw_att + e * nh * (4 * kq + 2 * vd + 4))
        extra_b = 8 * pw_att
    if k == 1:
        p_w, p_b = in0 * h, h
    else:
        p_w = in0 * l + (k - 2) * l * l + l * h
        p_b = (k - 1) * l + h
    flops = n * 4 * p_w + (0 if att else e * h) + 6 * n * h + extra_f
    bytes_ = 12 * n * h + 4 * e + 4 * n + 8 * (p_w + p_b) + extra_b
    return flops, bytes_


def make_params(seed, hp, final_scale):
    """Bench weights (same layout as tests' fixtures): W ~ N(0, 2/(fan_in+fan_out)) (glorot variance,
    gnn.py:171-172), b ~ N(0, 0.1) clipped at 2 sigma (gnn.py:173), last layer x final_scale."""
    rng = np.random.default_rng(seed)
    h, l, k, t = hp["D"] // 2, hp["latent"], hp["K"], hp["T"]
    in0 = 2 * h if hp["combine"] == "concat" else h
    att = hp.get("attn")
    if att:
        in0 = (h if att["concat"] else 0) + att["out_dim"]

    def attn_weights():   # xavier-uniform q/k/v (gnn.py:504-506), 1/sqrt(fan_in) output projection
        nq = att["num_heads"] * att["kq_dim"]
        def xav(fi, fo):
            a = np.sqrt(6.0 / (fi + fo))
            return rng.uniform(-a, a, size=(fi, fo)).astype(np.float32)
        nv = att["num_heads"] * att["v_dim"]
        return dict(att, wq=xav(h, nq), wk=xav(h, nq), wv=xav(h, att["v_dim"]),
                    wo=(rng.standard_normal((nv, att["out_dim"])) / np.sqrt(nv)).astype(np.float32))

    def mlp():
        layers, fan_in = [], in0
        sizes = [l] * (k - 1) + [h]
        for j, fan_out in enumerate(sizes):
            w = rng.standard_normal((fan_in, fan_out)) * np.sqrt(2.0 / (fan_in + fan_out))
            b = np.clip(rng.standard_normal(fan_out), -2.0, 2.0) * 0.1
            if j == len(sizes) - 1:
                w, b = w * final_scale, b * final_scale
            layers.append((w.astype(np.float32), b.astype(np.float32)))
            fan_in = fan_out
        return layers

    def net():
        return {"attn": attn_weights(), "mlp": mlp()} if att else mlp()

    out = {"s": [[net() for _ in range(t)] for _ in range(2)], "t": [[net() for _ in range(t)] for _ in range(2)]}
    if hp.get("use_batch_norm"):   # the reference's initial values (tf.layers.BatchNormalization defaults)
        out["bn"] = [[{"gamma": np.ones(h, np.float32), "beta": np.zeros(h, np.float32),
                       "moving_mean": np.zeros(h, np.float32), "moving_variance": np.ones(h, np.float32)}
                      for _ in range(t)] for _ in range(2)]
    return out


def make_batch(n_gpus, rank, seed=12345):
    """Global batch = 64*n_gpus graphs drawn with replacement from the 80% train split
    (graph_data.py:77-78,112-122), sharded by greedy balance on nodes+edges; returns this rank's shard."""
    from gnf_amd import datasets as D
    from gnf_amd.sharding import shard_graph_ids
    g_total = GRAPHS_PER_GPU * n_gpus
    name = WORKLOAD["dataset"]
    if name == "synthetic_protein":
        pool = D.synthetic_protein(g_total, seed=seed)
        ids = np.arange(g_total)
    elif name == "synthetic_ego":
        pool = D.synthetic_ego(g_total, seed=seed)
        ids = np.arange(g_total)
    else:
        ds = D.GraphDataset(name, HP["D"], seed=seed)
        pool, ids = ds.all, ds.sample_ids(g_total)
    if WORKLOAD["fc"]:
        pool = D.with_fully_connected_topology(pool)
    nn, ne = pool.n_node[ids], pool.n_edge[ids]
    mine = shard_graph_ids(nn, ne, n_gpus)[rank]           # positions in the global batch, ascending
    # node features belong to the GLOBAL batch (drawn in batch order from one stream), so that N ranks hold exactly
    # the batch one rank would: the all-reduced log-prob of an N-rank run equals a 1-rank run over 64*N graphs
    rng = np.random.default_rng(seed + 1000)
    mine_set, feats = set(mine.tolist()), {}
    for pos, n in enumerate(nn):
        f = rng.standard_normal((int(n), HP["D"])).astype(np.float32)
        if pos in mine_set:
            feats[pos] = f
    it = iter([feats[p] for p in mine])
    dicts = pool.data_dicts(ids[mine], lambda n: next(it))
    return dicts, int(nn.sum()), int(ne.sum())


def cpu_baseline(dicts, params, budget_s=15.0, min_iters=3, warm=1):
    """The oracle's fp32 gather/index_add restatement (reference op order, un-fused) timed on this
    box's host cores: 'CPU restatement of reference (TensorFlow unavailable)', BASELINE.md section 3."""
    from oracle import gnf_oracle as O
    from gnf_amd.graphs import data_dicts_to_graphs_tuple
    g = data_dicts_to_graphs_tuple(dicts)
    n = g.nodes.shape[0]
    o = O.Fp32Gather(g.senders.numpy(), g.receivers.numpy(), n, agg=HP["agg"], combine=HP["combine"],
                     epsilon=HP["epsilon"], activation=HP["activation"])
    pt = o.prep_params(params)
    x = g.nodes.clone()
    # torch-CPU defaults to one thread per logical core, which is far from optimal for these small
    # un-fused ops on a many-core host: pick the best thread count first (each candidate: 1 warm + 1
    # timed forward), then time the bounded sample at that setting.  A fair baseline, not a strawman.
    max_thr = torch.get_num_threads()
    cands = sorted({t for t in (max_thr, 64, 32, 16, 8, 4) if 1 <= t <= max_thr}, reverse=True)
    best_thr, best_t, sweep = max_thr, float("inf"), []
    for thr in cands:
        torch.set_num_threads(thr)
        o.log_prob(x, pt, HP["T"])
        t0 = time.perf_counter()
        o.log_prob(x, pt, HP["T"])
        dt = time.perf_counter() - t0
        sweep.append(f"{thr}thr:{1e3 * dt:.0f}ms")
        if dt < best_t:
            best_thr, best_t = thr, dt
    torch.set_num_threads(best_thr)
    res = None
    for _ in range(warm):
        res = o.log_prob(x, pt, HP["T"])
    iters, t0, per_iter = 0, time.perf_counter(), []
    while True:
        t1 = time.perf_counter()
        res = o.log_prob(x, pt, HP["T"])
        iters += 1
        dt = time.perf_counter() - t0
        per_iter.append(1e3 * (time.perf_counter() - t1))
        if iters >= min_iters and (dt >= budget_s or iters >= 200):
            break
    torch.set_num_threads(max_thr)
    pc = percentiles(per_iter)
    return {"value": n * HP["T"] * iters / dt, "unit": "node-updates/s", "cores": best_thr,
            "kind": "port", "ms_per_step": 1e3 * dt / iters, "p50_ms": pc["p50_ms"], "p95_ms": pc["p95_ms"],
            "sample": f"{iters} forwards of the same rank-0 batch (N={n} nodes, T={HP['T']}) in {dt:.1f}s, "
                      f"torch-CPU fp32 restatement of gnn.py in reference op order; thread sweep "
                      f"[{' '.join(sweep)}] -> {best_thr} threads; os.cpu_count()={os.cpu_count()}"}, res


def self_launch(n, argv, backend):
    """`python bench.py --gpus N ...` with no launcher in front (WORLD_SIZE unset): start the N ranks ourselves - one child
    process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment exactly as torch.distributed.run would
    set them, rendezvous on 127.0.0.1 and a free port.  Rank 0 owns stdout (the ONE JSON line); the other ranks' stdout
    goes to stderr; stderr is shared.  Returns the exit code: 0 only if every rank exited 0; the first failing rank's
    code otherwise (the remaining ranks are terminated - they would sit in a collective for ever).  With fewer than N
    devices a JSON line with an `error` field is printed instead of hanging in init_process_group."""
    import socket
    import subprocess
    one_device = os.environ.get("GNF_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    err = None
    if have < (1 if one_device else n):
        err = f"bench.py --gpus {n}: {have} HIP device(s) visible, {1 if one_device else n} needed"
    elif one_device and backend == "nccl":
        err = "GNF_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and RCCL refuses two ranks of one device: add --dist-backend gloo"
    if err:
        print(json.dumps({"error": err, "n_gpus": n, "devices_visible": have}), flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GNF_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.time() + float(os.environ.get("GNF_BENCH_LAUNCH_TIMEOUT_S", "3000"))
    rc, live = 0, set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    print(f"[bench.py launcher] rank {r} exited with code {code}", file=sys.stderr, flush=True)
                    rc = code if code > 0 else 1
        if time.time() > deadline:
            print("[bench.py launcher] timed out waiting for the ranks", file=sys.stderr, flush=True)
            rc = 124
        if live and rc == 0:
            time.sleep(0.05)
    for r in live:                      # (exactly the processes started above, by handle)
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=20)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="device pre-warm in front of the W warmup steps: the same step in a loop for this long (a fresh "
                         "process runs its first ~100 iterations up to 8 %% slower - clocks, first touches); 0 = off")
    ap.add_argument("--layered", action="store_true", help="run the generic layered kernels instead of the fused one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary legs (CSR rebuild per step, host_fed, two_streams): what the rocprofv3 "
                         "summaries under profiles/ are taken with, so that a kernel's average is the timed region's")
    ap.add_argument("--sync-each-step", action="store_true", help="latency mode: host waits for every step's scalar")
    ap.add_argument("--kernel-timing-steps", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config2")
    ap.add_argument("--graphs-per-gpu", type=int, default=0,
                    help="override the workload's graphs per rank (tests: one rank over the batch that N ranks shard)")
    ap.add_argument("--latency-steps", type=int, default=200, help="iterations of the p50/p95 leg (0: skip)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group + collectives) even with one rank")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each, back to back: the line reports the median one (+ spread)")
    ap.add_argument("--no-secondary-workloads", action="store_true",
                    help="the default config2 run on one GPU also runs config4, config5, config2_attn and default_flags as child "
                         "processes (3 timed regions each) and attaches their headline fields as secondary_workloads - the "
                         "at-scale roofline fractions in the same driver-timed line; this flag (or --no-secondary) turns that off")
    ap.add_argument("--secondary-steps", type=int, default=0, help="steps per timed region of the secondary workloads (0: --steps, capped at 40)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + GNF_BENCH_ONE_DEVICE=1 runs several ranks on ONE GPU to exercise the N>1 logic")
    args = ap.parse_args()
    global WORKLOAD, GRAPHS_PER_GPU
    WORKLOAD = WORKLOADS[args.workload]
    GRAPHS_PER_GPU = args.graphs_per_gpu or WORKLOAD["graphs"]
    HP.update(WORKLOAD["hp"])
    inverse = WORKLOAD["inverse"]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself (no launcher in front): this process becomes the launcher of its N ranks
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.dist_backend))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world          # under a launcher the launcher's world size is the truth
    if os.environ.get("GNF_BENCH_TEST_FAIL_RANK") == str(rank) and world > 1:      # tests: a rank that dies before the rendezvous
        raise SystemExit(7)
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if os.environ.get("GNF_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # --force-dist: ONE rank through every branch of the N > 1 path (process group, asynchronous all-reduce of the
    # shard sums, drain, barrier, MAX over ranks, shard-balance gather) - on a 1-GPU box the RCCL side of bench.py's
    # sharded protocol would otherwise first run on the driver's 8-GPU node
    multi = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    from gnf_amd.factories import make_product_grevnet
    from gnf_amd import _abi
    from gnf_amd.flow import forward_shard_sums, log_prob_from_sums
    from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple
    from gnf_amd.sharding import all_reduce_shard_sums, all_reduce_shard_sums_async
    _abi.lib()

    dicts, n_global, e_global = make_batch(world, rank)
    params = make_params(WEIGHT_SEED, HP, FINAL_SCALE)
    graph = data_dicts_to_graphs_tuple(dicts, dev)
    n_local, e_local = int(graph.nodes.shape[0]), int(graph.senders.shape[0])
    net = make_product_grevnet(HP, params)
    net.fused = not args.layered
    # batch-norm workloads under sharding: the bijectors take their moments over the whole (global) batch like the
    # single-device reference does (one small all-reduce per bijector call)
    net.sync_batch_norm = bool(HP.get("use_batch_norm")) and multi

    from gnf_amd.graphs import build_csr_device
    build_csr_device(graph)                  # warm (first-call kernel load), then time one build
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    build_csr_device(graph)
    torch.cuda.synchronize()
    csr_ms = 1e3 * (time.perf_counter() - t0)
    csr = csr_of(graph)                      # device CSR (gnf_build_csr), cached per batch afterwards

    sums3 = torch.zeros(3, dtype=torch.float64, device=dev)
    sums3[2] = float(n_local)
    host = torch.zeros(args.steps + args.warmup + 1, 3, dtype=torch.float64).pin_memory()
    host[:, 2] = float(n_local)          # single-GPU steps write [logdet, sum z^2] of step i into host[i, :2] directly
    host[:, :2] = float("nan")           # "not landed" marker

    trainer = None
    if WORKLOAD.get("train"):
        from gnf_amd.train import GRevNetTrainer
        trainer = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
    # Training workloads step on ONE fixed batch of noise: a wide flow memorises it and the likelihood grows without bound
    # (finite through 300 steps of the data driver's nets, NaN somewhere past that - measured).  Every timed region
    # therefore starts from the same initial variables and optimiser state (restored OUTSIDE the timed region, the
    # gnf_amd.train.trainer_state / load_trainer_state), and the line's log-prob is the one after exactly K steps from there.
    state0 = None

    def step(i):
        if trainer is not None:   # gradient all-reduce (one flat RCCL all-reduce) when sharded
            trainer.step(graph, all_reduce=multi)
            return
        if inverse:   # config 4: sampling direction g (gnn.py:343-373); no scalar comes back
            net(graph, inverse=False)
            return
        if multi:
            # two alternating sum buffers: batch i's 3-scalar all-reduce runs on RCCL's stream while batch i + 1 is
            # already computing; its result is read (stream-ordered wait, pinned-host copy) one step later
            buf = sums_pair[i & 1]
            _, s3 = forward_shard_sums(net, graph, buf)
            s3[2] = float(n_local)           # all-reduce sums in place: restore this rank's count first
            pending.append((all_reduce_shard_sums_async(s3, force=args.force_dist), s3, i))
            if len(pending) > 1 or args.sync_each_step:
                drain(1 if not args.sync_each_step else len(pending))
        else:
            # one GPU: the final reduction kernel writes the two batch scalars straight into this step's row of the
            # pinned host buffer (device-visible host memory; no device-to-host copy node behind the flow)
            forward_shard_sums(net, graph, host[i])
        if args.sync_each_step:
            torch.cuda.current_stream().synchronize()

    sums_pair = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in range(2)]
    pending = []

    def drain(count):
        for _ in range(min(count, len(pending))):
            work, t, j = pending.pop(0)
            work.wait()
            host[j].copy_(t, non_blocking=True)

    if trainer is not None:
        from gnf_amd.train import load_trainer_state, trainer_state
        trainer.loss_and_grads(graph)        # (connects the variables)
        state0 = trainer_state(trainer)
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        t_pre = time.perf_counter()
        # (sharded runs: a fixed count, the same on every rank - the steps hold collectives)
        fixed = int(args.prewarm_ms / 1.0) if multi else 0
        while (prewarm_steps < fixed) if multi else (time.perf_counter() - t_pre < 1e-3 * args.prewarm_ms):
            for _ in range(8):
                step(0)
            prewarm_steps += 8
            drain(len(pending))
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    drain(len(pending))
    torch.cuda.synchronize()

    def timed_region():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; MAX over ranks.  Every step of a region
        writes the same host rows (args.warmup + i): the rows are results, not a log."""
        if state0 is not None:
            load_trainer_state(trainer, state0)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        t_enq = time.perf_counter() - t0      # the host has handed over every step (not a result: says whether the
                                              # loop is bound by the device or by the host's launch rate)
        drain(len(pending))                   # inside the timed region: every batch's log-prob has reached the host buffer
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        return el, t_enq

    # the timed region is repeated --repeats times back to back (each one exactly K steps): ms_per_step / value are the
    # MEDIAN region's, `spread` carries min / max - one 20-step region of a 0.56 ms step is 11 ms of a fresh box
    regions = [timed_region() for _ in range(max(1, args.repeats))]
    order = sorted(range(len(regions)), key=lambda q: regions[q][0])
    elapsed, t_enqueued = regions[order[len(order) // 2]] if len(order) % 2 else regions[order[len(order) // 2 - 1]]
    spread = {"timed_regions": len(regions), "steps_each": args.steps,
              "ms_per_step_min": round(1e3 * regions[order[0]][0] / args.steps, 4),
              "ms_per_step_median": round(1e3 * elapsed / args.steps, 4),
              "ms_per_step_max": round(1e3 * regions[order[-1]][0] / args.steps, 4),
              "ms_per_step_first": round(1e3 * regions[0][0] / args.steps, 4)}
    ms_per_step = 1e3 * elapsed / args.steps
    if trainer is not None:
        host[args.warmup + args.steps - 1, :2] = net.last_sums[:2].cpu()
        host[args.warmup + args.steps - 1, 2] = float(n_local)
    last = ({"log_prob_xs_per_node": None} if inverse else
            log_prob_from_sums(host[args.warmup + args.steps - 1].tolist(), HP["D"]))
    value = n_global * HP["T"] * args.steps / elapsed

    # ---- secondary figure: the same step when the batch's topology is new every step (training loop
    # of run_grevnet.py:440-447 draws a fresh batch per step): CSR rebuilt on device inside the step
    rebuild = None
    if not inverse and not multi and trainer is None and not args.no_secondary:
        from gnf_amd.graphs import clear_csr_cache
        nreb = max(10, min(50, args.steps))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nreb):
            clear_csr_cache()
            step(args.warmup + args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        rebuild = {"ms_per_step": round(1e3 * dt / nreb, 4), "value": round(n_global * HP["T"] * nreb / dt, 1),
                   "note": "gnf_build_csr (one launch: a workgroup per graph) + torch allocations inside every step"}

    # ---- secondary figure: the PCIe-inclusive rate.  The C ABI takes device pointers, but the reference's drivers
    # hand every batch over as host arrays (feed_dict of a GraphsTuple, run_grevnet.py:440-447): here the batch's
    # nodes / senders / receivers / n_node / n_edge start in pinned host memory every step, are uploaded, the CSR
    # is rebuilt on device, then the same forward runs.  Never the headline value.
    host_fed = None
    if not inverse and not multi and trainer is None and not args.no_secondary:
        from gnf_amd.graphs import clear_csr_cache
        fields = ("nodes", "senders", "receivers", "n_node", "n_edge")
        # ONE packed pinned staging buffer and ONE device landing buffer, both allocated once: a batch is one
        # host-to-device copy (the five separate small copies + per-step allocations of the first version made this
        # figure vary by 4x between boxes); the GraphsTuple fields are typed views into the landing buffer
        parts, off = {}, 0
        for f in fields:
            t = getattr(graph, f)
            nb = t.numel() * t.element_size()
            parts[f] = (off, nb, t.dtype, tuple(t.shape))
            off += (nb + 255) // 256 * 256
        stage = torch.empty(off, dtype=torch.uint8).pin_memory()
        land = torch.empty(off, dtype=torch.uint8, device=dev)
        for f in fields:
            o_, nb, _, _ = parts[f]
            stage[o_:o_ + nb].copy_(getattr(graph, f).cpu().contiguous().view(torch.uint8).reshape(-1))
        views = {f: land[o_:o_ + nb].view(dt).reshape(shape) for f, (o_, nb, dt, shape) in parts.items()}
        g2 = graph._replace(**views)
        nfed = max(10, min(50, args.steps))

        def fed_step(i):
            land.copy_(stage, non_blocking=True)
            clear_csr_cache()                      # the topology is "new": rebuild the CSR on device
            _, s3 = forward_shard_sums(net, g2, sums3)
            host[i].copy_(s3, non_blocking=True)
        for i in range(3):
            fed_step(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nfed):
            fed_step(args.warmup + args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        clear_csr_cache()
        up = sum(nb for (_, nb, _, _) in parts.values())
        host_fed = {"ms_per_step": round(1e3 * dt / nfed, 4), "value": round(n_global * HP["T"] * nfed / dt, 1),
                    "uploaded_bytes_per_step": up,
                    "note": "GraphsTuple fields packed in ONE pinned host buffer each step: one upload + gnf_build_csr + forward"}

    # ---- secondary figure: independent forwards on TWO HIP streams.  A 64-graph batch fills 170 of the 256 CUs
    # (one 16-node tile per CU); evaluation of many batches (the steps here are independent of each
    # other) can use the idle third of the chip by keeping two batches in flight.  Not the headline protocol
    # (one batch at a time); reported next to it.
    two_streams = None
    if not inverse and not multi and trainer is None and not args.no_secondary:
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        bufs = [torch.zeros(3, dtype=torch.float64, device=dev) for _ in streams]
        host2 = torch.zeros(args.steps, 3, dtype=torch.float64).pin_memory()
        for b in bufs:
            b[2] = float(n_local)

        def step2(i):
            q = i & 1
            with torch.cuda.stream(streams[q]):
                _, s3 = forward_shard_sums(net, graph, bufs[q])
                host2[i].copy_(s3, non_blocking=True)
        torch.cuda.synchronize()
        for i in range(min(10, args.steps)):
            step2(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step2(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        same = bool(torch.equal(host2[args.steps - 1], host2[args.steps - 2])) and \
            abs(log_prob_from_sums(host2[args.steps - 1].tolist(), HP["D"])["log_prob_xs_per_node"]
                - last["log_prob_xs_per_node"]) == 0.0
        two_streams = {"ms_per_step": round(1e3 * dt / args.steps, 4), "value": round(n_global * HP["T"] * args.steps / dt, 1),
                       "results_identical_to_single_stream": same,
                       "note": "same forwards, two batches in flight on two HIP streams (throughput mode)"}

    # ---- steady-state p50 / p95 per step, "inputs in memory -> scalar log-prob on the host":
    # every iteration is closed by a stream synchronise (latency mode), >= 10 warm-ups, args.latency_steps iterations
    latency = None
    if not inverse and not multi and trainer is None and args.latency_steps > 0:
        cur = torch.cuda.current_stream()
        for i in range(10):
            step(args.warmup)
            cur.synchronize()
        lat = []
        for i in range(args.latency_steps):
            t1 = time.perf_counter()
            step(args.warmup)
            cur.synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        latency = dict(percentiles(lat), mode="one batch at a time, host waits for each step's scalars (sync per step)")

    # ---- the same step replayed from a captured HIP graph (one hipGraphLaunch per step instead of ~20 launches):
    # removes the host's launch work from the loop; the device-side kernel boundaries stay
    graph_replay = None
    if not inverse and not multi and trainer is None and not args.no_secondary:
        gs3 = torch.zeros(3, dtype=torch.float64, device=dev)
        gs3[2] = float(n_local)
        ghost = torch.zeros(1, 3, dtype=torch.float64).pin_memory()
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            forward_shard_sums(net, graph, gs3)
            ghost[0].copy_(gs3, non_blocking=True)
        for _ in range(5):
            cg.replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            cg.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        same = abs(log_prob_from_sums(ghost[0].tolist(), HP["D"])["log_prob_xs_per_node"] - last["log_prob_xs_per_node"]) == 0.0
        graph_replay = {"ms_per_step": round(1e3 * dt / args.steps, 4), "value": round(n_global * HP["T"] * args.steps / dt, 1),
                        "results_identical_to_eager": bool(same),
                        "note": "torch.cuda.graph capture of gnf_grevnet_f32 + the pinned-host copy, replayed per step"}
        del cg

    # ---- BASELINE.md section 4, config 2 secondary: the same batch topology at D = 100 (run_gnn.py:111; H = 50 is not
    # a multiple of 16: padded fragments)
    d100 = None
    if args.workload == "config2" and not multi and not args.no_secondary and not args.layered:
        hp100 = dict(HP, D=100)
        rng100 = np.random.default_rng(77)
        g100 = graph.replace(nodes=torch.as_tensor(rng100.standard_normal((n_local, 100)).astype(np.float32)).to(dev))
        net100 = make_product_grevnet(hp100, make_params(WEIGHT_SEED, hp100, FINAL_SCALE))
        s100 = torch.zeros(3, dtype=torch.float64, device=dev)
        for _ in range(5):
            forward_shard_sums(net100, g100, s100)
        torch.cuda.synchronize()
        n100 = max(20, min(100, args.steps))
        t1 = time.perf_counter()
        for _ in range(n100):
            forward_shard_sums(net100, g100, s100)
            host[0].copy_(s100, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        f100, _ = algorithmic_half_step(n_local, e_local, hp100)
        d100 = {"ms_per_step": round(1e3 * dt / n100, 4), "value": round(n_global * HP["T"] * n100 / dt, 1),
                "frac_of_fp32_matrix_peak_end_to_end": round(f100 * 2 * HP["T"] / (dt / n100) / 1e12 / PEAK_FP32_MATRIX_TFLOPS, 4),
                "note": "D=100 (H=50): same graphs, weights seed 99; unpadded algorithmic flops"}
        del net100, g100

    # ---- half-step timing with HIP events on the launch stream, around the PRODUCT's own flow call (the path the timed
    # steps ran: out-of-place first half-step, packed attention front-end, large-batch kernel + its aggregation launch,
    # the final reduction).  kernel_us = flow time / 2T: the dominant kernel's launch plus its share of the small
    # launches around it - an upper bound of the kernel's own duration (the rocprofv3 per-kernel averages are under
    # profiles/), so kernel_us x launches_per_step <= ms_per_step up to timer noise.  (Round 2 timed 2T calls of
    # gnf_coupling_half_f32 here, which for attention nets took the unpacked two-launch front-end the step never runs.)
    import ctypes as C
    lib = _abi.lib()
    h = HP["D"] // 2
    st = _abi.stream_ptr(dev)
    evs = []
    ksums = torch.zeros(3, dtype=torch.float64, device=dev)
    # (events made and recorded once up front: creating them between the launches stalled the host for 35-55 ms once in ~12
    # calls - the device then idles, clocks down, and the following intervals come back 5-9 % long while it ramps up again)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.kernel_timing_steps + 3)]
    for a, b in pairs:
        a.record(), b.record()
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 1e-3 * min(args.prewarm_ms, 100.0):   # the clocks are back up before the first timed pair
        net(graph, inverse=False) if inverse else forward_shard_sums(net, graph, ksums)
    for it in range(args.kernel_timing_steps + 3):
        a, b = pairs[it]
        a.record()
        if inverse:
            net(graph, inverse=False)
        else:
            forward_shard_sums(net, graph, ksums)
        b.record()
        if 2 <= it < args.kernel_timing_steps + 2:   # (the first two pairs warm up, the last one is not used either)
            evs.append((a, b))
    torch.cuda.synchronize()
    if os.environ.get("GNF_BENCH_DEBUG"):
        print("flow event times (ms):", [round(a.elapsed_time(b), 3) for a, b in evs], file=sys.stderr)
    # MEDIAN over the pairs: one pair in ~12 of the attention workload comes back 45-55 ms long (a host-side stall between
    # two launches of that call - the device interval includes the idle time; the timed regions above do not show it)
    kernel_us = 1e3 * float(np.median([a.elapsed_time(b) for a, b in evs])) / (2 * HP["T"]) if evs else float("nan")
    # ---- kernel A alone (gnf_aggregate_f32: the CSR segment-reduce the north_star asks HBM evidence for;
    # on the hot path it is fused into the half-step kernel's prologue) ------------------------------
    kernel_a = None      # (attention GNNs replace the aggregation: a workload that never launches k_aggregate reports none)
    if not HP.get("attn"):
        agg_out = torch.empty(n_local, h, dtype=torch.float32, device=dev)
        xin = graph.nodes[:, :h]

        def run_agg():
            _abi.check(lib.gnf_aggregate_f32(C.byref(csr.desc), _abi.ptr(xin), xin.stride(0), h,
                                             _abi.GNF_AGG_MEAN if HP["agg"] == "mean" else _abi.GNF_AGG_SUM,
                                             _abi.ptr(agg_out), h, st), "gnf_aggregate_f32")
        for _ in range(5):
            run_agg()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(50):
            run_agg()
        eb.record()
        torch.cuda.synchronize()
        agg_us = 1e3 * ea.elapsed_time(eb) / 50
        agg_bytes = 8 * n_local * h + 4 * e_local + 4 * n_local
        kernel_a = {"kernel": "k_aggregate<4> (gnf_aggregate_f32)", "us": round(agg_us, 2),
                    "algorithmic_bytes": agg_bytes, "achieved_gbs": round(agg_bytes / agg_us / 1e3, 1),
                    "peak_gbs": PEAK_HBM_GBS, "frac": round(agg_bytes / agg_us / 1e3 / PEAK_HBM_GBS, 4),
                    "traffic": None, "traffic_gbs": None,
                    "note": "HIP events around gnf_aggregate_f32 alone, back to back (the launch inside a flow sees the previous "
                            "kernel's rows come back through memory: the rocprofv3 average next to `traffic`); launch-latency-bound "
                            "on small batches (a few us); 2.9 TB/s algorithmic on 77k-308k node batches (tools/probe_agg.py, profiles/)"}
        try:   # counter evidence of kernel A inside this workload's flow, when a PMC pass of this build exists (k_half_big workloads)
            pm_a = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            ka = next((v for k, v in pm_a.get("workloads", {}).get(args.workload, {}).get("kernels", {}).items() if k.startswith("k_aggregate")), None)
            if ka and pm_a["workloads"][args.workload].get("source_stamp") == kernel_source_stamp(args.workload):
                kernel_a["traffic"] = round(ka["traffic_bytes_per_launch"])
                kernel_a["traffic_gbs"] = round(ka["hbm_gbs_over_rocprof_avg"], 1)
                kernel_a["traffic_note"] = (f"rocprofv3 --pmc FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE of k_aggregate inside the flow, "
                                            f"mean per launch; / its rocprofv3 average of {ka['rocprof_avg_us']} us = HBM-side GB/s "
                                            f"(profiles/pmc_traffic.json, tag {pm_a['workloads'][args.workload].get('tag')}, kernel sources = this build)")
        except (OSError, ValueError, KeyError):
            pass

    flops, abytes = algorithmic_half_step(n_local, e_local, HP)
    # HBM-side bytes per launch come from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    # correction + WRITE_SIZE; tools/pmc_shape.sh + tools/summarize_profile.py): PMC counters cannot be
    # collected from inside this process.  Only quoted for the workload they were measured on.
    traffic, traffic_note, traffic_kernel = None, "no PMC pass recorded for this workload", None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        ent = pm.get("workloads", {}).get(args.workload)
        if ent and net.fused:
            if ent.get("source_stamp") == kernel_source_stamp(args.workload):
                traffic = round(ent["traffic_bytes_per_launch"])
                traffic_kernel = ent.get("kernel")
                traffic_note = (f"HBM-side bytes per launch of {ent.get('kernel')} from profiles/pmc_traffic.json "
                                f"(rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE passes, tag {ent.get('tag')}, "
                                f"kernel sources {ent.get('source_stamp')} = this build's {', '.join(sorted(set(WORKLOAD_SOURCES[args.workload])))})")
            else:
                traffic_note = (f"profiles/pmc_traffic.json's entry was taken on kernel sources {ent.get('source_stamp')}, this build's "
                                f"{args.workload} sources are {kernel_source_stamp(args.workload)}: not quoted (re-run tools/final_profile.sh pmc)")
    except (OSError, ValueError, KeyError):
        pass
    achieved_tflops = flops / (kernel_us * 1e-6) / 1e12
    roofline = {"bound": "mfma", "achieved": round(achieved_tflops, 3), "peak": PEAK_FP32_MATRIX_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved_tflops / PEAK_FP32_MATRIX_TFLOPS, 4), "traffic": traffic,
                "traffic_note": traffic_note,
                "kernel": ("one coupling half-step = k_half_fused, or k_aggregate + k_half_big on large batches (+ the attention "
                           "front-end / k_coupling where the path has them): HIP events around the product's flow call / 2T "
                           "(includes the final reduction's share: an upper bound of the kernel's own time)") if net.fused else
                          "layered half-step (aggregate + 2K x k_linear + k_coupling)",
                "kernel_us": round(kernel_us, 2), "launches_per_step": 2 * HP["T"],
                "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": abytes,
                "hbm_floor_us": round(abytes / (PEAK_HBM_GBS * 1e3), 3)}

    out = {
        "metric": ("node-updates/sec (fwd+logdet) on community_medium batch" if args.workload == "config2" else
                   f"node-updates/sec ({'inverse' if inverse else 'fwd+logdet'}) on {args.workload}"), "value": round(value, 1),
        "unit": "node-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "spread": spread, "prewarm_steps": prewarm_steps, "host_enqueue_ms_per_step": round(1e3 * t_enqueued / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {WORKLOAD['desc']} batch={GRAPHS_PER_GPU}/GPU ({GRAPHS_PER_GPU * world} graphs total), "
                               f"{HP['T']}-step GRevNet {'inverse (sampling)' if inverse else 'fwd+logdet'}, D={HP['D']} L={HP['latent']} K={HP['K']} "
                               + ("dm_self_attn GNN relu" if HP.get("attn") else "avg_then_mlp eps=1 leaky_relu(0.2)")
                               + (", fully connected topology" if WORKLOAD["fc"] else ", sparse topology+self loops"),
                   "nodes_total": n_global, "edges_total": e_global, "nodes_rank0": n_local, "edges_rank0": e_local,
                   "weights": f"N(0,2/(fan_in+fan_out)), seed {WEIGHT_SEED}, last layer x{FINAL_SCALE}",
                   "parallelism": f"graph-shard dp{world}, 1 all-reduce of 3xfp64 per step (async: overlapped with the next batch's forward)" if multi else "single GPU",
                   "path": "fused MFMA half-step kernel" if net.fused else "layered kernels",
                   "csr": f"built on device once per batch before the timed region (gnf_build_csr: {csr_ms:.3f} ms wall incl. host launch), cached",
                   "host_sync": "every step" if args.sync_each_step else "results copied to pinned host memory each step; one sync at end"},
        "log_prob_xs_per_node": last["log_prob_xs_per_node"],
        "steps_landed_on_host": int(torch.isfinite(host[args.warmup:args.warmup + args.steps, :2]).all(dim=1).sum()) if trainer is None and not inverse else None,
        "latency": latency,
        "graph_replay": graph_replay,
        "secondary_D100": d100,
        "with_csr_rebuild_each_step": rebuild,
        "two_streams": two_streams,
        "host_fed": host_fed,
        "roofline": roofline,
        "kernel_a": kernel_a,
    }

    if trainer is not None:
        # one trainer step = forward + dL/dx chain + weight gradients: 3 x the forward's algorithmic flops (what tf.gradients
        # executes; the reversible walk's recompute, where a stash does not replace it, is NOT counted as useful work)
        t_tf = 3.0 * flops * 2 * HP["T"] / (ms_per_step * 1e-3) / 1e12
        out["train_roofline"] = {"bound": "mfma", "achieved": round(t_tf, 3), "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(t_tf / PEAK_FP32_MATRIX_TFLOPS, 4),
                                 "algorithmic_flops_per_step": 3 * flops * 2 * HP["T"],
                                 "traffic": traffic, "traffic_kernel": traffic_kernel,
                                 "note": "3 x the forward's algorithmic flops per step (forward, dX chain, dW) / ms_per_step: the whole "
                                         "step incl. Adam, the re-pack and every small launch; `roofline` above is the forward flow alone"}
        out["train_state"] = ("variables and optimiser state restored to the initial ones before every timed region (outside it): "
                              "log_prob_xs_per_node is the value after exactly `steps` steps on the one fixed batch")
    if inverse:   # round-trip check f(g(z)) = z on the device (size-independent property)
        zg = net(graph, inverse=False)
        back, _ = net(zg, inverse=True)
        torch.cuda.synchronize()
        out["round_trip_max_abs_err"] = float((back.nodes - graph.nodes).abs().max())
        out["log_prob_xs_per_node"] = None
    if rank == 0 and not multi and not args.no_cpu_baseline and not inverse and trainer is None:
        cb, ref = cpu_baseline(dicts, params)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = round(value / cb["value"], 2)
        out["log_prob_delta_vs_cpu_fp32"] = abs(last["log_prob_xs_per_node"] - ref["log_prob_xs_per_node"])
    if multi:   # how evenly the shards came out (whole graphs per rank, greedy balance): every rank's nodes / edges
        shard = torch.tensor([n_local, e_local], dtype=torch.int64, device=dev if args.dist_backend == "nccl" else "cpu")
        allsh = [torch.zeros_like(shard) for _ in range(world)]
        dist.all_gather(allsh, shard)
        nodes_r, edges_r = [int(t[0]) for t in allsh], [int(t[1]) for t in allsh]
        out["nccl_ranks_seen"] = dist.get_world_size()      # what the process group itself says after init
        out["dist_backend"] = dist.get_backend()
        out["launcher"] = "bench.py itself (one child process per GPU)" if os.environ.get("GNF_BENCH_SELF_LAUNCHED") else "external (torch.distributed.run)"
        out["config"]["nodes_per_rank"] = nodes_r
        out["config"]["edges_per_rank"] = edges_r
        out["config"]["shard_imbalance_max_over_mean"] = {"nodes": round(max(nodes_r) * world / max(1, sum(nodes_r)), 4),
                                                           "edges": round(max(edges_r) * world / max(1, sum(edges_r)), 4)}
    if (rank == 0 and not multi and args.workload == "config2" and not args.layered and not args.no_secondary
            and not args.no_secondary_workloads and not args.sync_each_step):
        import subprocess
        sec, t_sec = {}, time.perf_counter()
        ksteps = args.secondary_steps or min(args.steps, 40)
        torch.cuda.synchronize()
        # forward workloads at --secondary-steps; the two trainer workloads (run_grevnet.py:344-377,440-447 on the config-2 batch;
        # train_grevnet_with_data.py:380,478-554 with that driver's literal defaults) with their own step caps: the
        # whole default run has to stay within a minute or so
        children = [(wl, ksteps) for wl in ("config4", "config5", "config2_attn", "default_flags", "data_default_flags")]
        children += [("config2_train", min(ksteps, 20)), ("data_default_flags_train", min(ksteps, 6))]
        for wl, wsteps in children:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", str(wsteps), "--warmup", str(min(args.warmup, 10 if wsteps > 6 else 2)),
                   "--repeats", "3", "--no-cpu-baseline", "--no-secondary", "--latency-steps", "0"]
            if wl == "data_default_flags_train":
                cmd += ["--prewarm-ms", "100"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                err = None if (r.returncode == 0 and lines) else (r.stderr or r.stdout)[-400:]
            except subprocess.TimeoutExpired:
                err, lines = "timed out after 600 s", []
            if err is None:
                sec[wl] = secondary_entry(json.loads(lines[-1]))
            else:
                sec[wl] = {"error": err}
        out["secondary_workloads"] = sec
        out["secondary_workloads_wall_s"] = round(time.perf_counter() - t_sec, 1)
    out["consistency"] = line_consistency_errors(out)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
