#!/usr/bin/env python3
"""Generate the committed golden fixtures tests/golden/*.npz from the oracle.

The reference itself cannot be run (no TensorFlow 1.x / Sonnet / graph_nets / TFP in this image and
`grevnet.py` is missing from its tree), and it ships no golden vectors, so these fixtures are
outputs of oracle/gnf_oracle.py's float64 dense-adjacency restatement, written only after the
float32 gather restatement agrees with it (asserted below).  "Parity unpinned" at the TF boundary.

Each fixture holds: the batch (n_node, n_edge, senders, receivers with global ids), x, the flow
hyper-parameters, every weight (w_{kind}_{half}_{i}_{layer}, b_...), and the expected z, logdet,
log_prob_zs, log_prob_xs_per_node, plus x_roundtrip = g(z).

Run from the repo root:  python tests/golden/make_golden.py
Inputs: data/*.npz (edge lists converted from the reference's pickles by tools/convert_datasets.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gnf_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def load(name):
    d = np.load(os.path.join(ROOT, "data", name + ".npz"))
    return d["n_node"], d["n_edge"], d["senders"], d["receivers"]


# attention cases: (name, dataset, ids, D, latent, K, T, attention kwargs, weight_sharing, final_scale)
ATTN_CASES = [
    ("attn_cfg1_grid_small", "grid_small", [6, 2], 8, 32, 3, 2,
     dict(num_heads=8, kq_dim=10, v_dim=10, out_dim=80, concat=True, kq_dim_division=False, residual=False), False, 0.5),
    ("attn_small_community_noconcat_div", "community_medium", None, 12, 24, 2, 2,
     dict(num_heads=3, kq_dim=7, v_dim=5, out_dim=20, concat=False, kq_dim_division=True, residual=False), True, 0.4),
    # --attn_layer_norm --attn_residual (run_grevnet.py:80, gnn.py:547-552): non-trivial ln_gamma / ln_beta
    ("attn_layer_norm_residual", "grid_small", [0, 3, 9], 10, 24, 3, 2,
     dict(num_heads=4, kq_dim=6, v_dim=5, out_dim=12, concat=True, kq_dim_division=False, residual=True,
          layer_norm=True), False, 0.5),
]
ONLY = set(sys.argv[1:])      # `make_golden.py name ...` regenerates just those fixtures

CASES = [
    # name, dataset, graph ids, D, latent, K, T, agg, combine, eps, activation, weight_sharing, final_scale
    ("cfg1_grid_small_d2", "grid_small", [6], 2, 16, 3, 1, "mean", "agg", 1.0, "leaky_relu", False, 0.5),
    ("cfg1_grid_small_d8", "grid_small", [6], 8, 32, 5, 1, "mean", "agg", 1.0, "leaky_relu", False, 0.5),
    ("cfg2_small_community", "community_medium", None, 16, 32, 3, 2, "mean", "agg", 1.0, "leaky_relu", False, 0.5),
    ("sum_concat_relu_shared", "grid_small", [0, 6, 7, 11], 6, 24, 4, 3, "sum", "concat", 0.0, "relu", True, 0.3),
]


def main():
    for (name, ds, ids, d, latent, k, t, agg, combine, eps, act, ws, fscale) in CASES:
        if ONLY and name not in ONLY:
            continue
        n_node, n_edge, sl, rl = load(ds)
        rng = np.random.default_rng(12345)             # run_grevnet.py:108
        if ids is None:                                # config 2 draw: with replacement from the 80% train split
            ids = rng.choice(int(0.8 * len(n_node)), size=8, replace=True).tolist()
        nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, ids)
        n = int(nn.sum())
        x = rng.standard_normal((n, d)).astype(np.float32)
        p = O.make_grevnet_params(2024, d // 2, latent, k, t, combine=combine, weight_sharing=ws,
                                  final_scale=fscale)
        o64 = O.Fp64Dense(s, r, n, agg=agg, combine=combine, epsilon=eps, activation=act)
        res = o64.log_prob(x, p, t, ws)
        o32 = O.Fp32Gather(s, r, n, agg=agg, combine=combine, epsilon=eps, activation=act)
        r32 = o32.log_prob(o32.to_t(x), o32.prep_params(p), t, ws)
        assert abs(res["log_prob_xs_per_node"] - r32["log_prob_xs_per_node"]) < 1e-5, name
        assert np.abs(r32["z"].numpy() - res["z"]).max() < 5e-5, name
        xr = o64.g(res["z"], p, t, ws)
        assert np.abs(xr - x).max() < 1e-9
        blob = dict(n_node=nn, n_edge=ne, senders=s, receivers=r, x=x, D=d, latent=latent, K=k, T=t,
                    agg=agg, combine=combine, epsilon=eps, activation=act, weight_sharing=ws,
                    z=res["z"], logdet=res["log_det_jacobian"], log_prob_zs=res["log_prob_zs"],
                    log_prob_xs=res["log_prob_xs"], log_prob_xs_per_node=res["log_prob_xs_per_node"],
                    x_roundtrip=xr)
        for kind in ("s", "t"):
            for half in range(2):
                nets = [p[kind][half]] if ws else p[kind][half]
                for i, mlp in enumerate(nets):
                    for j, (w, b) in enumerate(mlp):
                        blob[f"w_{kind}_{half}_{i}_{j}"] = w
                        blob[f"b_{kind}_{half}_{i}_{j}"] = b
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: N={n} E={len(s)} per-node log-prob={res['log_prob_xs_per_node']:.6f} "
              f"logdet={res['log_det_jacobian']:.4f} -> {os.path.getsize(path)} B")


def main_attn():
    for (name, ds, ids, d, latent, k, t, akw, ws, fscale) in ATTN_CASES:
        if ONLY and name not in ONLY:
            continue
        n_node, n_edge, sl, rl = load(ds)
        rng = np.random.default_rng(12345)
        if ids is None:
            ids = rng.choice(int(0.8 * len(n_node)), size=6, replace=True).tolist()
        nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, ids)
        n = int(nn.sum())
        x = rng.standard_normal((n, d)).astype(np.float32)
        p = O.make_attn_grevnet_params(2025, d // 2, latent, k, t, weight_sharing=ws, final_scale=fscale, **akw)
        o64 = O.Fp64Dense(s, r, n, activation="relu")          # attention GNNs use tf.nn.relu (run_grevnet.py:205)
        res = o64.log_prob(x, p, t, ws)
        o32 = O.Fp32Gather(s, r, n, activation="relu")
        r32 = o32.log_prob(o32.to_t(x), o32.prep_params(p), t, ws)
        assert abs(res["log_prob_xs_per_node"] - r32["log_prob_xs_per_node"]) < 1e-5, name
        assert np.abs(r32["z"].numpy() - res["z"]).max() < 5e-5, name
        xr = o64.g(res["z"], p, t, ws)
        assert np.abs(xr - x).max() < 1e-9
        blob = dict(n_node=nn, n_edge=ne, senders=s, receivers=r, x=x, D=d, latent=latent, K=k, T=t,
                    agg="mean", combine="agg", epsilon=0.0, activation="relu", weight_sharing=ws, gnn="dm_self_attn",
                    z=res["z"], logdet=res["log_det_jacobian"], log_prob_zs=res["log_prob_zs"],
                    log_prob_xs=res["log_prob_xs"], log_prob_xs_per_node=res["log_prob_xs_per_node"],
                    x_roundtrip=xr, **{"attn_" + kk: vv for kk, vv in akw.items()})
        for kind in ("s", "t"):
            for half in range(2):
                nets = [p[kind][half]] if ws else p[kind][half]
                for i, net in enumerate(nets):
                    for key in O.attn_weight_keys(net["attn"]):
                        blob[f"a_{kind}_{half}_{i}_{key}"] = net["attn"][key]
                    for j, (w, b) in enumerate(net["mlp"]):
                        blob[f"w_{kind}_{half}_{i}_{j}"] = w
                        blob[f"b_{kind}_{half}_{i}_{j}"] = b
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(f"{name}: N={n} E={len(s)} per-node log-prob={res['log_prob_xs_per_node']:.6f} "
              f"logdet={res['log_det_jacobian']:.4f} -> {os.path.getsize(path)} B")


def main_bn():
    """use_batch_norm=True (gnn.py:260-263, 310-313): the message-passing flow of cfg2 with a batch-norm
    bijector in front of every half-step (non-trivial gamma / beta / moving statistics).  x_roundtrip = g(z)
    uses the MOVING statistics, so it is NOT x."""
    name, ds, d, latent, k, t = "bn_small_community", "community_medium", 16, 32, 3, 2
    n_node, n_edge, sl, rl = load(ds)
    rng = np.random.default_rng(12345)
    ids = rng.choice(int(0.8 * len(n_node)), size=8, replace=True).tolist()
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, ids)
    n = int(nn.sum())
    x = (rng.standard_normal((n, d)) * 1.7 + 0.4).astype(np.float32)
    p = O.make_grevnet_params(2026, d // 2, latent, k, t, final_scale=0.5)
    p["bn"] = O.make_bn_params(2027, d // 2, t)
    o64 = O.Fp64Dense(s, r, n)
    res = o64.log_prob(x, p, t)
    o32 = O.Fp32Gather(s, r, n)
    r32 = o32.log_prob(o32.to_t(x), o32.prep_params(p), t)
    assert abs(res["log_prob_xs_per_node"] - r32["log_prob_xs_per_node"]) < 2e-5, name
    assert np.abs(r32["z"].numpy() - res["z"]).max() < 5e-5, name
    xr = o64.g(res["z"], p, t)
    blob = dict(n_node=nn, n_edge=ne, senders=s, receivers=r, x=x, D=d, latent=latent, K=k, T=t,
                agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu", weight_sharing=False,
                use_batch_norm=True, z=res["z"], logdet=res["log_det_jacobian"], log_prob_zs=res["log_prob_zs"],
                log_prob_xs=res["log_prob_xs"], log_prob_xs_per_node=res["log_prob_xs_per_node"], x_roundtrip=xr)
    for kind in ("s", "t"):
        for half in range(2):
            for i, mlp in enumerate(p[kind][half]):
                for j, (w, b) in enumerate(mlp):
                    blob[f"w_{kind}_{half}_{i}_{j}"] = w
                    blob[f"b_{kind}_{half}_{i}_{j}"] = b
    for half in range(2):
        for i in range(t):
            for key in ("gamma", "beta", "moving_mean", "moving_variance"):
                blob[f"bn_{half}_{i}_{key}"] = p["bn"][half][i][key]
            m, v = o64.last_bn_moments[(half, i)]
            blob[f"bn_{half}_{i}_batch_mean"] = m
            blob[f"bn_{half}_{i}_batch_variance"] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name}: N={n} E={len(s)} per-node log-prob={res['log_prob_xs_per_node']:.6f} "
          f"logdet={res['log_det_jacobian']:.4f} -> {os.path.getsize(path)} B")


def main_data_driver():
    """The DATA driver's default GNN and flags (train_grevnet_with_data.py:40-46, 100-117, 303-310, 336-343): dm_attn with
    ONE head, kq = v = 64, C = 64, kq_dim_division=True, concat, around a relu MLP; use_batch_norm=True; complete graphs
    incl. self loops in transform_example's edge order (utils.py:164-183).  Widths other than the head geometry are cut
    down (D = 40, latent 48, T = 2) so that the fixture stays small; the literal widths run in tests/test_data_driver_gpu.py."""
    name, d, latent, k, t = "attn_data_driver_defaults", 40, 48, 3, 2
    akw = dict(num_heads=1, kq_dim=64, v_dim=64, out_dim=64, concat=True, kq_dim_division=True, residual=False)
    n_node = load("grid_small")[0][[6, 2, 9, 0]]
    s_l, r_l = [], []
    for n_ in n_node:                                   # sender-major ordered pairs of every graph, global ids
        off = int(sum(len(np.unique(a)) for a in s_l))
        a = np.repeat(np.arange(n_, dtype=np.int32), n_) + off
        b = np.tile(np.arange(n_, dtype=np.int32), n_) + off
        s_l.append(a)
        r_l.append(b)
    s, r = np.concatenate(s_l), np.concatenate(r_l)
    nn, ne = n_node.astype(np.int32), (n_node.astype(np.int64) ** 2).astype(np.int32)
    n = int(nn.sum())
    rng = np.random.default_rng(12345)
    x = (rng.standard_normal((n, d)) * 1.3 + 0.2).astype(np.float32)
    p = O.make_attn_grevnet_params(2028, d // 2, latent, k, t, final_scale=0.4, **akw)
    p["bn"] = O.make_bn_params(2029, d // 2, t)
    o64 = O.Fp64Dense(s, r, n, activation="relu")
    res = o64.log_prob(x, p, t)
    o32 = O.Fp32Gather(s, r, n, activation="relu")
    r32 = o32.log_prob(o32.to_t(x), o32.prep_params(p), t)
    assert abs(res["log_prob_xs_per_node"] - r32["log_prob_xs_per_node"]) < 2e-5, name
    assert np.abs(r32["z"].numpy() - res["z"]).max() < 5e-5, name
    xr = o64.g(res["z"], p, t)                          # (moving statistics: not x)
    blob = dict(n_node=nn, n_edge=ne, senders=s, receivers=r, x=x, D=d, latent=latent, K=k, T=t,
                agg="mean", combine="agg", epsilon=0.0, activation="relu", weight_sharing=False, gnn="dm_self_attn",
                use_batch_norm=True, z=res["z"], logdet=res["log_det_jacobian"], log_prob_zs=res["log_prob_zs"],
                log_prob_xs=res["log_prob_xs"], log_prob_xs_per_node=res["log_prob_xs_per_node"],
                x_roundtrip=xr, **{"attn_" + kk: vv for kk, vv in akw.items()})
    for kind in ("s", "t"):
        for half in range(2):
            for i, net in enumerate(p[kind][half]):
                for key in O.attn_weight_keys(net["attn"]):
                    blob[f"a_{kind}_{half}_{i}_{key}"] = net["attn"][key]
                for j, (w, b) in enumerate(net["mlp"]):
                    blob[f"w_{kind}_{half}_{i}_{j}"] = w
                    blob[f"b_{kind}_{half}_{i}_{j}"] = b
    for half in range(2):
        for i in range(t):
            for key in ("gamma", "beta", "moving_mean", "moving_variance"):
                blob[f"bn_{half}_{i}_{key}"] = p["bn"][half][i][key]
            m, v = o64.last_bn_moments[(half, i)]
            blob[f"bn_{half}_{i}_batch_mean"] = m
            blob[f"bn_{half}_{i}_batch_variance"] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name}: N={n} E={len(s)} per-node log-prob={res['log_prob_xs_per_node']:.6f} "
          f"logdet={res['log_det_jacobian']:.4f} -> {os.path.getsize(path)} B")


if __name__ == "__main__":
    main()
    main_attn()
    if not ONLY or "attn_data_driver_defaults" in ONLY:
        main_data_driver()
    if not ONLY or "bn_small_community" in ONLY:
        main_bn()
