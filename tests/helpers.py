"""Shared test helpers: fixtures -> oracle params -> product objects."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["cfg1_grid_small_d2", "cfg1_grid_small_d8", "cfg2_small_community", "sum_concat_relu_shared"]
ATTN_GOLDEN_CASES = ["attn_cfg1_grid_small", "attn_small_community_noconcat_div", "attn_layer_norm_residual"]
BN_GOLDEN_CASES = ["bn_small_community"]
DATA_DRIVER_GOLDEN_CASES = ["attn_data_driver_defaults"]   # dm_attn 1 head kq = v = 64 + batch norm on complete graphs
ATTN_KEYS = ("num_heads", "kq_dim", "v_dim", "out_dim", "concat", "kq_dim_division", "residual")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: d[k] for k in d.files}
    for k in ("D", "latent", "K", "T"):
        g[k] = int(g[k])
    for k in ("agg", "combine", "activation"):
        g[k] = str(g[k])
    g["epsilon"] = float(g["epsilon"])
    g["weight_sharing"] = bool(g["weight_sharing"])
    ws, t, k = g["weight_sharing"], g["T"], g["K"]
    attn = None
    if "gnn" in g and str(g["gnn"]) == "dm_self_attn":
        attn = {}
        for key in ATTN_KEYS:
            v = g["attn_" + key]
            attn[key] = bool(v) if key in ("concat", "kq_dim_division", "residual") else int(v)
        attn["layer_norm"] = bool(g["attn_layer_norm"]) if "attn_layer_norm" in g else False
        g["attn"] = attn
    params = {}
    for kind in ("s", "t"):
        halves = []
        for half in range(2):
            nets = []
            for i in range(1 if ws else t):
                mlp = [(g[f"w_{kind}_{half}_{i}_{j}"], g[f"b_{kind}_{half}_{i}_{j}"]) for j in range(k)]
                if attn is None:
                    nets.append(mlp)
                else:
                    a = dict(attn)
                    for key in ("wq", "wk", "wv", "wo") + (("ln_gamma", "ln_beta") if attn["layer_norm"] else ()):
                        a[key] = g[f"a_{kind}_{half}_{i}_{key}"]
                    nets.append({"attn": a, "mlp": mlp})
            halves.append(nets[0] if ws else nets)
        params[kind] = halves
    if "use_batch_norm" in g and bool(g["use_batch_norm"]):
        params["bn"] = [[{key: g[f"bn_{half}_{i}_{key}"] for key in ("gamma", "beta", "moving_mean", "moving_variance")}
                         for i in range(t)] for half in range(2)]
        for half in range(2):
            for i in range(t):
                params["bn"][half][i]["epsilon"] = 1e-3
    g["params"] = params
    return g


def make_product_grevnet(hp, params):
    """The drivers' factory wiring (run_grevnet.py:154-211) lives in the package: gnf_amd.factories."""
    from gnf_amd.factories import make_product_grevnet as mk
    return mk(hp, params)


def graph_from_arrays(n_node, n_edge, senders, receivers, x, device="cpu"):
    import torch
    from gnf_amd.graphs import GraphsTuple
    return GraphsTuple(nodes=torch.as_tensor(np.asarray(x, np.float32)).to(device),
                       edges=torch.zeros(len(senders)).to(device),
                       receivers=torch.as_tensor(np.asarray(receivers, np.int32)).to(device),
                       senders=torch.as_tensor(np.asarray(senders, np.int32)).to(device),
                       globals=torch.zeros(len(n_node)).to(device),
                       n_node=torch.as_tensor(np.asarray(n_node, np.int32)).to(device),
                       n_edge=torch.as_tensor(np.asarray(n_edge, np.int32)).to(device))
