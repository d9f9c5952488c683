"""Shared test helpers: fixtures -> oracle params -> product objects."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = ["cfg1_grid_small_d2", "cfg1_grid_small_d8", "cfg2_small_community", "sum_concat_relu_shared"]
ATTN_GOLDEN_CASES = ["attn_cfg1_grid_small", "attn_small_community_noconcat_div", "attn_layer_norm_residual"]
BN_GOLDEN_CASES = ["bn_small_community"]
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
    """Build the product GRevNet through the reference-shaped factories (run_grevnet.py:154-180).
    hp: dict with D, latent, K, T, agg, combine, epsilon, activation, weight_sharing."""
    from functools import partial
    from gnf_amd import gnn
    act = gnn.leaky_relu if hp["activation"] == "leaky_relu" else gnn.relu
    mk_mlp = partial(gnn.make_mlp_model, hp["latent"], hp["D"] / 2, hp["K"], act, 0.01, 0.1)
    if hp.get("attn"):                      # run_grevnet.py:199-211 make_dm_self_attn_gnn
        a = hp["attn"]
        mk = partial(gnn.dm_self_attn_gnn, kq_dim=a["kq_dim"], v_dim=a["v_dim"], make_mlp_fn=mk_mlp,
                     num_heads=a["num_heads"], concat_heads_output_dim=a["out_dim"], concat=a["concat"],
                     residual=a["residual"], layer_norm=a.get("layer_norm", False), kq_dim_division=a["kq_dim_division"])
    elif hp["combine"] == "concat":
        mk = partial(gnn.sum_concat_then_mlp_gnn if hp["agg"] == "sum" else gnn.avg_concat_then_mlp_gnn, mk_mlp)
    else:
        mk = partial(gnn.sum_then_mlp_gnn if hp["agg"] == "sum" else gnn.avg_then_mlp_gnn, mk_mlp, hp["epsilon"])
    net = gnn.GRevNet(mk, hp["T"], hp["D"], use_batch_norm=bool(params is not None and params.get("bn")),
                      weight_sharing=hp["weight_sharing"])
    if params is not None:
        net.set_params(params)
    return net


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
