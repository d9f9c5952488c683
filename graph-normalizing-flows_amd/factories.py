"""The drivers' make_gnn_fn wiring as one function: hyper-parameter dict -> GRevNet through the reference-shaped
factories, exactly as /root/reference/run_grevnet.py:154-211 (make_mlp_fn partial, the four *_then_mlp_gnn partials,
make_dm_self_attn_gnn) and train_grevnet_with_data.py:272-343 (dm_attn partial, GRevNet(...)) put it together.
bench.py, __graft_entry__.smoke() and the tests build their nets through this."""
from functools import partial

from . import gnn


def make_gnn_fn(hp):
    """hp: D, latent, K, activation ("relu" | "leaky_relu"), then either
         agg ("sum" | "mean"), combine ("agg" | "concat"), epsilon             (run_grevnet.py:154-180)
       or attn = dict(num_heads, kq_dim, v_dim, out_dim, concat, kq_dim_division, residual[, layer_norm])
                                                                                (run_grevnet.py:199-211)."""
    act = gnn.leaky_relu if hp["activation"] == "leaky_relu" else gnn.relu
    mk_mlp = partial(gnn.make_mlp_model, hp["latent"], hp["D"] / 2, hp["K"], act, 0.01, hp.get("bias_init_stddev", 0.1))
    a = hp.get("attn")
    if a:
        return partial(gnn.dm_self_attn_gnn, kq_dim=a["kq_dim"], v_dim=a["v_dim"], make_mlp_fn=mk_mlp,
                       num_heads=a["num_heads"], concat_heads_output_dim=a["out_dim"], concat=a["concat"],
                       residual=a["residual"], layer_norm=a.get("layer_norm", False), kq_dim_division=a["kq_dim_division"])
    if hp["combine"] == "concat":
        return partial(gnn.sum_concat_then_mlp_gnn if hp["agg"] == "sum" else gnn.avg_concat_then_mlp_gnn, mk_mlp)
    return partial(gnn.sum_then_mlp_gnn if hp["agg"] == "sum" else gnn.avg_then_mlp_gnn, mk_mlp, hp["epsilon"])


def make_product_grevnet(hp, params=None):
    """GRevNet(make_gnn_fn, T, D, use_batch_norm, weight_sharing) (run_grevnet.py:277-281) with the weights of `params`
    (the oracle / fixture container layout, GRevNet.set_params) when given; batch-norm bijectors are switched on by
    params["bn"] or hp["use_batch_norm"]."""
    use_bn = bool(params.get("bn")) if params is not None else bool(hp.get("use_batch_norm"))
    net = gnn.GRevNet(make_gnn_fn(hp), hp["T"], hp["D"], use_batch_norm=use_bn, weight_sharing=hp["weight_sharing"])
    if params is not None:
        net.set_params(params)
    return net
