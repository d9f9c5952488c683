"""Pins for the oracle itself.  CPU only.

The reference has no tests / golden vectors and cannot be imported, so these analytic known-answer
tests are what anchors oracle/gnf_oracle.py to the maths of /root/reference/gnn.py:304-373.
"""
import math

import numpy as np
import pytest
import torch

from oracle import gnf_oracle as O


def tiny_graph():
    # 5 nodes, ring + chord, directed both ways + self loops (graph_data.py:33-50 shape of data)
    und = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0), (1, 3)]
    s = [i for i in range(5)] + [a for a, b in und] + [b for a, b in und]
    r = [i for i in range(5)] + [b for a, b in und] + [a for a, b in und]
    return np.array(s, np.int32), np.array(r, np.int32), 5


@pytest.mark.parametrize("agg", ["sum", "mean"])
@pytest.mark.parametrize("combine", ["agg", "concat"])
def test_dual_restatement_agreement(grid_small, agg, combine):
    """(1) fp64 dense-adjacency form vs fp32 gather/index_add form."""
    n_node, n_edge, sl, rl = grid_small
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, [6, 0, 3])
    n = int(nn.sum())
    rng = np.random.default_rng(12345)
    for d, latent, k, t in [(2, 16, 3, 1), (8, 32, 5, 3)]:
        x = rng.standard_normal((n, d)).astype(np.float32)
        p = O.make_grevnet_params(7, d // 2, latent, k, t, combine=combine, final_scale=0.5)
        a = O.Fp64Dense(s, r, n, agg=agg, combine=combine)
        b = O.Fp32Gather(s, r, n, agg=agg, combine=combine)
        ra = a.log_prob(x, p, t)
        rb = b.log_prob(b.to_t(x), b.prep_params(p), t)
        assert abs(ra["log_prob_xs_per_node"] - rb["log_prob_xs_per_node"]) < 1e-5
        np.testing.assert_allclose(rb["z"].numpy(), ra["z"], atol=2e-5, rtol=2e-5)


def test_hand_computed_micro_case():
    """(4) 2-node graph, H=1, one Linear layer per net: closed form.
    edges: 0->0, 1->1, 0->1 (node 1 receives from 0 and itself).  sum aggregation, eps=1.
      agg(x)[0] = x[0],  agg(x)[1] = x[0]+x[1];  h = x + agg
      step half 0: s = ws*h(x0)+bs, t = wt*h(x0)+bt ; x1 <- x1*exp(s)+t
    """
    s_idx = np.array([0, 1, 0], np.int32)
    r_idx = np.array([0, 1, 1], np.int32)
    x = np.array([[1.0, 2.0], [3.0, -1.0]])
    ws0, bs0, wt0, bt0 = 0.1, 0.05, -0.3, 0.2
    ws1, bs1, wt1, bt1 = -0.2, 0.0, 0.5, -0.1
    mk = lambda w, b: [(np.array([[w]]), np.array([b]))]
    p = {"s": [[mk(ws0, bs0)], [mk(ws1, bs1)]], "t": [[mk(wt0, bt0)], [mk(wt1, bt1)]]}
    o = O.Fp64Dense(s_idx, r_idx, 2, agg="sum", combine="agg", epsilon=1.0)
    z, ld = o.f(x, p, 1)
    # by hand
    h0 = np.array([1.0 + 1.0, 3.0 + (1.0 + 3.0)])          # x0 = [1,3]
    s = ws0 * h0 + bs0
    t = wt0 * h0 + bt0
    x1 = np.array([2.0, -1.0]) * np.exp(s) + t
    ld_hand = s.sum()
    h1 = np.array([x1[0] + x1[0], x1[1] + (x1[0] + x1[1])])
    s2 = ws1 * h1 + bs1
    t2 = wt1 * h1 + bt1
    x0 = np.array([1.0, 3.0]) * np.exp(s2) + t2
    ld_hand += s2.sum()
    np.testing.assert_allclose(z, np.stack([x0, x1], axis=1), rtol=1e-14)
    assert abs(ld - ld_hand) < 1e-14


@pytest.mark.parametrize("agg,combine", [("mean", "agg"), ("sum", "concat")])
def test_logdet_equals_jacobian_logdet(agg, combine):
    """(2) logdet from f == log|det d vec(z)/d vec(x)| of the whole [N*D]->[N*D] map (autograd, fp64).
    Pins sign and the 'sum over all nodes and features' convention of gnn.py:322,337."""
    s, r, n = tiny_graph()
    d, t = 4, 2
    p = O.make_grevnet_params(3, d // 2, 8, 3, t, combine=combine, final_scale=0.7, dtype=np.float64)
    o = O.Fp32Gather(s, r, n, agg=agg, combine=combine, dtype=torch.float64)
    pt = o.prep_params(p)
    x = torch.as_tensor(np.random.default_rng(0).standard_normal((n, d)))
    z, ld = o.f(x, pt, t)
    jac = torch.autograd.functional.jacobian(lambda v: o.f(v.reshape(n, d), pt, t)[0].reshape(-1),
                                             x.reshape(-1))
    sign, logabs = torch.linalg.slogdet(jac)
    assert abs(float(logabs) - float(ld)) < 1e-9
    # cross-check the fp64 dense formulation on the same input
    z2, ld2 = O.Fp64Dense(s, r, n, agg=agg, combine=combine).f(x.numpy(), p, t)
    assert abs(ld2 - float(ld)) < 1e-10
    np.testing.assert_allclose(z2, z.numpy(), atol=1e-11)


@pytest.mark.parametrize("weight_sharing", [False, True])
def test_round_trip(grid_small, weight_sharing):
    """(3) g(f(x)) = x and f(g(z)) = z (gnn.py:343-373 undoes gnn.py:304-341 in reverse order)."""
    n_node, n_edge, sl, rl = grid_small
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, [6, 7])
    n = int(nn.sum())
    d, t = 8, 4
    p = O.make_grevnet_params(11, d // 2, 32, 4, t, weight_sharing=weight_sharing, final_scale=0.5)
    x = np.random.default_rng(1).standard_normal((n, d))
    o = O.Fp64Dense(s, r, n)
    z, _ = o.f(x, p, t, weight_sharing)
    np.testing.assert_allclose(o.g(z, p, t, weight_sharing), x, atol=1e-11)
    np.testing.assert_allclose(o.f(o.g(x, p, t, weight_sharing), p, t, weight_sharing)[0], x, atol=1e-11)
    o32 = O.Fp32Gather(s, r, n)
    p32 = o32.prep_params(p)
    x32 = o32.to_t(x)
    z32, _ = o32.f(x32, p32, t, weight_sharing)
    assert float((o32.g(z32, p32, t, weight_sharing) - x32).abs().max()) < 1e-4


def test_additivity_and_permutation(community_medium):
    """(5) log-prob of a batch = sum over single-graph runs (block-diagonal batching: the property
    multi-GPU sharding relies on); relabelling nodes leaves it unchanged."""
    n_node, n_edge, sl, rl = community_medium
    ids = [3, 50, 120]
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, ids)
    n = int(nn.sum())
    d, t = 8, 2
    p = O.make_grevnet_params(5, d // 2, 16, 3, t, final_scale=0.5)
    x = np.random.default_rng(2).standard_normal((n, d))
    full = O.Fp64Dense(s, r, n).log_prob(x, p, t)
    acc, off = 0.0, 0
    for gid in ids:
        n1, e1, s1, r1 = O.batch_graphs(n_node, n_edge, sl, rl, [gid])
        k = int(n1.sum())
        acc += O.Fp64Dense(s1, r1, k).log_prob(x[off:off + k], p, t)["log_prob_xs"]
        off += k
    assert abs(acc - full["log_prob_xs"]) < 1e-8
    perm = np.random.default_rng(3).permutation(n)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)          # old id v -> new id inv[v]; new node j holds old node perm[j]
    pr = O.Fp64Dense(inv[s], inv[r], n).log_prob(x[perm], p, t)
    assert abs(pr["log_prob_xs"] - full["log_prob_xs"]) < 1e-8
    np.testing.assert_allclose(pr["z"], full["z"][perm], atol=1e-11)


def test_aggregator_semantics():
    """(6) mean = sum / in-degree; sum on a fully connected graph (+self loops) = per-graph column sum
    (grevnet_synthetic_data.py:17-21, utils.py:164-183)."""
    n = 6
    s = np.repeat(np.arange(n), n).astype(np.int32)   # sender-major all ordered pairs incl. self
    r = np.tile(np.arange(n), n).astype(np.int32)
    x = np.random.default_rng(4).standard_normal((n, 3))
    ident = [(np.eye(3), np.zeros(3))]
    o_sum = O.Fp64Dense(s, r, n, agg="sum", epsilon=0.0)
    np.testing.assert_allclose(o_sum.gnn(x, ident), np.tile(x.sum(0), (n, 1)), atol=1e-13)
    o_mean = O.Fp64Dense(s, r, n, agg="mean", epsilon=0.0)
    np.testing.assert_allclose(o_mean.gnn(x, ident), np.tile(x.mean(0), (n, 1)), atol=1e-13)
    # isolated node (no incoming edge): empty segment -> 0 for both sum and mean (max(count,1))
    o_iso = O.Fp64Dense(np.array([0], np.int32), np.array([0], np.int32), 2, agg="mean", epsilon=0.0)
    out = o_iso.gnn(np.array([[2.0], [5.0]]), [(np.eye(1), np.zeros(1))])
    np.testing.assert_allclose(out, [[2.0], [0.0]])
    o_iso32 = O.Fp32Gather(np.array([0], np.int32), np.array([0], np.int32), 2, agg="mean", epsilon=0.0)
    out32 = o_iso32.gnn(o_iso32.to_t([[2.0], [5.0]]), o_iso32.prep_params({"m": [(np.eye(1), np.zeros(1))]})["m"])
    np.testing.assert_allclose(out32.numpy(), [[2.0], [0.0]])


def test_gaussian_term_vs_scipy():
    """(7) -0.5*sum z^2 - D/2 ln 2pi  vs scipy.stats.multivariate_normal."""
    from scipy.stats import multivariate_normal
    z = np.random.default_rng(5).standard_normal((7, 6))
    ref = multivariate_normal(mean=np.zeros(6), cov=np.eye(6)).logpdf(z).sum()
    assert abs(O.gaussian_log_prob_sum(z) - ref) < 1e-10


def test_activation_semantics():
    o = O.Fp64Dense(np.array([0], np.int32), np.array([0], np.int32), 1)
    np.testing.assert_allclose(o.act(np.array([-2.0, 0.0, 3.0])), [-0.4, 0.0, 3.0])
    o.activation = "relu"
    np.testing.assert_allclose(o.act(np.array([-2.0, 0.0, 3.0])), [0.0, 0.0, 3.0])


# ------------------------------------------------------------------------------------------------
# edge-list attention GNN (DMSelfAttentionMLP, gnn.py:385-553)
# ------------------------------------------------------------------------------------------------
def test_attention_dual_restatement_agreement(grid_small):
    n_node, n_edge, sl, rl = grid_small
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, [6, 0, 3])
    n = int(nn.sum())
    rng = np.random.default_rng(7)
    for concat, div, res in [(True, False, False), (False, True, True)]:
        d, t = 8, (1 if res else 2)      # residual feeds x straight into s: keep that flow shallow
        p = O.make_attn_grevnet_params(3, d // 2, 24, 3, t, num_heads=4, kq_dim=5, v_dim=6, out_dim=12,
                                       concat=concat, kq_dim_division=div, residual=res, final_scale=0.5)
        x = (rng.standard_normal((n, d)) * (0.3 if res else 1.0)).astype(np.float32)
        a = O.Fp64Dense(s, r, n, activation="relu")
        b = O.Fp32Gather(s, r, n, activation="relu")
        ra = a.log_prob(x, p, t)
        rb = b.log_prob(b.to_t(x), b.prep_params(p), t)
        assert abs(ra["log_prob_xs_per_node"] - rb["log_prob_xs_per_node"]) < 1e-5
        np.testing.assert_allclose(rb["z"].numpy(), ra["z"], atol=3e-5, rtol=3e-5)
        np.testing.assert_allclose(a.g(ra["z"], p, t), x, atol=1e-9)


def test_attention_reduces_to_mean_aggregation_when_logits_are_constant():
    """Wq = 0 -> every logit is 0 -> softmax weights = 1/in-degree -> the attended value is the MEAN of
    the senders' values: with Wv = I, one head and Wo = I the block equals avg_concat_then_mlp."""
    s, r, n = tiny_graph()
    h = 3
    x = np.random.default_rng(0).standard_normal((n, h))
    mlp = O.make_mlp_params(np.random.default_rng(1), 2 * h, 8, h, 2, dtype=np.float64)
    net = {"attn": {"num_heads": 1, "kq_dim": 2, "v_dim": h, "concat": True, "kq_dim_division": False,
                    "residual": False, "wq": np.zeros((h, 2)), "wk": np.ones((h, 2)), "wv": np.eye(h),
                    "wo": np.eye(h)}, "mlp": mlp}
    got = O.Fp64Dense(s, r, n, activation="relu").gnn(x, net)
    want = O.Fp64Dense(s, r, n, agg="mean", combine="concat", activation="relu").gnn(x, mlp)
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_attention_hand_computed_two_edges():
    """node 1 receives from node 0 and itself; one head, kq = 1: weights = softmax([q0*k1, q1*k1])."""
    s_idx = np.array([0, 1, 0], np.int32)
    r_idx = np.array([0, 1, 1], np.int32)
    x = np.array([[2.0], [-1.0]])
    net = {"attn": {"num_heads": 1, "kq_dim": 1, "v_dim": 1, "concat": False, "kq_dim_division": False,
                    "residual": False, "wq": np.array([[0.5]]), "wk": np.array([[2.0]]),
                    "wv": np.array([[3.0]]), "wo": np.array([[1.0]])},
           "mlp": [(np.array([[1.0]]), np.array([0.0]))]}
    out = O.Fp64Dense(s_idx, r_idx, 2, activation="relu").gnn(x, net)
    q, k, v = 0.5 * x[:, 0], 2.0 * x[:, 0], 3.0 * x[:, 0]
    l0, l1 = q[0] * k[1], q[1] * k[1]                  # sender 0 -> 1, sender 1 -> 1
    w0, w1 = np.exp(l0) / (np.exp(l0) + np.exp(l1)), np.exp(l1) / (np.exp(l0) + np.exp(l1))
    np.testing.assert_allclose(out[:, 0], [v[0], w0 * v[0] + w1 * v[1]], rtol=1e-13)


def test_attention_logdet_equals_jacobian_logdet():
    s, r, n = tiny_graph()
    d, t = 4, 2
    p = O.make_attn_grevnet_params(5, d // 2, 8, 2, t, num_heads=2, kq_dim=3, v_dim=2, out_dim=4,
                                   final_scale=0.6, dtype=np.float64)
    o = O.Fp32Gather(s, r, n, activation="relu", dtype=torch.float64)
    pt = o.prep_params(p)
    x = torch.as_tensor(np.random.default_rng(2).standard_normal((n, d)))
    z, ld = o.f(x, pt, t)
    jac = torch.autograd.functional.jacobian(lambda v: o.f(v.reshape(n, d), pt, t)[0].reshape(-1), x.reshape(-1))
    assert abs(float(torch.linalg.slogdet(jac)[1]) - float(ld)) < 1e-9


def test_pred_adj_blocks_against_dense_masked_formula():
    """Restatement check: the per-graph blocks equal the reference's dense construction
    (distance on ALL nodes, times the block-diagonal mask, diagonal removed)."""
    rng = np.random.default_rng(0)
    n_node = [3, 1, 4]
    z = rng.standard_normal((8, 6)) * 0.5
    blocks = O.pred_adj_blocks(z, n_node)
    r = (z * z).sum(1, keepdims=True)
    dense = 1.0 / (1.0 + np.exp(-10.0 * (1.0 - (r - 2 * z @ z.T + r.T) / np.sqrt(6))))
    mask = np.zeros((8, 8))
    off = 0
    for n in n_node:
        mask[off:off + n, off:off + n] = 1.0
        off += n
    dense = dense * mask * (1.0 - np.eye(8))
    off = 0
    for b, n in zip(blocks, n_node):
        np.testing.assert_allclose(b, dense[off:off + n, off:off + n], atol=1e-12)
        off += n
    assert blocks[1].shape == (1, 1) and blocks[1][0, 0] == 0.0
    # symmetric, in (0, 1), identical points -> sigmoid(10) off the diagonal
    assert np.allclose(blocks[2], blocks[2].T)
    same = O.pred_adj_blocks(np.ones((2, 4)), [2])[0]
    assert abs(same[0, 1] - 1.0 / (1.0 + np.exp(-10.0))) < 1e-15


# ---- training-step oracle (run_grevnet.py:291-295, 340-377) -----------------------------------------
def _params_f64(p):
    if isinstance(p, tuple):
        return tuple(np.asarray(a, np.float64) for a in p)
    if isinstance(p, dict):
        return {k: _params_f64(v) for k, v in p.items()}
    return [_params_f64(q) for q in p]


@pytest.mark.parametrize("agg,combine,activation", [("mean", "agg", "leaky_relu"), ("sum", "concat", "relu")])
def test_gradient_oracle_matches_finite_differences(agg, combine, activation):
    """autograd of the torch restatement vs central differences of the independent numpy fp64-dense form."""
    import copy
    s, r, n = tiny_graph()
    t = 2
    p = _params_f64(O.make_grevnet_params(3, 2, 6, 3, t, combine=combine, final_scale=0.5))
    x = np.random.default_rng(0).standard_normal((n, 4))
    kw = dict(agg=agg, combine=combine, epsilon=0.7, activation=activation)
    res = O.loss_and_grads(s, r, n, x, p, t, **kw)
    dense = O.Fp64Dense(s, r, n, **kw)

    def loss(pp):
        return -dense.log_prob(x, pp, t)["log_prob_xs"]

    assert abs(res["total_loss"] - loss(p)) < 1e-9
    eps = 1e-6
    for (kind, half, i, j, which) in [("s", 1, 0, 1, 0), ("t", 0, 1, 0, 0), ("s", 0, 0, 2, 1), ("t", 1, 1, 2, 0)]:
        w = p[kind][half][i][j][which]
        ga = res["grads"][kind][half][i][j][which]
        it = np.nditer(w, flags=["multi_index"])
        for _ in it:
            idx = it.multi_index
            pp = copy.deepcopy(p)
            pp[kind][half][i][j][which][idx] += eps
            lp = loss(pp)
            pp[kind][half][i][j][which][idx] -= 2 * eps
            lm = loss(pp)
            assert abs((lp - lm) / (2 * eps) - ga[idx]) < 1e-5 * max(1.0, abs(ga[idx]))


def test_gradient_oracle_weight_sharing_sums_uses():
    """With weight sharing a net is used at every timestep; its gradient is the sum over the uses: equal to
    the sum of the per-timestep gradients of the un-shared flow built from the same weights."""
    s, r, n = tiny_graph()
    t = 3
    shared = O.make_grevnet_params(5, 2, 5, 2, t, weight_sharing=True, final_scale=0.5)
    unshared = {k: [[shared[k][0]] * t, [shared[k][1]] * t] for k in ("s", "t")}
    x = np.random.default_rng(1).standard_normal((n, 4))
    a = O.loss_and_grads(s, r, n, x, shared, t, weight_sharing=True)
    b = O.loss_and_grads(s, r, n, x, unshared, t, weight_sharing=False)
    assert abs(a["total_loss"] - b["total_loss"]) < 1e-10
    for kind in ("s", "t"):
        for half in range(2):
            for j in range(2):
                for which in range(2):
                    tot = sum(b["grads"][kind][half][i][j][which] for i in range(t))
                    np.testing.assert_allclose(a["grads"][kind][half][j][which], tot, atol=1e-10)


def test_adam_first_step_closed_form():
    """t = 1 from zero moments: w1 = w0 - lr * g / (|g| + eps / sqrt(1 - beta2))."""
    rng = np.random.default_rng(2)
    w, g = rng.standard_normal(50), rng.standard_normal(50)
    lr, b1, b2, eps = 1e-3, 0.9, 0.9, 1e-8
    w1, m1, v1 = O.adam_step(w, g, np.zeros(50), np.zeros(50), 1, lr, b1, b2, eps)
    np.testing.assert_allclose(w1, w - lr * g / (np.abs(g) + eps / math.sqrt(1 - b2)), rtol=1e-12)
    np.testing.assert_allclose(m1, (1 - b1) * g)
    np.testing.assert_allclose(v1, (1 - b2) * g * g)


def test_clippers():
    g = np.array([-3.0, 0.5, 7.0])
    np.testing.assert_allclose(O.clip_by_value(g, -1.0, 5.0), [-1.0, 0.5, 5.0])
    np.testing.assert_allclose(O.clip_by_norm(g, 100.0), g)                       # below the norm: untouched
    np.testing.assert_allclose(np.linalg.norm(O.clip_by_norm(g, 2.0)), 2.0)


# ---- batch-norm bijector inside the flow (gnn.py:260-263, 310-313, 356-358) -------------------------------
def test_bn_bijector_hand_computed():
    """Two nodes, one feature: x = (0, 2) -> mean 1, biased var 1; y = (x - 1)/sqrt(1 + eps) * gamma + beta;
    ildj(event_ndims=2) = N * (log gamma - 0.5 log(1 + eps))."""
    bn = {"gamma": np.array([1.5]), "beta": np.array([0.25]), "moving_mean": np.array([3.0]),
          "moving_variance": np.array([4.0]), "epsilon": 1e-3}
    x = np.array([[0.0], [2.0]])
    y, ildj, mean, var = O.Fp64Dense.bn_inverse(x, bn)
    np.testing.assert_allclose(mean, [1.0])
    np.testing.assert_allclose(var, [1.0])
    np.testing.assert_allclose(y[:, 0], np.array([-1.0, 1.0]) / math.sqrt(1.001) * 1.5 + 0.25, rtol=1e-14)
    assert abs(ildj - 2 * (math.log(1.5) - 0.5 * math.log(1.001))) < 1e-14
    # bn.forward uses the MOVING statistics: (z - beta) / gamma * sqrt(4 + eps) + 3
    z = np.array([[0.25], [1.75]])
    np.testing.assert_allclose(O.Fp64Dense.bn_forward(z, bn)[:, 0], np.array([0.0, 1.0]) * math.sqrt(4.001) + 3.0)


def test_bn_flow_dual_agreement_and_round_trip(grid_small):
    n_node, n_edge, sl, rl = grid_small
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, [6, 0, 3])
    n = int(nn.sum())
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((n, 8)) * 2 - 1).astype(np.float32)
    t = 2
    p = O.make_grevnet_params(11, 4, 16, 3, t, final_scale=0.5)
    p["bn"] = O.make_bn_params(12, 4, t)
    o64 = O.Fp64Dense(s, r, n)
    a = o64.log_prob(x, p, t)
    o32 = O.Fp32Gather(s, r, n)
    b = o32.log_prob(o32.to_t(x), o32.prep_params(p), t)
    assert abs(a["log_prob_xs_per_node"] - b["log_prob_xs_per_node"]) < 2e-5
    np.testing.assert_allclose(b["z"].numpy(), a["z"], atol=5e-5)
    # the two directions are inverses of each other exactly when the moving statistics equal the batch moments
    q = {k: v for k, v in p.items()}
    q["bn"] = [[dict(p["bn"][h][i], moving_mean=o64.last_bn_moments[(h, i)][0],
                     moving_variance=o64.last_bn_moments[(h, i)][1]) for i in range(t)] for h in range(2)]
    np.testing.assert_allclose(o64.g(a["z"], q, t), x, atol=1e-9)
    assert np.abs(o64.g(a["z"], p, t) - x).max() > 1e-2      # ... and not with other moving statistics
    # with use_batch_norm the log-det has the N * sum(log gamma - 0.5 log(var + eps)) terms on top of sum(s)
    plain = {k: v for k, v in p.items() if k != "bn"}
    assert abs(a["log_det_jacobian"] - o64.log_prob(x, plain, t)["log_det_jacobian"]) > 1.0


def test_gradient_oracle_with_batch_norm_matches_finite_differences():
    """tf.gradients differentiates through the batch moments and the bijector's log-det term; so does autograd
    over the torch restatement; central differences of the numpy dense form agree."""
    import copy
    s, r, n = tiny_graph()
    t = 2
    p = _params_f64(O.make_grevnet_params(3, 2, 6, 3, t, final_scale=0.5))
    p["bn"] = [[{k: (np.asarray(v, np.float64) if k != "epsilon" else v) for k, v in b.items()} for b in half]
               for half in O.make_bn_params(4, 2, t)]
    x = np.random.default_rng(0).standard_normal((n, 4)) * 2 + 1
    res = O.loss_and_grads(s, r, n, x, p, t)
    dense = O.Fp64Dense(s, r, n)

    def loss(pp):
        return -dense.log_prob(x, pp, t)["log_prob_xs"]

    assert abs(res["total_loss"] - loss(p)) < 1e-9
    eps = 1e-6
    for (half, i, key) in [(0, 0, "gamma"), (1, 1, "beta"), (1, 0, "gamma"), (0, 1, "beta")]:
        for c in range(2):
            pp = copy.deepcopy(p)
            pp["bn"][half][i][key][c] += eps
            lp = loss(pp)
            pp["bn"][half][i][key][c] -= 2 * eps
            lm = loss(pp)
            assert abs((lp - lm) / (2 * eps) - res["grads"]["bn"][half][i][key][c]) < 1e-6
    for (kind, half, i, j) in [("s", 0, 0, 0), ("t", 1, 1, 2)]:       # MLP weights upstream of a bijector
        w = p[kind][half][i][j][0]
        for idx in [(0, 1), (1, 0)]:
            pp = copy.deepcopy(p)
            pp[kind][half][i][j][0][idx] += eps
            lp = loss(pp)
            pp[kind][half][i][j][0][idx] -= 2 * eps
            lm = loss(pp)
            assert abs((lp - lm) / (2 * eps) - res["grads"][kind][half][i][j][0][idx]) < 1e-6


def test_gradient_oracle_attention_matches_finite_differences():
    """Attention nets: wq, wk, wv, wo gradients (autograd of the edge-list restatement) vs central differences of
    the dense masked-softmax restatement; a duplicated edge is in the graph on purpose."""
    import copy
    s = np.array([0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1, 1], np.int32)
    r = np.array([0, 1, 2, 3, 4, 1, 2, 3, 4, 0, 2, 3, 3], np.int32)
    n, t = 5, 2
    raw = O.make_attn_grevnet_params(7, 2, 6, 2, t, num_heads=3, kq_dim=4, v_dim=3, out_dim=5, final_scale=0.5)

    def f64(p):
        if isinstance(p, tuple):
            return tuple(np.asarray(a, np.float64) for a in p)
        if isinstance(p, dict):
            return {k: (f64(v) if isinstance(v, (list, tuple, dict, np.ndarray)) else v) for k, v in p.items()}
        if isinstance(p, np.ndarray):
            return p.astype(np.float64)
        return [f64(q) for q in p]

    p = f64(raw)
    x = np.random.default_rng(0).standard_normal((n, 4))
    res = O.loss_and_grads(s, r, n, x, p, t, activation="relu")
    dense = O.Fp64Dense(s, r, n, activation="relu")

    def loss(pp):
        return -dense.log_prob(x, pp, t)["log_prob_xs"]

    assert abs(res["total_loss"] - loss(p)) < 1e-9
    eps = 1e-6
    for (kind, half, i) in [("s", 1, 0), ("t", 0, 1)]:
        for key in ("wq", "wk", "wv", "wo"):
            w = p[kind][half][i]["attn"][key]
            for idx in [(0, 0), (1, 2), (w.shape[0] - 1, w.shape[1] - 1)]:
                pp = copy.deepcopy(p)
                pp[kind][half][i]["attn"][key][idx] += eps
                lp = loss(pp)
                pp[kind][half][i]["attn"][key][idx] -= 2 * eps
                lm = loss(pp)
                assert abs((lp - lm) / (2 * eps) - res["grads"][kind][half][i]["attn"][key][idx]) < 1e-6


# ------------------------------------------------------------------------------------------------
# DMSelfAttentionMLP(layer_norm=True): snt.LayerNorm over the block output (gnn.py:550-552)
# ------------------------------------------------------------------------------------------------
def test_layer_norm_known_answers():
    """rows come out with mean beta-weighted / unit biased variance; a constant row maps to beta exactly
    (var = 0 -> (h - mean) = 0); eps = 1e-5 sits inside the square root."""
    h = np.array([[1.0, 2.0, 3.0, 6.0], [5.0, 5.0, 5.0, 5.0], [-1.0, 1.0, -1.0, 1.0]])
    one, zero = np.ones(4), np.zeros(4)
    y = O.layer_norm_rows(h, one, zero)
    np.testing.assert_allclose(y.mean(axis=1), 0.0, atol=1e-15)
    np.testing.assert_allclose(y[1], 0.0, atol=0)
    np.testing.assert_allclose(y[2], np.array([-1.0, 1.0, -1.0, 1.0]) / np.sqrt(1.0 + 1e-5), rtol=1e-15)
    var0 = ((h[0] - 3.0) ** 2).mean()                      # 3.5: biased (population) variance, tf.nn.moments
    np.testing.assert_allclose(y[0], (h[0] - 3.0) / np.sqrt(var0 + 1e-5), rtol=1e-15)
    g, b = np.array([2.0, 0.5, 1.0, -1.0]), np.array([0.1, 0.2, 0.3, 0.4])
    np.testing.assert_allclose(O.layer_norm_rows(h, g, b), y * g + b, rtol=1e-15)


def test_layer_norm_attention_dual_agreement_round_trip_and_jacobian(grid_small):
    n_node, n_edge, sl, rl = grid_small
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, [6, 0, 3])
    n = int(nn.sum())
    rng = np.random.default_rng(11)
    for concat, res, ws in [(True, True, False), (False, False, True)]:
        d, t = 12, 2
        p = O.make_attn_grevnet_params(4, d // 2, 24, 3, t, weight_sharing=ws, num_heads=4, kq_dim=5, v_dim=6, out_dim=12,
                                       concat=concat, residual=res, layer_norm=True, final_scale=0.5)
        assert p["s"][0]["attn"]["layer_norm"] if ws else p["s"][0][0]["attn"]["layer_norm"]
        x = rng.standard_normal((n, d)).astype(np.float32)
        a = O.Fp64Dense(s, r, n, activation="relu")
        b = O.Fp32Gather(s, r, n, activation="relu")
        ra = a.log_prob(x, p, t, ws)
        rb = b.log_prob(b.to_t(x), b.prep_params(p), t, ws)
        assert abs(ra["log_prob_xs_per_node"] - rb["log_prob_xs_per_node"]) < 1e-5
        np.testing.assert_allclose(rb["z"].numpy(), ra["z"], atol=3e-5, rtol=3e-5)
        np.testing.assert_allclose(a.g(ra["z"], p, t, ws), x, atol=1e-9)
    # log-det = log |det J| of the flow with the normalisation inside the nets
    s, r, n = tiny_graph()
    d, t = 4, 2
    p = O.make_attn_grevnet_params(5, d // 2, 8, 2, t, num_heads=2, kq_dim=3, v_dim=2, out_dim=4, layer_norm=True,
                                   residual=True, final_scale=0.6, dtype=np.float64)
    o = O.Fp32Gather(s, r, n, activation="relu", dtype=torch.float64)
    pt = o.prep_params(p)
    x = torch.as_tensor(np.random.default_rng(2).standard_normal((n, d)))
    z, ld = o.f(x, pt, t)
    jac = torch.autograd.functional.jacobian(lambda v: o.f(v.reshape(n, d), pt, t)[0].reshape(-1), x.reshape(-1))
    assert abs(float(torch.linalg.slogdet(jac)[1]) - float(ld)) < 1e-9


def test_gradient_oracle_layer_norm_matches_finite_differences():
    """ln_gamma / ln_beta (and a weight in front of the normalisation) vs central differences of the dense restatement."""
    import copy
    s = np.array([0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 0, 1, 1], np.int32)
    r = np.array([0, 1, 2, 3, 4, 1, 2, 3, 4, 0, 2, 3, 3], np.int32)
    n, t, hd = 5, 2, 3
    raw = O.make_attn_grevnet_params(9, hd, 6, 2, t, num_heads=3, kq_dim=4, v_dim=3, out_dim=5, final_scale=0.5,
                                     layer_norm=True, residual=True, dtype=np.float64)
    x = np.random.default_rng(0).standard_normal((n, 2 * hd))
    res = O.loss_and_grads(s, r, n, x, raw, t, activation="relu")
    dense = O.Fp64Dense(s, r, n, activation="relu")

    def loss(pp):
        return -dense.log_prob(x, pp, t)["log_prob_xs"]

    assert abs(res["total_loss"] - loss(raw)) < 1e-9
    eps = 1e-6
    for (kind, half, i) in [("s", 1, 0), ("t", 0, 1)]:
        for key, idxs in (("ln_gamma", [(0,), (2,)]), ("ln_beta", [(1,)]), ("wo", [(1, 2)]), ("wq", [(0, 0)])):
            for idx in idxs:
                pp = copy.deepcopy(raw)
                pp[kind][half][i]["attn"][key][idx] += eps
                lp = loss(pp)
                pp[kind][half][i]["attn"][key][idx] -= 2 * eps
                lm = loss(pp)
                assert abs((lp - lm) / (2 * eps) - res["grads"][kind][half][i]["attn"][key][idx]) < 1e-6
        w = raw[kind][half][i]["mlp"][-1][0]
        pp = copy.deepcopy(raw)
        pp[kind][half][i]["mlp"][-1][0][0, 1] += eps
        lp = loss(pp)
        pp[kind][half][i]["mlp"][-1][0][0, 1] -= 2 * eps
        lm = loss(pp)
        assert abs((lp - lm) / (2 * eps) - res["grads"][kind][half][i]["mlp"][-1][0][0, 1]) < 1e-6


# ------------------------------------------------------------------------------------------------
# Independent checks of the restated third-party semantics against implementations that ARE installed
# (torch's own layer_norm / batch_norm / softmax / leaky_relu / Normal).  The reference's TensorFlow stack is
# absent, so parity stays "unpinned"; these shrink the risk that the oracle, the fixtures generated from it
# and the kernels all share one misreading of the formula (DESIGN.md section 12 names the upstream files).
# ------------------------------------------------------------------------------------------------
def test_layer_norm_rows_vs_torch_layer_norm():
    """snt.LayerNorm (gnn.py:550-552): per-row moments over features, biased variance, eps 1e-5, gamma / beta."""
    rng = np.random.default_rng(3)
    for n, h in ((7, 5), (3, 1), (12, 150)):
        x = rng.standard_normal((n, h)) * 3.0 + 1.5
        gamma, beta = rng.standard_normal(h) + 1.0, rng.standard_normal(h)
        want = torch.nn.functional.layer_norm(torch.as_tensor(x), (h,), torch.as_tensor(gamma), torch.as_tensor(beta), eps=1e-5)
        np.testing.assert_allclose(O.layer_norm_rows(x, gamma, beta), want.numpy(), rtol=1e-12, atol=1e-12)
    # H = 1: variance 0 -> output is beta (the kernels' H = 1 test relies on it)
    np.testing.assert_allclose(O.layer_norm_rows(np.array([[4.2]]), np.array([2.0]), np.array([-0.3])), [[-0.3]], atol=1e-12)


def test_batch_norm_bijector_vs_torch_batch_norm_training_mode():
    """tfb.BatchNormalization.inverse in training mode (gnn.py:260-263, 310-313): batch moments over the node axis with
    the BIASED variance, eps 1e-3 (tf.layers.BatchNormalization default), scale gamma / offset beta - what
    torch.nn.functional.batch_norm(training=True) computes; the moments it reports; and the log-det term
    N * sum_f(log gamma_f - 0.5 log(var_f + eps)) against the Jacobian of torch's own function."""
    rng = np.random.default_rng(4)
    n, h = 11, 6
    x = rng.standard_normal((n, h)) * 2.0 + 0.7
    bn = dict(gamma=np.abs(rng.standard_normal(h)) + 0.5, beta=rng.standard_normal(h), epsilon=1e-3,
              moving_mean=np.zeros(h), moving_variance=np.ones(h))
    y, ildj, mean, var = O.Fp64Dense.bn_inverse(x, bn)
    rm, rv = torch.zeros(h, dtype=torch.float64), torch.ones(h, dtype=torch.float64)
    want = torch.nn.functional.batch_norm(torch.as_tensor(x), rm, rv, torch.as_tensor(bn["gamma"]), torch.as_tensor(bn["beta"]),
                                          training=True, momentum=1.0, eps=1e-3)
    np.testing.assert_allclose(y, want.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mean, rm.numpy(), rtol=1e-12, atol=1e-12)            # momentum 1: running mean = batch mean
    np.testing.assert_allclose(var * n / (n - 1), rv.numpy(), rtol=1e-12)           # torch's running var is the UNBIASED one
    # log |det d y / d x| of the per-element affine map with the batch statistics held fixed (what TFP's
    # inverse_log_det_jacobian reports: the statistics are not differentiated), summed over all N * H elements
    scale = torch.as_tensor(bn["gamma"]) / torch.sqrt(torch.as_tensor(var) + 1e-3)
    assert abs(ildj - n * float(torch.log(scale).sum())) < 1e-10
    # fp32 restatement in TF op order agrees
    f32 = O.Fp32Gather(np.zeros(1, np.int32), np.zeros(1, np.int32), 1)
    bn_t = {k: torch.as_tensor(np.asarray(v, np.float32)) if k != "epsilon" else v for k, v in bn.items()}
    y32, ildj32 = f32.bn_inverse(torch.as_tensor(x.astype(np.float32)), bn_t)
    np.testing.assert_allclose(y32.numpy(), y, rtol=2e-5, atol=2e-5)
    assert abs(float(ildj32) - ildj) < 1e-3
    # bn.forward is the inverse map with the moving statistics in place of the batch's
    bn2 = dict(bn, moving_mean=mean, moving_variance=var)
    np.testing.assert_allclose(O.Fp64Dense.bn_forward(y, bn2), x, rtol=1e-10, atol=1e-10)


def test_segment_softmax_vs_per_receiver_torch_softmax():
    """graph_nets _unsorted_segment_softmax (gnn.py:413, 462-464): per head, softmax over the incoming edges of each
    receiver; duplicate edges count twice; a receiver without incoming edges contributes nothing.  Checked edge by edge
    against torch.softmax over each receiver's own logits, through both restatements' attention blocks."""
    rng = np.random.default_rng(5)
    n = 6
    s = np.array([0, 1, 2, 2, 3, 0, 1, 1, 4, 2], np.int32)      # duplicate edge 1 -> 3; node 5 has no incoming edge
    r = np.array([0, 0, 0, 1, 1, 2, 3, 3, 4, 4], np.int32)
    nh, kq, vd, c, hd = 2, 3, 2, 4, 3
    a = dict(num_heads=nh, kq_dim=kq, v_dim=vd, out_dim=c, concat=False, kq_dim_division=True, residual=False,
             wq=rng.standard_normal((hd, nh * kq)), wk=rng.standard_normal((hd, nh * kq)), wv=rng.standard_normal((hd, vd)),
             wo=np.eye(nh * vd, c))                               # identity-like output projection: agg readable in `new`
    net = {"attn": a, "mlp": [(np.eye(c), np.zeros(c))]}          # identity MLP: the block returns `new`
    x = rng.standard_normal((n, hd))
    q = (x @ a["wq"]).reshape(n, nh, kq)
    k = (x @ a["wk"]).reshape(n, nh, kq)
    v = x @ a["wv"]
    want = np.zeros((n, nh, vd))
    for recv in range(n):
        es = np.nonzero(r == recv)[0]
        if es.size == 0:
            continue
        logits = torch.as_tensor(np.einsum("ehd,hd->eh", q[s[es]], k[recv]) / math.sqrt(kq))   # <xWq[sender], xWk[receiver]>
        w = torch.softmax(logits, dim=0).numpy()                                              # over this receiver's edges
        want[recv] = np.einsum("eh,ej->hj", w, v[s[es]])
    want = want.reshape(n, nh * vd) @ a["wo"]
    got64 = O.Fp64Dense(s, r, n, activation="relu").attn_gnn(x, net)
    np.testing.assert_allclose(got64, want, rtol=1e-12, atol=1e-12)
    f32 = O.Fp32Gather(s, r, n, activation="relu")
    net32 = {"attn": {kk: (torch.as_tensor(np.asarray(vv, np.float32)) if isinstance(vv, np.ndarray) else vv) for kk, vv in a.items()},
             "mlp": [(torch.eye(c), torch.zeros(c))]}
    got32 = f32.attn_gnn(torch.as_tensor(x.astype(np.float32)), net32).numpy()
    # (the edge-list form divides 0 / 0 for the isolated receiver like the TF graph would; everything else agrees)
    np.testing.assert_allclose(got32[:5], want[:5], rtol=2e-5, atol=2e-5)
    assert np.all(got64[5] == 0.0)


def test_activations_and_gaussian_vs_torch():
    """tf.nn.leaky_relu default alpha 0.2, tf.nn.relu; MultivariateNormalDiag(0, 1).log_prob summed over nodes."""
    x = np.linspace(-3, 3, 13)
    d = O.Fp64Dense(np.zeros(1, np.int32), np.zeros(1, np.int32), 1, activation="leaky_relu")
    np.testing.assert_allclose(d.act(x), torch.nn.functional.leaky_relu(torch.as_tensor(x), 0.2).numpy(), rtol=0, atol=0)
    d = O.Fp64Dense(np.zeros(1, np.int32), np.zeros(1, np.int32), 1, activation="relu")
    np.testing.assert_allclose(d.act(x), torch.relu(torch.as_tensor(x)).numpy(), rtol=0, atol=0)
    z = np.random.default_rng(6).standard_normal((9, 4))
    want = torch.distributions.Normal(0.0, 1.0).log_prob(torch.as_tensor(z)).sum()
    assert abs(O.gaussian_log_prob_sum(z) - float(want)) < 1e-10


def test_unsorted_segment_mean_vs_torch_scatter():
    """tf.unsorted_segment_mean = segment_sum / max(count, 1) (empty segment -> 0), against torch's scatter_reduce("mean")."""
    rng = np.random.default_rng(7)
    n = 5
    s = np.array([0, 1, 2, 3, 3, 1], np.int32)
    r = np.array([1, 1, 2, 0, 0, 0], np.int32)                    # nodes 3 and 4 receive nothing
    x = rng.standard_normal((n, 3))
    o = O.Fp64Dense(s, r, n, agg="mean", combine="concat")
    agg = (o.adj @ x) / o.deg
    want = torch.zeros(n, 3, dtype=torch.float64).scatter_reduce(0, torch.as_tensor(r.astype(np.int64)).unsqueeze(1).expand(-1, 3),
                                                                  torch.as_tensor(x[s]), "mean", include_self=False)
    np.testing.assert_allclose(agg, want.numpy(), rtol=1e-12, atol=1e-12)
    assert np.all(agg[3:] == 0.0)
