"""CPU oracle for the GRevNet forward / inverse + log-det hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED at the TensorFlow / Sonnet / graph_nets / TFP boundary: the reference cannot be
imported here (tensorflow 1.x, dm-sonnet 1.34, graph_nets, tensorflow-probability 0.7.0 are absent,
`requirements.txt:1-7`; `grevnet.py` is missing from the tree) and it ships no tests or golden
vectors.  This file is therefore a *restatement* of the algorithm in
`/root/reference/gnn.py`, pinned by (1) agreement of two independent formulations below,
(2) analytic known-answer tests (closed-form micro case, Jacobian log-det, round trip, additivity,
permutation equivariance, scipy Gaussian) in tests/test_oracle.py, (3) the committed fixtures in
tests/golden/ generated from it by tests/golden/make_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product (graph-normalizing-flows_amd/) never does; it fails loudly without its HIP library.

Two formulations of the same maths:
  * `Fp64Dense`  - numpy float64, dense block adjacency matmul  (the numerical ground truth)
  * `Fp32Gather` - torch-CPU float32, gather + index_add_ in the reference's op order and
                   granularity (un-fused, s-net and t-net each redo gather+segment-reduce, exactly
                   like the TF graph of gnn.py:317-323).  This is also the timed CPU baseline
                   ("CPU restatement of reference (TensorFlow unavailable)").

Upstream semantics restated (all third-party, not under /root/reference):
  EdgeBlock(use_sender_nodes only) + IdentityModule      -> edges[e] = nodes[senders[e]]      (gnn.py:130-156)
  ReceivedEdgesToNodesAggregator(reducer)                 -> reducer(edges, receivers, sum(n_node))  (gnn.py:103,117)
  tf.unsorted_segment_sum / _mean                         -> sum ; sum / max(count, 1)          (gnn.py:239,245,251,256)
  snt.nets.MLP(activate_final=False), snt.Linear          -> x @ W + b, W:[in,out]              (gnn.py:159-180)
  tf.nn.leaky_relu default alpha = 0.2 ; tf.nn.relu                                             (run_grevnet.py:158,179,205)
  snt.LayerNorm() (DMSelfAttentionMLP(layer_norm=True))  -> (h - mean_f) / sqrt(var_f + 1e-5) * gamma + beta  (gnn.py:550-552)
  tfd.MultivariateNormalDiag(0,1).log_prob(z)             -> -0.5*sum(z^2) - D/2*ln(2*pi)       (run_grevnet.py:292-294)

Parameter container used everywhere in tests (plain python, no framework):
  mlp    := list of (W[in,out], b[out]) pairs, one per Linear layer
  params := {"s": [[mlp]*T, [mlp]*T], "t": [[mlp]*T, [mlp]*T]}      weight_sharing=False  (gnn.py:288-296)
            {"s": [mlp, mlp],          "t": [mlp, mlp]}              weight_sharing=True   (gnn.py:284-286)
  index [half][i]: half 0 nets read x0 and update x1, half 1 nets read x1 and update x0 (gnn.py:320-338).
"""
import math

import numpy as np

LN_2PI = math.log(2.0 * math.pi)
LN_EPS = 1e-5       # snt.LayerNorm default eps (Sonnet 1.x layer_norm.py; upstream-unpinned like the rest)


def attn_weight_keys(attn):
    """trainable tensors of one DMSelfAttentionMLP front-end (gnn.py:509-552)"""
    return ("wq", "wk", "wv", "wo") + (("ln_gamma", "ln_beta") if attn.get("layer_norm", False) else ())


def layer_norm_rows(h, gamma, beta):
    """gnn.py:550-552 `snt.LayerNorm()(new_nodes)` on a [N, H] array: per-row moments over the feature axis
    (tf.nn.moments(x, [1]): biased variance), then tf.nn.batch_normalization with eps = 1e-5, scale gamma [H],
    offset beta [H]."""
    mean = h.mean(axis=1, keepdims=True)
    var = ((h - mean) ** 2).mean(axis=1, keepdims=True)
    return (h - mean) / np.sqrt(var + LN_EPS) * gamma + beta


# ----------------------------------------------------------------------------------------------
# shared helpers
# ----------------------------------------------------------------------------------------------
def _net(params, kind, half, i, weight_sharing):
    """gnn.py:314-321 / 329-336: self.s[half] (shared) or self.s[half][i]."""
    return params[kind][half] if weight_sharing else params[kind][half][i]


def gaussian_log_prob_sum(z):
    """run_grevnet.py:292-294: sum_n MultivariateNormalDiag(0_D, 1_D).log_prob(z_n), in float64."""
    z = np.asarray(z, dtype=np.float64)
    n, d = z.shape
    return float(-0.5 * np.sum(z * z) - 0.5 * d * LN_2PI * n)


def assemble_log_prob(log_prob_zs, logdet, n_total):
    """run_grevnet.py:294-302: the scalars the reference logs, incl. the per-node forms."""
    log_prob_xs = log_prob_zs + logdet
    n = float(n_total)
    return {
        "log_prob_zs": log_prob_zs,
        "log_det_jacobian": logdet,
        "log_prob_xs": log_prob_xs,
        "total_loss": -log_prob_xs,
        "num_nodes": n,
        "loss_per_node": -log_prob_xs / n,
        "log_prob_xs_per_node": log_prob_xs / n,
        "log_prob_zs_per_node": log_prob_zs / n,
        "log_det_jacobian_per_node": logdet / n,
    }


# ----------------------------------------------------------------------------------------------
# formulation (i): numpy float64, dense adjacency
# ----------------------------------------------------------------------------------------------
class Fp64Dense:
    """Ground truth.  agg = A @ x with A[r, s] = number of edges s->r (a dense [N,N] matrix)."""

    def __init__(self, senders, receivers, n_total, agg="mean", combine="agg", epsilon=1.0,
                 activation="leaky_relu", alpha=0.2):
        assert agg in ("sum", "mean") and combine in ("agg", "concat")
        assert activation in ("leaky_relu", "relu")
        self.n = int(n_total)
        a = np.zeros((self.n, self.n), dtype=np.float64)
        np.add.at(a, (np.asarray(receivers, np.int64), np.asarray(senders, np.int64)), 1.0)
        self.adj = a
        self.deg = np.maximum(a.sum(axis=1, keepdims=True), 1.0)  # unsorted_segment_mean: max(count,1)
        self.agg, self.combine, self.eps = agg, combine, float(epsilon)
        self.activation, self.alpha = activation, float(alpha)

    def act(self, h):
        if self.activation == "relu":
            return np.maximum(h, 0.0)
        return np.maximum(h, self.alpha * h)  # tf.nn.leaky_relu = max(alpha*x, x)

    def mlp(self, h, layers):
        """gnn.py:159-180: (K-1) x [Linear + act] then Linear, activate_final=False."""
        k = len(layers)
        for j, (w, b) in enumerate(layers):
            h = h @ np.asarray(w, np.float64) + np.asarray(b, np.float64)
            if j < k - 1:
                h = self.act(h)
        return h

    def attn_gnn(self, x, net):
        """gnn.py:480-553 DMSelfAttentionMLP around gnn.py:385-477 DMSelfAttention, dense form.
        NB the reference calls attn_module(project_v, project_q, project_k, graph) (gnn.py:528), i.e. the
        module's "keys" are the Wq projection (taken at the SENDER) and its "queries" the Wk projection
        (taken at the RECEIVER): logit[e,h] = <xWq[sender], xWk[receiver]>_h (gnn.py:446-457).
        Softmax over the incoming edges of each receiver per head (graph_nets _unsorted_segment_softmax:
        exp(x - segment_max) / segment_sum, upstream-unpinned); a node without incoming edges gets 0."""
        a = net["attn"]
        nh, kq, vd = int(a["num_heads"]), int(a["kq_dim"]), int(a["v_dim"])
        n = x.shape[0]
        q = (x @ np.asarray(a["wq"], np.float64)).reshape(n, nh, kq)
        k = (x @ np.asarray(a["wk"], np.float64)).reshape(n, nh, kq)
        v = x @ np.asarray(a["wv"], np.float64)                       # [N, v]; repeated over heads (gnn.py:524)
        logits = np.einsum("shd,rhd->rsh", q, k)                        # [receiver, sender, head]
        if a.get("kq_dim_division", False):
            logits = logits / math.sqrt(kq)
        mult = self.adj[:, :, None]                                     # edge multiplicity r <- s
        masked = np.where(mult > 0, logits, -np.inf)
        mx = masked.max(axis=1, keepdims=True)
        mx = np.where(np.isfinite(mx), mx, 0.0)
        ex = mult * np.exp(np.where(mult > 0, logits - mx, -np.inf))
        den = ex.sum(axis=1, keepdims=True)
        w = np.where(den > 0, ex / np.where(den > 0, den, 1.0), 0.0)   # [r, s, h]
        agg = np.einsum("rsh,sj->rhj", w, v).reshape(n, nh * vd)       # [N, nh*v]
        new = agg @ np.asarray(a["wo"], np.float64)                     # snt.Linear(C, use_bias=False) gnn.py:538-540
        h0 = np.concatenate([x, new], axis=1) if a.get("concat", True) else new
        out = self.mlp(h0, net["mlp"])
        if a.get("residual", False):
            out = out + x
        if a.get("layer_norm", False):
            out = layer_norm_rows(out, np.asarray(a["ln_gamma"], np.float64), np.asarray(a["ln_beta"], np.float64))
        return out

    def gnn(self, x, layers):
        """gnn.py:155-156 NodeBlockGNN -> gnn.py:122-126 AggThenMLPBlock / 107-111 ConcatThenMLPBlock."""
        if isinstance(layers, dict):
            return self.attn_gnn(x, layers)
        agg = self.adj @ x
        if self.agg == "mean":
            agg = agg / self.deg
        h = np.concatenate([x, agg], axis=1) if self.combine == "concat" else self.eps * x + agg
        return self.mlp(h, layers)

    @staticmethod
    def bn_inverse(x, bn):
        """bn.inverse + bn.inverse_log_det_jacobian(x, 2) of make_batch_norm() (gnn.py:260-263, 310-313) in
        training mode.  Restated from tensorflow-probability 0.7 `bijectors/batch_normalization.py` and
        tf.layers.BatchNormalization (third party, absent: UNPINNED):
          _normalize:  (x - mean_B) / sqrt(var_B + eps) * gamma + beta, moments of this batch over axis 0 (biased)
          _inverse_log_det_jacobian: sum_f log gamma_f - 0.5 sum_f log(var_B,f + eps)  (a scalar), which
          Bijector._reduce_jacobian_det_over_event sums over the extra event dimension (event_ndims = 2): x N."""
        eps = float(bn.get("epsilon", 1e-3))
        mean = x.mean(axis=0)
        var = x.var(axis=0)
        gamma, beta = np.asarray(bn["gamma"], np.float64), np.asarray(bn["beta"], np.float64)
        y = (x - mean) / np.sqrt(var + eps) * gamma + beta
        ildj = x.shape[0] * float(np.sum(np.log(gamma)) - 0.5 * np.sum(np.log(var + eps)))
        return y, ildj, mean, var

    @staticmethod
    def bn_forward(z, bn):
        """bn.forward (gnn.py:356-358): _de_normalize with the MOVING statistics."""
        eps = float(bn.get("epsilon", 1e-3))
        gamma, beta = np.asarray(bn["gamma"], np.float64), np.asarray(bn["beta"], np.float64)
        mm, mv = np.asarray(bn["moving_mean"], np.float64), np.asarray(bn["moving_variance"], np.float64)
        return (z - beta) / gamma * np.sqrt(mv + eps) + mm

    def f(self, x, params, num_timesteps, weight_sharing=False):
        """gnn.py:304-341.  Returns (z[N,D], logdet scalar).  params["bn"] ([[bn]*T]*2) switches on
        use_batch_norm; the batch moments of every bijector are left in self.last_bn_moments."""
        x = np.asarray(x, np.float64)
        hdim = x.shape[1] // 2
        x0, x1 = x[:, :hdim].copy(), x[:, hdim:].copy()
        logdet = 0.0
        bns = params.get("bn")
        self.last_bn_moments = {}
        for i in range(num_timesteps):
            if bns is not None:
                x0, ildj, m, v = self.bn_inverse(x0, bns[0][i])
                self.last_bn_moments[(0, i)] = (m, v)
                logdet += ildj
            s = self.gnn(x0, _net(params, "s", 0, i, weight_sharing))
            t = self.gnn(x0, _net(params, "t", 0, i, weight_sharing))
            logdet += float(np.sum(s))
            x1 = x1 * np.exp(s) + t
            if bns is not None:
                x1, ildj, m, v = self.bn_inverse(x1, bns[1][i])
                self.last_bn_moments[(1, i)] = (m, v)
                logdet += ildj
            s = self.gnn(x1, _net(params, "s", 1, i, weight_sharing))
            t = self.gnn(x1, _net(params, "t", 1, i, weight_sharing))
            logdet += float(np.sum(s))
            x0 = x0 * np.exp(s) + t
        return np.concatenate([x0, x1], axis=1), logdet

    def g(self, z, params, num_timesteps, weight_sharing=False):
        """gnn.py:343-373.  Returns x[N,D].  With params["bn"]: s, t come from the still-normalised half,
        which is then de-normalised with the moving statistics (gnn.py:356-358, 369-371)."""
        z = np.asarray(z, np.float64)
        hdim = z.shape[1] // 2
        z0, z1 = z[:, :hdim].copy(), z[:, hdim:].copy()
        bns = params.get("bn")
        for i in reversed(range(num_timesteps)):
            s = self.gnn(z1, _net(params, "s", 1, i, weight_sharing))
            t = self.gnn(z1, _net(params, "t", 1, i, weight_sharing))
            if bns is not None:
                z1 = self.bn_forward(z1, bns[1][i])
            z0 = (z0 - t) * np.exp(-s)
            s = self.gnn(z0, _net(params, "s", 0, i, weight_sharing))
            t = self.gnn(z0, _net(params, "t", 0, i, weight_sharing))
            if bns is not None:
                z0 = self.bn_forward(z0, bns[0][i])
            z1 = (z1 - t) * np.exp(-s)
        return np.concatenate([z0, z1], axis=1)

    def log_prob(self, x, params, num_timesteps, weight_sharing=False):
        """run_grevnet.py:290-302 on top of f."""
        z, logdet = self.f(x, params, num_timesteps, weight_sharing)
        out = assemble_log_prob(gaussian_log_prob_sum(z), logdet, self.n)
        out["z"] = z
        return out


# ----------------------------------------------------------------------------------------------
# formulation (ii): torch-CPU float32, gather + index_add_, reference op order (also the CPU baseline)
# ----------------------------------------------------------------------------------------------
class Fp32Gather:
    """Un-fused fp32 restatement in the op order of the TF graph the reference builds."""

    def __init__(self, senders, receivers, n_total, agg="mean", combine="agg", epsilon=1.0,
                 activation="leaky_relu", alpha=0.2, dtype=None):
        import torch
        self.torch = torch
        self.dtype = dtype or torch.float32
        self.n = int(n_total)
        self.senders = torch.as_tensor(np.asarray(senders, np.int64))
        self.receivers = torch.as_tensor(np.asarray(receivers, np.int64))
        self.agg, self.combine, self.eps = agg, combine, float(epsilon)
        self.activation, self.alpha = activation, float(alpha)
        cnt = torch.zeros(self.n, dtype=self.dtype).index_add_(
            0, self.receivers, torch.ones(len(self.receivers), dtype=self.dtype))
        self.cnt = torch.clamp(cnt, min=1.0).unsqueeze(1)

    def to_t(self, a):
        return self.torch.as_tensor(np.asarray(a), dtype=self.dtype)

    def prep_params(self, params):
        """numpy -> torch once, outside any timed region."""
        def conv(m):
            if isinstance(m, dict) and "gamma" in m:          # a batch-norm bijector
                return {k: (self.to_t(v) if k != "epsilon" else v) for k, v in m.items()}
            if isinstance(m, dict) and "attn" in m:
                a = dict(m["attn"])
                for key in attn_weight_keys(a):
                    a[key] = self.to_t(a[key])
                return {"attn": a, "mlp": conv(m["mlp"])}
            if isinstance(m, list) and m and isinstance(m[0], tuple):
                return [(self.to_t(w), self.to_t(b)) for (w, b) in m]
            return [conv(q) for q in m]
        return {k: conv(v) for k, v in params.items()}

    def act(self, h):
        torch = self.torch
        if self.activation == "relu":
            return torch.relu(h)
        return torch.maximum(h, self.alpha * h)

    def mlp(self, h, layers):
        k = len(layers)
        call = self._mlp_calls = getattr(self, "_mlp_calls", -1) + 1
        for j, (w, b) in enumerate(layers):
            h = h @ w + b                       # snt.Linear: MatMul + Add
            if j < k - 1:
                h = self.act(h) if getattr(self, "kink", None) is None else self._act_kink(h, call, j)
        return h

    def _act_kink(self, h, call, j):
        """Kink-aware activation for gradient comparisons (tests/test_fullsize_gpu.py): a relu / leaky relu whose
        pre-activation is within kink["tol"] of zero has no side that float32 and float64 agree on, and which side a
        hidden unit takes decides whether a whole term of the gradient exists.  kink["masks"][(call, j)] (call = index of
        the MLP evaluation inside f: ((i * 2 + half) * 2 + net), j = hidden layer) is another implementation's
        "activation > 0" for every (node, unit); inside the tolerance band its side is taken, outside it this
        restatement's own - and every disagreement out there is counted in kink["outside"]."""
        torch = self.torch
        pos = h.detach() > 0
        dev = self.kink["masks"].get((call, j))
        if dev is not None:
            dev = torch.as_tensor(np.asarray(dev), dtype=torch.bool)
            amb = h.detach().abs() < self.kink["tol"]
            self.kink["ambiguous"] = self.kink.get("ambiguous", 0) + int(amb.sum())
            self.kink["flipped"] = self.kink.get("flipped", 0) + int((amb & (dev != pos)).sum())
            self.kink["outside"] = self.kink.get("outside", 0) + int((~amb & (dev != pos)).sum())
            pos = torch.where(amb, dev, pos)
        slope = 0.0 if self.activation == "relu" else self.alpha
        return torch.where(pos, h, slope * h)

    def attn_gnn(self, x, net):
        """Edge-list form of gnn.py:385-553 in the reference's op order (gathers materialised per edge,
        segment max / sum over receivers)."""
        torch = self.torch
        a = net["attn"]
        nh, kq, vd = int(a["num_heads"]), int(a["kq_dim"]), int(a["v_dim"])
        n, e = x.shape[0], self.senders.shape[0]
        q = (x @ a["wq"]).reshape(n, nh, kq)
        k = (x @ a["wk"]).reshape(n, nh, kq)
        v = (x @ a["wv"]).unsqueeze(1).expand(n, nh, vd)               # keras.backend.repeat
        sender_keys = q.index_select(0, self.senders)                   # "keys"    = Wq projection at the sender
        receiver_queries = k.index_select(0, self.receivers)            # "queries" = Wk projection at the receiver
        logits = (sender_keys * receiver_queries).sum(dim=-1)           # [E, nh]
        if a.get("kq_dim_division", False):
            logits = logits / math.sqrt(kq)
        idx = self.receivers.unsqueeze(1).expand(e, nh)
        seg_max = torch.full((n, nh), -float("inf"), dtype=self.dtype).scatter_reduce(0, idx, logits, "amax")
        ex = torch.exp(logits - seg_max.index_select(0, self.receivers))
        seg_sum = torch.zeros(n, nh, dtype=self.dtype).index_add_(0, self.receivers, ex)
        w = ex / seg_sum.index_select(0, self.receivers)
        attended = v.index_select(0, self.senders) * w.unsqueeze(-1)    # [E, nh, v]
        agg = torch.zeros(n, nh, vd, dtype=self.dtype).index_add_(0, self.receivers, attended)
        new = agg.reshape(n, nh * vd) @ a["wo"]
        h0 = torch.cat([x, new], dim=1) if a.get("concat", True) else new
        out = self.mlp(h0, net["mlp"])
        if a.get("residual", False):
            out = out + x
        if a.get("layer_norm", False):
            # gnn.py:550-552 snt.LayerNorm(): moments over axis 1, tf.nn.batch_normalization(eps=1e-5)
            mean = out.mean(dim=1, keepdim=True)
            var = ((out - mean) ** 2).mean(dim=1, keepdim=True)
            inv = torch.rsqrt(var + LN_EPS) * a["ln_gamma"]
            out = out * inv + (a["ln_beta"] - mean * inv)
        return out

    def gnn(self, x, layers):
        torch = self.torch
        if isinstance(layers, dict):
            return self.attn_gnn(x, layers)
        edges = x.index_select(0, self.senders)                       # GatherV2 (gnn.py:151-156), materialised [E,H]
        agg = torch.zeros_like(x).index_add_(0, self.receivers, edges)  # UnsortedSegmentSum
        if self.agg == "mean":
            agg = agg / self.cnt
        h = torch.cat([x, agg], dim=1) if self.combine == "concat" else self.eps * x + agg
        return self.mlp(h, layers)

    def bn_inverse(self, x, bn):
        """Same bijector in the TF op order: tf.nn.moments (mean, then mean of squared differences),
        tf.nn.batch_normalization ((x - mean) * rsqrt(var + eps) * gamma + beta)."""
        torch = self.torch
        eps = float(bn.get("epsilon", 1e-3))
        mean = x.mean(dim=0, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=0, keepdim=True)
        inv = torch.rsqrt(var + eps) * bn["gamma"]
        y = x * inv + (bn["beta"] - mean * inv)
        ildj = x.shape[0] * (torch.log(bn["gamma"]).sum() - 0.5 * torch.log(var + eps).sum())
        return y, ildj

    def bn_forward(self, z, bn):
        torch = self.torch
        eps = float(bn.get("epsilon", 1e-3))
        return (z - bn["beta"]) / bn["gamma"] * torch.sqrt(bn["moving_variance"] + eps) + bn["moving_mean"]

    def f(self, x, params, num_timesteps, weight_sharing=False):
        torch = self.torch
        self._mlp_calls = -1
        hdim = x.shape[1] // 2
        x0, x1 = x[:, :hdim], x[:, hdim:]                             # tf.split
        logdet = torch.zeros((), dtype=self.dtype)
        bns = params.get("bn")
        for i in range(num_timesteps):
            if bns is not None:
                x0, ildj = self.bn_inverse(x0, bns[0][i])
                logdet = logdet + ildj
            s = self.gnn(x0, _net(params, "s", 0, i, weight_sharing))
            t = self.gnn(x0, _net(params, "t", 0, i, weight_sharing))
            logdet = logdet + s.sum()
            x1 = x1 * torch.exp(s) + t
            if bns is not None:
                x1, ildj = self.bn_inverse(x1, bns[1][i])
                logdet = logdet + ildj
            s = self.gnn(x1, _net(params, "s", 1, i, weight_sharing))
            t = self.gnn(x1, _net(params, "t", 1, i, weight_sharing))
            logdet = logdet + s.sum()
            x0 = x0 * torch.exp(s) + t
        return torch.cat([x0, x1], dim=1), logdet

    def g(self, z, params, num_timesteps, weight_sharing=False):
        torch = self.torch
        hdim = z.shape[1] // 2
        z0, z1 = z[:, :hdim], z[:, hdim:]
        bns = params.get("bn")
        for i in reversed(range(num_timesteps)):
            s = self.gnn(z1, _net(params, "s", 1, i, weight_sharing))
            t = self.gnn(z1, _net(params, "t", 1, i, weight_sharing))
            if bns is not None:
                z1 = self.bn_forward(z1, bns[1][i])
            z0 = (z0 - t) * torch.exp(-s)
            s = self.gnn(z0, _net(params, "s", 0, i, weight_sharing))
            t = self.gnn(z0, _net(params, "t", 0, i, weight_sharing))
            if bns is not None:
                z0 = self.bn_forward(z0, bns[0][i])
            z1 = (z1 - t) * torch.exp(-s)
        return torch.cat([z0, z1], dim=1)

    def log_prob(self, x, params, num_timesteps, weight_sharing=False):
        """f + run_grevnet.py:292-302 entirely in the working dtype (what the TF graph does)."""
        torch = self.torch
        z, logdet = self.f(x, params, num_timesteps, weight_sharing)
        d = z.shape[1]
        lp = (-0.5 * (z * z).sum(dim=1) - 0.5 * d * LN_2PI).sum()
        out = assemble_log_prob(float(lp), float(logdet), self.n)
        out["z"] = z
        return out


# ----------------------------------------------------------------------------------------------
# training step: total_loss, its gradient, Adam  (run_grevnet.py:291-295, 340-377)
# ----------------------------------------------------------------------------------------------
def loss_and_grads(senders, receivers, n_total, x, params, num_timesteps, weight_sharing=False, **gnn_kw):
    """total_loss = -(sum_n MVN(0, I).log_prob(z_n) + log_det_jacobian) (run_grevnet.py:291-295) and
    d total_loss / d(every W, b), i.e. what optimizer.compute_gradients(total_loss) (run_grevnet.py:361-362)
    returns.  The reference differentiates its TF graph with tf.gradients; here torch autograd
    differentiates the float64 edition of the `Fp32Gather` restatement above (same op graph), which
    tests/test_oracle.py pins against central finite differences of `Fp64Dense`.
    Returns {"total_loss", "log_det_jacobian", "log_prob_zs", "z", "grads"}; grads has the layout of params.
    dtype=torch.float32 (keyword) runs the same autograd in single precision: what float32 arithmetic costs on the given
    inputs - the full-batch GPU tests derive their gradient tolerances from it.  kink={"masks": ..., "tol": ...}: the
    relu side of pre-activations within tol of zero is taken from another implementation (Fp32Gather._act_kink)."""
    import torch
    dtype = gnn_kw.pop("dtype", torch.float64)
    kink = gnn_kw.pop("kink", None)       # see Fp32Gather._act_kink
    o = Fp32Gather(senders, receivers, n_total, dtype=dtype, **gnn_kw)
    o.kink = kink
    pt = o.prep_params(params)
    leaves = []

    def mark(m):
        if isinstance(m, dict) and "attn" in m:           # attention net: wq, wk, wv, wo and the MLP are trainable
            a = dict(m["attn"])
            for k in attn_weight_keys(a):
                a[k] = a[k].clone().requires_grad_(True)
            return {"attn": a, "mlp": mark(m["mlp"])}
        if isinstance(m, dict) and "gamma" in m:          # batch-norm bijector: gamma and beta are trainable
            out = dict(m)
            for k in ("gamma", "beta"):
                out[k] = m[k].clone().requires_grad_(True)
            return out
        if isinstance(m, list) and m and isinstance(m[0], tuple):
            out = []
            for (w, b) in m:
                w = w.clone().requires_grad_(True)
                b = b.clone().requires_grad_(True)
                leaves.extend([w, b])
                out.append((w, b))
            return out
        return [mark(q) for q in m]

    pt = {k: mark(v) for k, v in pt.items()}
    z, logdet = o.f(o.to_t(x), pt, num_timesteps, weight_sharing)
    d = z.shape[1]
    log_prob_zs = (-0.5 * (z * z).sum(dim=1) - 0.5 * d * LN_2PI).sum()
    total_loss = -(log_prob_zs + logdet)
    total_loss.backward()

    def grads_of(m):
        if isinstance(m, dict) and "attn" in m:
            return {"attn": {k: m["attn"][k].grad.numpy().copy() for k in attn_weight_keys(m["attn"])},
                    "mlp": grads_of(m["mlp"])}
        if isinstance(m, dict) and "gamma" in m:
            return {"gamma": m["gamma"].grad.numpy().copy(), "beta": m["beta"].grad.numpy().copy()}
        if isinstance(m, list) and m and isinstance(m[0], tuple):
            return [(w.grad.numpy().copy(), b.grad.numpy().copy()) for (w, b) in m]
        return [grads_of(q) for q in m]

    return {"total_loss": float(total_loss.detach()), "log_det_jacobian": float(logdet.detach()),
            "log_prob_zs": float(log_prob_zs.detach()),
            "z": z.detach().numpy(), "grads": {k: grads_of(v) for k, v in pt.items()}}


def adam_step(w, g, m, v, t, lr, beta1=0.9, beta2=0.9, epsilon=1e-8):
    """tf.train.AdamOptimizer (TensorFlow 1.x, third party; training_ops ApplyAdam) as configured at
    run_grevnet.py:352-356, step t = 1, 2, ...:
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
        m <- beta1 m + (1 - beta1) g ;  v <- beta2 v + (1 - beta2) g^2 ;  w <- w - lr_t m / (sqrt(v) + epsilon)
    float64 numpy; returns (w, m, v)."""
    w, g, m, v = (np.asarray(a, np.float64) for a in (w, g, m, v))
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    return w - lr_t * m / (np.sqrt(v) + epsilon), m, v


def clip_by_value(g, lo, hi):
    """tf.clip_by_value (run_grevnet.py:363-367)."""
    return np.clip(np.asarray(g, np.float64), lo, hi)


def clip_by_norm(g, clip_norm):
    """tf.clip_by_norm on one tensor (run_grevnet.py:369-372): g * clip_norm / max(||g||_2, clip_norm)."""
    g = np.asarray(g, np.float64)
    return g * clip_norm / max(float(np.sqrt((g * g).sum())), clip_norm)


# ----------------------------------------------------------------------------------------------
# deterministic test-parameter generator (NOT the reference's initializer; just reproducible weights)
# ----------------------------------------------------------------------------------------------
def make_mlp_params(rng, in_dim, latent, out_dim, num_layers, bias_std=0.1, final_scale=1.0,
                    dtype=np.float32):
    """K Linear layers sized [latent]*(K-1)+[out] (gnn.py:165-166).  W ~ N(0, 2/(fan_in+fan_out))
    (glorot-style variance, gnn.py:171-172), b ~ N(0, bias_std) clipped at 2 sigma (gnn.py:173).
    `final_scale` multiplies the last layer (W and b) so |s| stays O(1) over many coupling steps."""
    sizes = [latent] * (num_layers - 1) + [out_dim]
    layers, fan_in = [], in_dim
    for j, fan_out in enumerate(sizes):
        std = math.sqrt(2.0 / (fan_in + fan_out))
        w = rng.standard_normal((fan_in, fan_out)) * std
        b = np.clip(rng.standard_normal(fan_out), -2.0, 2.0) * bias_std
        if j == len(sizes) - 1:
            w, b = w * final_scale, b * final_scale
        layers.append((w.astype(dtype), b.astype(dtype)))
        fan_in = fan_out
    return layers


def make_grevnet_params(seed, hdim, latent, num_layers, num_timesteps, combine="agg",
                        weight_sharing=False, bias_std=0.1, final_scale=1.0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    in_dim = 2 * hdim if combine == "concat" else hdim

    def one():
        return make_mlp_params(rng, in_dim, latent, hdim, num_layers, bias_std, final_scale, dtype)

    if weight_sharing:
        return {"s": [one(), one()], "t": [one(), one()]}
    return {"s": [[one() for _ in range(num_timesteps)] for _ in range(2)],
            "t": [[one() for _ in range(num_timesteps)] for _ in range(2)]}


def make_bn_params(seed, hdim, num_timesteps):
    """Non-trivial batch-norm variables for tests (the reference initialises gamma=1, beta=0, moving_mean=0,
    moving_variance=1; trained values are what matters for parity)."""
    rng = np.random.default_rng(seed)

    def one():
        return {"gamma": rng.uniform(0.5, 1.5, hdim).astype(np.float32),
                "beta": (0.2 * rng.standard_normal(hdim)).astype(np.float32),
                "moving_mean": (0.3 * rng.standard_normal(hdim)).astype(np.float32),
                "moving_variance": rng.uniform(0.5, 2.0, hdim).astype(np.float32), "epsilon": 1e-3}
    return [[one() for _ in range(num_timesteps)] for _ in range(2)]


def make_attn_net_params(rng, hdim, latent, num_layers, num_heads=8, kq_dim=10, v_dim=10, out_dim=80,
                         concat=True, kq_dim_division=False, residual=False, bias_std=0.1,
                         final_scale=1.0, dtype=np.float32, layer_norm=False):
    """One DMSelfAttentionMLP net (gnn.py:480-553): Wq, Wk [H, nh*kq], Wv [H, v] xavier-uniform
    (gnn.py:504-506), Wo [nh*v, C] (snt.Linear default init ~ 1/sqrt(fan_in)), then the MLP on
    [x || new] (H + C inputs) or new (C inputs)."""
    def xavier(fi, fo):
        a = math.sqrt(6.0 / (fi + fo))
        return rng.uniform(-a, a, size=(fi, fo)).astype(dtype)
    attn = {"num_heads": num_heads, "kq_dim": kq_dim, "v_dim": v_dim, "concat": concat,
            "kq_dim_division": kq_dim_division, "residual": residual,
            "wq": xavier(hdim, num_heads * kq_dim), "wk": xavier(hdim, num_heads * kq_dim),
            "wv": xavier(hdim, v_dim),
            "wo": (rng.standard_normal((num_heads * v_dim, out_dim)) / math.sqrt(num_heads * v_dim)).astype(dtype)}
    if layer_norm:      # snt.LayerNorm starts at gamma = 1, beta = 0; test parameters are non-trivial in both and keep
        attn["layer_norm"] = True           # s = LayerNorm(...) small enough that exp(s) stays tame over a few steps
        attn["ln_gamma"] = rng.uniform(0.2, 0.6, hdim).astype(dtype)
        attn["ln_beta"] = (0.1 * rng.standard_normal(hdim)).astype(dtype)
    in_dim = hdim + out_dim if concat else out_dim
    return {"attn": attn, "mlp": make_mlp_params(rng, in_dim, latent, hdim, num_layers, bias_std, final_scale, dtype)}


def make_attn_grevnet_params(seed, hdim, latent, num_layers, num_timesteps, weight_sharing=False, **kw):
    rng = np.random.default_rng(seed)

    def one():
        return make_attn_net_params(rng, hdim, latent, num_layers, **kw)

    if weight_sharing:
        return {"s": [one(), one()], "t": [one(), one()]}
    return {"s": [[one() for _ in range(num_timesteps)] for _ in range(2)],
            "t": [[one() for _ in range(num_timesteps)] for _ in range(2)]}


def pred_adj_blocks(z, n_node):
    """loss.py:154-159 pred_adj with distance_fn = scaled_hacky_sigmoid_l2 (loss.py:45-53), restated in the
    reference's own formula and float64:  D = r - 2 z z^T + r^T, D /= sqrt(dim), sigmoid(10 (1 - D)); the
    block-diagonal loss_mask (loss.py:131-151) and remove_diag keep, per graph, the [n_g, n_g] block with
    a zero diagonal - returned as a list of blocks (the rest of the dense matrix is zero by the mask)."""
    z = np.asarray(z, np.float64)
    dim = z.shape[1]
    out, off = [], 0
    for n in n_node:
        n = int(n)
        zz = z[off:off + n]
        r = np.sum(zz * zz, axis=1).reshape(-1, 1)
        d = r - 2.0 * zz @ zz.T + r.T
        d = d / math.sqrt(dim)
        p = 1.0 / (1.0 + np.exp(-10.0 * (1.0 - d)))
        np.fill_diagonal(p, 0.0)
        out.append(p)
        off += n
    return out


def batch_graphs(n_node, n_edge, senders_local, receivers_local, graph_ids):
    """Concatenate the chosen graphs with node-id offsets (what gn.utils_np.*_to_graphs_tuple does;
    graph_data.py:122, grevnet_synthetic_data.py:45-47).  Returns (n_node[B], n_edge[B], senders, receivers)."""
    eoff = np.concatenate([[0], np.cumsum(n_edge)])
    nn, ne, ss, rr, off = [], [], [], [], 0
    for gid in graph_ids:
        lo, hi = eoff[gid], eoff[gid + 1]
        ss.append(senders_local[lo:hi].astype(np.int64) + off)
        rr.append(receivers_local[lo:hi].astype(np.int64) + off)
        nn.append(int(n_node[gid]))
        ne.append(int(n_edge[gid]))
        off += int(n_node[gid])
    return (np.array(nn, np.int32), np.array(ne, np.int32),
            np.concatenate(ss).astype(np.int32), np.concatenate(rr).astype(np.int32))
