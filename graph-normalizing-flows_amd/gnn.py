"""Host-side mirror of the GRevNet / GNN call surface of /root/reference/gnn.py for the hot path
(gnn.py:100-180, 238-257, 266-267, 273-381): same class and function names, argument order and
return conventions, with PyTorch-ROCm tensors as containers and every bit of arithmetic executed by
the hand-written gfx950 kernels behind the C ABI (include/gnf.h).  Eager instead of TF graph mode.

Differences a reference user will notice (all deliberate):
  * `aggn_fn` / `activation` are tokens: pass this module's `unsorted_segment_sum`,
    `unsorted_segment_mean`, `relu`, `leaky_relu` (or the strings "sum" / "mean" / "relu" /
    "leaky_relu") where the reference passes `tf.unsorted_segment_*` / `tf.nn.*`.
  * parameters are explicit: `mlp.get_params()` / `mlp.set_params([(W, b), ...])`, `W` is [in, out]
    like snt.Linear; call `grevnet.repack()` after mutating weight tensors in place.
  * `use_batch_norm=True`: the bijectors are `grevnet.bns[half][i]` (`BatchNormBijector`: gamma, beta,
    moving_mean, moving_variance tensors); TFP-0.7 semantics restated.
  * there is no CPU fallback: without libgnf_hip.so / a HIP device every call raises GnfError.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _abi
from .graphs import GraphsTuple, csr_of

# ----------------------------------------------------------------------------------------------
# tokens standing in for tf.unsorted_segment_{sum,mean} (gnn.py:239,245,251,256) and tf.nn.*
# ----------------------------------------------------------------------------------------------
def unsorted_segment_sum(*_a, **_k):
    raise NotImplementedError("token only: pass it as aggn_fn; the reduction runs inside the HIP kernels")


def unsorted_segment_mean(*_a, **_k):
    raise NotImplementedError("token only: pass it as aggn_fn; the reduction runs inside the HIP kernels")


def relu(*_a, **_k):
    raise NotImplementedError("token only: pass it as activation; it runs inside the HIP kernels")


def leaky_relu(*_a, **_k):
    raise NotImplementedError("token only: pass it as activation; it runs inside the HIP kernels")


unsorted_segment_sum.gnf_agg = _abi.GNF_AGG_SUM
unsorted_segment_mean.gnf_agg = _abi.GNF_AGG_MEAN
relu.gnf_act = (_abi.GNF_ACT_RELU, 0.0)
leaky_relu.gnf_act = (_abi.GNF_ACT_LEAKY_RELU, 0.2)  # tf.nn.leaky_relu default alpha


def _agg_code(aggn_fn):
    if hasattr(aggn_fn, "gnf_agg"):
        return aggn_fn.gnf_agg
    name = aggn_fn if isinstance(aggn_fn, str) else getattr(aggn_fn, "__name__", "")
    if name in ("sum", "unsorted_segment_sum"):
        return _abi.GNF_AGG_SUM
    if name in ("mean", "avg", "unsorted_segment_mean"):
        return _abi.GNF_AGG_MEAN
    raise ValueError(f"unsupported aggn_fn {aggn_fn!r}: use unsorted_segment_sum / unsorted_segment_mean")


def _act_code(activation):
    if hasattr(activation, "gnf_act"):
        return activation.gnf_act
    name = activation if isinstance(activation, str) else getattr(activation, "__name__", "")
    if name == "relu":
        return (_abi.GNF_ACT_RELU, 0.0)
    if name == "leaky_relu":
        return (_abi.GNF_ACT_LEAKY_RELU, 0.2)
    raise ValueError(f"unsupported activation {activation!r}: use relu / leaky_relu")


# Mutation epoch of every parameter container in the process (MLP, attention block, batch-norm bijector bump it
# next to their own version counters): GRevNet._flow compares ONE integer instead of walking 4T nets per call
# (the walk cost 3 ms per call at T = 12, five calls per training iteration).
_EPOCH = [0]


def _touch():
    _EPOCH[0] += 1


_GEN = torch.Generator().manual_seed(12345)  # run_grevnet.py:108 default random_seed


def set_random_seed(seed):
    """Seed of the weight initialisers (the reference seeds TF at graph construction)."""
    _GEN.manual_seed(int(seed))


# ----------------------------------------------------------------------------------------------
# MLP  (gnn.py:159-180: snt.nets.MLP([latent]*(K-1)+[out], activate_final=False))
# ----------------------------------------------------------------------------------------------
class MLP:
    def __init__(self, layer_sizes, activation=relu, bias_init_stddev=0.1, name="mlp"):
        if not 1 <= len(layer_sizes) <= _abi.GNF_MAX_LAYERS:
            raise ValueError(f"num_layers must be in [1, {_abi.GNF_MAX_LAYERS}]")
        self.layer_sizes = [int(v) for v in layer_sizes]
        self.activation = activation
        self.act_code, self.alpha = _act_code(activation)
        self.bias_init_stddev = float(bias_init_stddev)
        self.name = name
        self.params = None      # list of (W[in,out], b[out]) fp32 tensors, created at first connect
        self.in_dim = None
        self.version = 0

    def ensure_built(self, in_dim, device):
        """Sonnet creates variables at first connection, inferring the input width; so do we.
        W ~ glorot_normal (truncated, fan_avg), b ~ truncated_normal(stddev) (gnn.py:171-174)."""
        if self.params is None:
            params, fan_in = [], int(in_dim)
            for fan_out in self.layer_sizes:
                std = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
                w = torch.empty(fan_in, fan_out)
                torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=_GEN)
                b = torch.empty(fan_out)
                bs = self.bias_init_stddev
                if bs > 0:
                    torch.nn.init.trunc_normal_(b, 0.0, bs, -2 * bs, 2 * bs, generator=_GEN)
                else:
                    b.zero_()
                params.append((w, b))
                fan_in = fan_out
            self.in_dim = int(in_dim)
            self.params = params
            self.version += 1
            _touch()
        if self.in_dim != int(in_dim):
            raise ValueError(f"{self.name}: built for input width {self.in_dim}, connected to {in_dim}")
        if self.params[0][0].device != torch.device(device):
            self.params = [(w.to(device), b.to(device)) for (w, b) in self.params]
            self.version += 1
            _touch()
        return self

    def set_params(self, layers):
        """layers: [(W[in,out], b[out]), ...] numpy or torch; shapes must chain and match layer_sizes."""
        if len(layers) != len(self.layer_sizes):
            raise ValueError(f"{self.name}: expected {len(self.layer_sizes)} layers, got {len(layers)}")
        params, fan_in = [], None
        for j, (w, b) in enumerate(layers):
            w = torch.as_tensor(np.asarray(w) if not isinstance(w, torch.Tensor) else w).to(torch.float32).contiguous()
            b = torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b).to(torch.float32).contiguous()
            if w.ndim != 2 or b.ndim != 1 or w.shape[1] != self.layer_sizes[j] or b.shape[0] != self.layer_sizes[j]:
                raise ValueError(f"{self.name}: layer {j} has W{tuple(w.shape)} b{tuple(b.shape)}, "
                                 f"expected out width {self.layer_sizes[j]}")
            if fan_in is not None and w.shape[0] != fan_in:
                raise ValueError(f"{self.name}: layer {j} input width {w.shape[0]} != previous output {fan_in}")
            fan_in = w.shape[1]
            params.append((w, b))
        self.in_dim = int(params[0][0].shape[0])
        self.params = params
        self.version += 1
        _touch()
        return self

    def get_params(self):
        return None if self.params is None else [(w.detach().cpu().numpy().copy(), b.detach().cpu().numpy().copy())
                                                 for (w, b) in self.params]

    def dims(self):
        return [self.in_dim] + self.layer_sizes

    def packed_floats(self):
        return sum(2 * _pad16(i) * _pad16(o) + _pad16(o) for i, o in zip(self.dims()[:-1], self.dims()[1:]))

    def fill_desc(self, desc, packed_ptr):
        desc.num_layers = len(self.layer_sizes)
        for j, d in enumerate(self.dims()):
            desc.dims[j] = d
        for j, (w, b) in enumerate(self.params):
            desc.W[j] = w.data_ptr()
            desc.b[j] = b.data_ptr()
        desc.packed = packed_ptr

    def __call__(self, *_a, **_k):
        raise NotImplementedError("the MLP is evaluated inside the fused coupling kernels; call the GNN block / GRevNet")


def _pad16(v):
    return (int(v) + 15) // 16 * 16


def make_mlp_model(latent_dim, output_dim, num_layers, activation=relu, l2_regularizer_weight=0.01,
                   bias_init_stddev=0.1):
    """gnn.py:159-180.  `l2_regularizer_weight` is accepted and ignored exactly as in the reference
    (its regularizers are commented out, gnn.py:175-178).  output_dim may arrive as a float
    (run_grevnet.py:157 passes node_embedding_dim / 2)."""
    layers = [int(latent_dim)] * (int(num_layers) - 1)
    layers.append(int(output_dim))
    return MLP(layers, activation=activation, bias_init_stddev=bias_init_stddev)


# ----------------------------------------------------------------------------------------------
# node blocks (gnn.py:100-156)
# ----------------------------------------------------------------------------------------------
class _NodeBlock:
    combine = None

    def spec(self):
        return _abi.GnfGnnSpec(self._agg, self.combine, float(getattr(self, "epsilon", 0.0)),
                               self._mlp.act_code, float(self._mlp.alpha))

    def in_dim(self, h):
        return 2 * h if self.combine == _abi.GNF_COMBINE_CONCAT else h

    def signature(self):
        """What must be identical across the s/t GNNs of one GRevNet (they come from one make_gnn_fn)."""
        return (type(self).__name__, self.combine, self._agg, float(getattr(self, "epsilon", 0.0)),
                self._mlp.act_code, self._mlp.alpha)

    def attn_version(self):
        return 0

    def attn_desc(self, h, device):
        """ctypes GnfAttn of the attention front-end, or None for the message-passing blocks."""
        return None

    def _build(self, graph):
        """One GNN evaluation outside a coupling: gnf_gnn_apply_f32."""
        lib = _abi.lib()
        x = graph.nodes
        if x.device.type != "cuda":
            raise _abi.GnfError("GNN blocks run on a HIP device only (no CPU path)")
        x = x.to(torch.float32).contiguous()
        n, h = x.shape
        mlp = self._mlp.ensure_built(self.in_dim(h), x.device)
        desc = _abi.GnfMlp()
        mlp.fill_desc(desc, 0)
        attn = self.attn_desc(h, x.device)          # kept alive until the call returns
        if attn is not None:
            desc.attn = C.pointer(attn)
        spec = self.spec()
        csr = csr_of(graph)
        out = torch.empty(n, mlp.layer_sizes[-1], dtype=torch.float32, device=x.device)
        ws_bytes = lib.gnf_gnn_workspace_bytes(n, h, C.byref(desc), spec.combine)
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _abi.check(lib.gnf_gnn_apply_f32(C.byref(csr.desc), C.byref(desc), C.byref(spec), _abi.ptr(x),
                                             x.stride(0), h, _abi.ptr(out), out.stride(0), _abi.ptr(ws),
                                             ws_bytes, _abi.stream_ptr(x.device)), "gnf_gnn_apply_f32")
        return graph.replace(nodes=out)

    __call__ = _build


class ConcatThenMLPBlock(_NodeBlock):
    """gnn.py:100-111: MLP(concat[nodes, aggregate(received edges)])."""
    combine = _abi.GNF_COMBINE_CONCAT

    def __init__(self, aggn_fn, make_mlp_fn, name="AggThenMLPBlock"):
        self._agg = _agg_code(aggn_fn)
        self._mlp = make_mlp_fn()
        self.name = name


class AggThenMLPBlock(_NodeBlock):
    """gnn.py:114-126: MLP(epsilon * nodes + aggregate(received edges))."""
    combine = _abi.GNF_COMBINE_EPS

    def __init__(self, aggn_fn, make_mlp_fn, epsilon, name="AggThenMLPBlock"):
        self._agg = _agg_code(aggn_fn)
        self._mlp = make_mlp_fn()
        self.epsilon = epsilon
        self.name = name


class DMSelfAttentionMLP(_NodeBlock):
    """gnn.py:480-553 (with DMSelfAttention, gnn.py:385-477, fused in): edge-list multi-head
    self-attention over the incoming edges of every node, heads concatenated and projected to
    `concat_heads_output_dim`, optionally concatenated with the node's own features, then the MLP.
    It is a GNN module by itself (the reference does not wrap it in NodeBlockGNN, gnn.py:556-573).
    Weights (all [in, out], no bias, gnn.py:509-540):
    wq, wk [H, heads*kq], wv [H, v], wo [heads*v, C]; created at first connection like Sonnet does
    (xavier-uniform for q/k/v, gnn.py:504-506; Sonnet's default 1/sqrt(fan_in) truncated normal for wo).
    `layer_norm=True` (gnn.py:550-552, snt.LayerNorm over the block's output) adds ln_gamma (ones) and ln_beta
    (zeros) of the MLP's output width; both are trained."""
    combine = _abi.GNF_COMBINE_EPS   # unused by attention nets
    _agg = _abi.GNF_AGG_SUM          # unused by attention nets

    def __init__(self, kq_dim, v_dim, make_mlp_fn, num_heads=8, concat_heads_output_dim=20, concat=True,
                 residual=False, layer_norm=False, kq_dim_division=False, name="dm_self_attention"):
        self.kq_dim = int(kq_dim)
        self.v_dim = int(v_dim)
        self.mlp = make_mlp_fn()
        self._mlp = self.mlp
        self.num_heads = int(num_heads)
        self.concat_heads_output_dim = int(concat_heads_output_dim)
        self.concat = bool(concat)
        self.residual = bool(residual)
        self.layer_norm = bool(layer_norm)
        self.kq_dim_division = bool(kq_dim_division)
        self.name = name
        self.attn_params = None     # {"wq", "wk", "wv", "wo"} fp32 tensors
        self._attn_version = 0

    def in_dim(self, h):
        return (h if self.concat else 0) + self.concat_heads_output_dim

    def signature(self):
        return (type(self).__name__, self.num_heads, self.kq_dim, self.v_dim, self.concat_heads_output_dim,
                self.concat, self.residual, self.layer_norm, self.kq_dim_division, self._mlp.act_code, self._mlp.alpha)

    def attn_version(self):
        return self._attn_version

    def _shapes(self, h):
        nq = self.num_heads * self.kq_dim
        shapes = {"wq": (h, nq), "wk": (h, nq), "wv": (h, self.v_dim),
                  "wo": (self.num_heads * self.v_dim, self.concat_heads_output_dim)}
        if self.layer_norm:
            w_out = self._mlp.layer_sizes[-1]
            shapes.update(ln_gamma=(w_out,), ln_beta=(w_out,))
        return shapes

    def attn_keys(self):
        return ("wq", "wk", "wv", "wo") + (("ln_gamma", "ln_beta") if self.layer_norm else ())

    def ensure_attn_built(self, h, device):
        if self.attn_params is None:
            p = {}
            for key, shp in self._shapes(h).items():
                if len(shp) == 1:               # snt.LayerNorm: gamma = 1, beta = 0
                    p[key] = torch.ones(shp) if key == "ln_gamma" else torch.zeros(shp)
                    continue
                fi, fo = shp
                w = torch.empty(fi, fo)
                if key == "wo":
                    std = 1.0 / math.sqrt(fi)
                    torch.nn.init.trunc_normal_(w, 0.0, std, -2 * std, 2 * std, generator=_GEN)
                else:
                    a = math.sqrt(6.0 / (fi + fo))
                    w.uniform_(-a, a, generator=_GEN)
                p[key] = w
            self.attn_params = p
            self._attn_version += 1
            _touch()
        for key, shp in self._shapes(h).items():
            if tuple(self.attn_params[key].shape) != shp:
                raise ValueError(f"{self.name}: {key} has shape {tuple(self.attn_params[key].shape)}, expected {shp}")
        if self.attn_params["wq"].device != torch.device(device):
            self.attn_params = {k: v.to(device) for k, v in self.attn_params.items()}
            self._attn_version += 1
            _touch()
        return self

    def set_attn_params(self, attn):
        """attn: dict with wq, wk, wv, wo (numpy or torch, [in, out]) and, for a layer_norm block, ln_gamma, ln_beta
        [MLP output width]; other keys are ignored."""
        p = {}
        for key in self.attn_keys():
            w = attn[key]
            w = torch.as_tensor(np.asarray(w) if not isinstance(w, torch.Tensor) else w).to(torch.float32).contiguous()
            if w.ndim != (1 if key.startswith("ln_") else 2):
                raise ValueError(f"{self.name}: {key} must be " + ("1-D [out]" if key.startswith("ln_") else "2-D [in, out]"))
            p[key] = w
        self.attn_params = p
        self._attn_version += 1
        _touch()
        return self

    def get_attn_params(self):
        return None if self.attn_params is None else {k: v.detach().cpu().numpy().copy() for k, v in self.attn_params.items()}

    def attn_desc(self, h, device):
        self.ensure_attn_built(h, device)
        p = self.attn_params
        return _abi.GnfAttn(self.num_heads, self.kq_dim, self.v_dim, self.concat_heads_output_dim,
                            int(self.concat), int(self.kq_dim_division), int(self.residual), int(self.layer_norm),
                            p["wq"].data_ptr(), p["wk"].data_ptr(), p["wv"].data_ptr(), p["wo"].data_ptr(),
                            p["ln_gamma"].data_ptr() if self.layer_norm else 0,
                            p["ln_beta"].data_ptr() if self.layer_norm else 0)


def dm_self_attn_gnn(kq_dim, v_dim, make_mlp_fn, num_heads, concat_heads_output_dim, concat=True,
                     residual=False, layer_norm=False, kq_dim_division=False):      # gnn.py:556-573
    return DMSelfAttentionMLP(kq_dim=kq_dim, v_dim=v_dim, make_mlp_fn=make_mlp_fn, num_heads=num_heads,
                              concat_heads_output_dim=concat_heads_output_dim, concat=concat,
                              residual=residual, layer_norm=layer_norm, kq_dim_division=kq_dim_division)


class IdentityModule:
    """gnn.py:130-132: the edge model (edges[e] = nodes[senders[e]]); fused away in the kernels."""

    def __call__(self, inputs):
        return inputs


EDGE_BLOCK_OPT = {
    "use_edges": False,
    "use_receiver_nodes": False,
    "use_sender_nodes": True,
    "use_globals": False,
}


class NodeBlockGNN:
    """gnn.py:143-156: node_block(edge_block(graph)); the edge block only broadcasts sender nodes
    to edges, which the CSR gather inside the kernels does without materialising [E, H]."""

    def __init__(self, node_block, edge_block_opt=EDGE_BLOCK_OPT, name="NodeBlockGNN"):
        if not isinstance(node_block, (AggThenMLPBlock, ConcatThenMLPBlock)):
            raise TypeError("NodeBlockGNN supports AggThenMLPBlock / ConcatThenMLPBlock node blocks "
                            "(GRU and dense attention blocks are outside the hot path, SURVEY.md 8f)")
        if dict(edge_block_opt) != EDGE_BLOCK_OPT:
            raise ValueError("only the reference's EDGE_BLOCK_OPT (sender nodes only) is supported")
        self._node_block = node_block
        self.name = name

    def _build(self, graph):
        return self._node_block(graph)

    __call__ = _build


def avg_then_mlp_gnn(make_mlp_fn, epsilon):          # gnn.py:238-241
    return NodeBlockGNN(AggThenMLPBlock(unsorted_segment_mean, make_mlp_fn, epsilon))


def sum_then_mlp_gnn(make_mlp_fn, epsilon):          # gnn.py:244-247
    return NodeBlockGNN(AggThenMLPBlock(unsorted_segment_sum, make_mlp_fn, epsilon))


def sum_concat_then_mlp_gnn(make_mlp_fn):            # gnn.py:250-252
    return NodeBlockGNN(ConcatThenMLPBlock(unsorted_segment_sum, make_mlp_fn))


def avg_concat_then_mlp_gnn(make_mlp_fn):            # gnn.py:255-257
    return NodeBlockGNN(ConcatThenMLPBlock(unsorted_segment_mean, make_mlp_fn))


def get_gnns(num_timesteps, make_gnn_fn):            # gnn.py:266-267
    return [make_gnn_fn() for _ in range(num_timesteps)]


# ----------------------------------------------------------------------------------------------
# make_batch_norm (gnn.py:260-263)
# ----------------------------------------------------------------------------------------------
class BatchNormBijector:
    """tfb.BatchNormalization(batchnorm_layer=tf.layers.BatchNormalization(axis=-1,
    gamma_constraint=lambda x: relu(x) + 1e-6), training=True): variables gamma (ones), beta (zeros),
    moving_mean (zeros), moving_variance (ones), epsilon 1e-3, momentum 0.99 (tf.layers defaults), created at
    first connection.  The arithmetic runs inside gnf_grevnet_f32 (GnfBatchNorm); this object owns the
    variables, the batch moments of the last f(), and the moving-average / constraint updates a training step
    applies (UPDATE_OPS + the optimizer's constraint projection)."""
    epsilon = 1e-3
    momentum = 0.99

    def __init__(self):
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None
        self.batch_mean = self.batch_variance = None
        self.version = 0

    def ensure_built(self, hdim, device):
        if self.gamma is None:
            self.gamma = torch.ones(hdim)
            self.beta = torch.zeros(hdim)
            self.moving_mean = torch.zeros(hdim)
            self.moving_variance = torch.ones(hdim)
            self.version += 1
            _touch()
        if self.gamma.shape[0] != hdim:
            raise ValueError(f"batch norm built for width {self.gamma.shape[0]}, connected to {hdim}")
        if self.gamma.device != torch.device(device) or self.batch_mean is None:
            for k in ("gamma", "beta", "moving_mean", "moving_variance"):
                setattr(self, k, getattr(self, k).to(device=device, dtype=torch.float32).contiguous())
            self.batch_mean = torch.zeros(hdim, dtype=torch.float32, device=device)
            self.batch_variance = torch.ones(hdim, dtype=torch.float32, device=device)
            self.version += 1
            _touch()
        return self

    def set_params(self, d):
        for k in ("gamma", "beta", "moving_mean", "moving_variance"):
            v = d[k]
            v = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(torch.float32).contiguous()
            setattr(self, k, v)
        self.batch_mean = None
        self.version += 1
        _touch()

    def get_params(self):
        return {k: getattr(self, k).detach().cpu().numpy().copy()
                for k in ("gamma", "beta", "moving_mean", "moving_variance")}

    def fill_desc(self, desc):
        desc.gamma, desc.beta = self.gamma.data_ptr(), self.beta.data_ptr()
        desc.moving_mean, desc.moving_variance = self.moving_mean.data_ptr(), self.moving_variance.data_ptr()
        desc.batch_mean, desc.batch_variance = self.batch_mean.data_ptr(), self.batch_variance.data_ptr()
        desc.epsilon = self.epsilon

    def update_moving_statistics(self):
        """tf.layers.BatchNormalization's UPDATE_OPS with the moments of the last f():
        moving <- moving * momentum + batch * (1 - momentum)."""
        self.moving_mean.mul_(self.momentum).add_(self.batch_mean, alpha=1.0 - self.momentum)
        self.moving_variance.mul_(self.momentum).add_(self.batch_variance, alpha=1.0 - self.momentum)

    def apply_gamma_constraint(self):
        """gamma_constraint = relu(x) + 1e-6, projected after an optimizer update."""
        self.gamma.clamp_(min=0.0).add_(1e-6)


def make_batch_norm():                                # gnn.py:260-263
    return BatchNormBijector()


# ----------------------------------------------------------------------------------------------
# GRevNet (gnn.py:273-381)
# ----------------------------------------------------------------------------------------------
class GRevNet:
    def __init__(self, make_gnn_fn, num_timesteps, node_embedding_dim, use_batch_norm=False,
                 weight_sharing=False, name="GRevNet", sync_batch_norm=False):
        # sync_batch_norm (not in the reference, which is single-device): under graph sharding with
        # torch.distributed, take the bijectors' batch moments over ALL ranks' nodes (one small all-reduce per
        # bijector call through GnfFlow.bn_allreduce), so that N ranks reproduce what one device computes for the
        # whole batch (gnn.py:310-313).  False: per-shard moments, no extra collective.
        self.sync_batch_norm = bool(sync_batch_norm)
        self.bn_process_group = None   # torch.distributed group of the cross-rank moments (None: the default group)
        # an RCCL communicator of the library's own (sharding.RcclComm): the moments are then all-reduced by
        # gnf_rccl_allreduce_sum_f64 straight from the C launch path (no Python callback, no torch.distributed)
        self.bn_rccl_comm = None
        self.num_timesteps = int(num_timesteps)
        self.weight_sharing = bool(weight_sharing)
        self.node_embedding_dim = node_embedding_dim  # accepted and unused, as in gnn.py:277
        if weight_sharing:                            # gnn.py:284-286
            self.s = [make_gnn_fn(), make_gnn_fn()]
            self.t = [make_gnn_fn(), make_gnn_fn()]
        else:                                         # gnn.py:288-296
            self.s = [get_gnns(num_timesteps, make_gnn_fn), get_gnns(num_timesteps, make_gnn_fn)]
            self.t = [get_gnns(num_timesteps, make_gnn_fn), get_gnns(num_timesteps, make_gnn_fn)]
        self.use_batch_norm = bool(use_batch_norm)
        # gnn.py:298-299: one bijector per half-step, created whether or not it is used
        self.bns = [[make_batch_norm() for _ in range(self.num_timesteps)],
                    [make_batch_norm() for _ in range(self.num_timesteps)]]
        self.name = name
        self._cache = None       # (key, flow desc, keep-alive objects)
        self._cache_fast = None  # (device, hdim, fused, sync world, mutation epoch) the cache was last validated at
        self.fused = True        # False: hide the packed weights -> the layered (generic) kernels run
        self.last_sums = None    # device fp64 [2]: log_det_jacobian, sum(z^2) of the last f()

    # ---- parameter plumbing ------------------------------------------------------------------
    def _gnns(self, kind):
        nets = self.s if kind == "s" else self.t
        return [nets[0], nets[1]] if self.weight_sharing else list(nets[0]) + list(nets[1])

    @staticmethod
    def _block_of(g):
        return g._node_block if isinstance(g, NodeBlockGNN) else g

    def blocks(self, kind):
        """Flat list of node blocks in ABI order: index half*T + i (or half with weight sharing)."""
        return [self._block_of(g) for g in self._gnns(kind)]

    def _blocks(self):
        return self.blocks("s") + self.blocks("t")

    def mlps(self, kind):
        """Flat list of MLPs in ABI order: index half*T + i (or half with weight sharing)."""
        return [b._mlp for b in self.blocks(kind)]

    def set_params(self, params):
        """params in the oracle / fixture layout: {"s": [[mlp]*T, [mlp]*T], "t": ...} or, with weight
        sharing, {"s": [mlp, mlp], "t": [mlp, mlp]}; mlp = [(W[in,out], b[out]), ...]."""
        for kind in ("s", "t"):
            flat = list(params[kind]) if self.weight_sharing else list(params[kind][0]) + list(params[kind][1])
            for blk, net in zip(self.blocks(kind), flat):
                if isinstance(net, dict):       # attention net: {"attn": {wq, wk, wv, wo, ...}, "mlp": [...]}
                    blk.set_attn_params(net["attn"])
                    blk._mlp.set_params(net["mlp"])
                else:
                    blk._mlp.set_params(net)
        if "bn" in params and params["bn"] is not None:   # [[{gamma, beta, moving_mean, moving_variance}]*T]*2
            for half in range(2):
                for i in range(self.num_timesteps):
                    self.bns[half][i].set_params(params["bn"][half][i])
        self._cache = None
        return self

    def get_params(self):
        out = {}
        for kind in ("s", "t"):
            flat = [({"attn": b.get_attn_params(), "mlp": b._mlp.get_params()} if isinstance(b, DMSelfAttentionMLP)
                     else b._mlp.get_params()) for b in self.blocks(kind)]
            t = self.num_timesteps
            out[kind] = flat if self.weight_sharing else [flat[:t], flat[t:]]
        if self.use_batch_norm and self.bns[0] and self.bns[0][0].gamma is not None:
            out["bn"] = [[b.get_params() for b in half] for half in self.bns]
        return out

    def repack(self):
        """Call after mutating weight tensors in place (the packed MFMA copies are stale otherwise)."""
        self._cache = None

    def _flow(self, hdim, device):
        fast = (str(device), hdim, bool(self.fused), self._bn_sync_key(), _EPOCH[0])
        if self._cache is not None and self._cache_fast == fast:
            return self._cache[1]
        flow = self._flow_slow(hdim, device)
        # building may itself have touched containers (first connect, device move): key on the epoch AFTER it
        self._cache_fast = (str(device), hdim, bool(self.fused), self._bn_sync_key(), _EPOCH[0])
        return flow

    def _flow_slow(self, hdim, device):
        lib = _abi.lib()
        blocks = self._blocks()
        b0 = blocks[0]
        for b in blocks:
            if b.signature() != b0.signature():
                raise ValueError("all s/t GNNs of a GRevNet must come from the same make_gnn_fn")
        s_mlps = [m.ensure_built(b0.in_dim(hdim), device) for m in self.mlps("s")]
        t_mlps = [m.ensure_built(b0.in_dim(hdim), device) for m in self.mlps("t")]
        attn_descs = [b.attn_desc(hdim, device) for b in blocks]          # None for message-passing blocks
        bn_list = [b.ensure_built(hdim, device) for half in self.bns for b in half] if self.use_batch_norm else []
        sync = self._bn_sync_world() if bn_list else 0
        key = (str(device), hdim, bool(self.fused), tuple(m.version for m in s_mlps + t_mlps),
               tuple(b.attn_version() for b in blocks), tuple(b.version for b in bn_list), sync,
               self._bn_sync_key() if bn_list else 0)
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        n = len(s_mlps)
        s_arr, t_arr = (_abi.GnfMlp * n)(), (_abi.GnfMlp * n)()
        attn_arr = (_abi.GnfAttn * (2 * n))()
        for q, ad in enumerate(attn_descs):
            if ad is not None:
                attn_arr[q] = ad
        sizes = [m.packed_floats() for m in s_mlps + t_mlps]
        packed = torch.empty(max(sum(sizes), 1), dtype=torch.float32, device=device)
        off = 0
        with torch.cuda.device(device):
            st = _abi.stream_ptr(device)
            for arr, mlps in ((s_arr, s_mlps), (t_arr, t_mlps)):
                for q, m in enumerate(mlps):
                    m.fill_desc(arr[q], packed.data_ptr() + 4 * off if self.fused else 0)
                    aq = q + (0 if arr is s_arr else n)
                    if attn_descs[aq] is not None:
                        arr[q].attn = C.cast(C.byref(attn_arr, aq * C.sizeof(_abi.GnfAttn)), C.POINTER(_abi.GnfAttn))
                    if lib.gnf_packed_floats(C.byref(arr[q])) != m.packed_floats():
                        raise _abi.GnfError("packed size mismatch between binding and library")
                    if self.fused:
                        _abi.check(lib.gnf_pack_mlp(C.byref(arr[q]), C.c_void_p(packed.data_ptr() + 4 * off), st),
                                   "gnf_pack_mlp")
                    off += m.packed_floats()
        bn_arr = None
        if bn_list:
            bn_arr = (_abi.GnfBatchNorm * len(bn_list))()
            for q, b in enumerate(bn_list):       # index half*T + i
                b.fill_desc(bn_arr[q])
        flow = _abi.GnfFlow(self.num_timesteps, int(self.weight_sharing),
                            C.cast(s_arr, C.POINTER(_abi.GnfMlp)), C.cast(t_arr, C.POINTER(_abi.GnfMlp)),
                            b0.spec(), C.cast(bn_arr, C.POINTER(_abi.GnfBatchNorm)) if bn_arr is not None else None)
        hook_keep = None
        if self.use_batch_norm and self.sync_batch_norm and self.bn_rccl_comm is not None:
            # the library's own hook on a native communicator (ABI v9)
            sync_buf = torch.zeros(2 * hdim + 1, dtype=torch.float64, device=device)
            flow.bn_allreduce = C.cast(lib.gnf_rccl_allreduce_sum_f64, _abi.BN_ALLREDUCE_FN)
            flow.bn_allreduce_ctx = self.bn_rccl_comm.handle
            flow.bn_sync_buf = sync_buf.data_ptr()
            hook_keep = (self.bn_rccl_comm, sync_buf)
        elif sync > 1:   # cross-rank batch moments: the library fills sync_buf, calls back, reads the reduced sums
            import torch.distributed as dist
            sync_buf = torch.zeros(2 * hdim + 1, dtype=torch.float64, device=device)
            group = self.bn_process_group

            def _hook(_ctx, _buf, _count, _stream):
                try:   # SUM over ranks, ordered on the current stream (the one the library call was given)
                    dist.all_reduce(sync_buf, group=group)
                    return 0
                except Exception as exc:   # never raise through the C frame
                    import sys
                    print(f"[gnf] bn_allreduce hook failed: {exc!r}", file=sys.stderr)
                    return -1
            cb = _abi.BN_ALLREDUCE_FN(_hook)
            flow.bn_allreduce = cb
            flow.bn_sync_buf = sync_buf.data_ptr()
            hook_keep = (cb, sync_buf)
        self._cache = (key, flow, (s_arr, t_arr, attn_arr, packed, s_mlps, t_mlps, blocks, bn_arr, bn_list, hook_keep))
        return flow

    def _bn_sync_world(self):
        """World size the batch-norm moments are taken over (1: this process only)."""
        if not (self.use_batch_norm and self.sync_batch_norm):
            return 1
        if self.bn_rccl_comm is not None:
            if not self.bn_rccl_comm.handle:   # (a cached GnfFlow would hold the dangling ncclComm_t)
                raise _abi.GnfError("sync_batch_norm: bn_rccl_comm has been destroyed")
            return self.bn_rccl_comm.n_ranks + 1000   # (distinct from any torch.distributed world size)
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.bn_process_group)

    def _bn_sync_key(self):
        """Cache key of the cross-rank moments: the world size AND the communicator's identity - a flow built on one
        native communicator must not be reused after that one was destroyed or swapped for another of the same size."""
        comm = self.bn_rccl_comm
        return (self._bn_sync_world(), comm.handle if comm is not None and self.use_batch_norm and self.sync_batch_norm else 0)

    def _run(self, graph, direction, sums_out=None):
        lib = _abi.lib()
        x = graph.nodes
        if x.device.type != "cuda":
            raise _abi.GnfError("GRevNet runs on a HIP device only (no CPU path)")
        if x.ndim != 2 or x.shape[1] % 2:
            raise ValueError(f"nodes must be [N, D] with even D (tf.split, gnn.py:306); got {tuple(x.shape)}")
        n, d = x.shape
        dev = x.device
        # TF ops are functional: the result goes to a NEW buffer; the library reads `src` and writes `out`
        # (gnf_grevnet_from_f32: no separate copy pass where the fused kernel runs the first half-step)
        src = x.to(torch.float32)
        if n > 0 and (src.stride(1) != 1 or src.stride(0) < d):   # (expand()ed / as_strided views: rows must not overlap)
            src = src.contiguous()
        out = torch.empty((n, d), dtype=torch.float32, device=dev)
        flow = self._flow(d // 2, dev)
        if n == 0 and self._bn_sync_world() > 1:
            raise ValueError("sync_batch_norm: a rank with an empty shard cannot take part in the cross-rank moments "
                             "(every rank must make the same sequence of all-reduce calls)")
        csr = csr_of(graph)
        with torch.cuda.device(dev):   # (the planners read the CURRENT device's CU count: size and launch on the same one)
            ws_bytes = lib.gnf_workspace_bytes(n, d, C.byref(flow))
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        sums = None
        if direction == _abi.GNF_FORWARD:
            # sums_out: caller-owned fp64 buffer whose first two slots receive [logdet, sum z^2]
            sums = sums_out if sums_out is not None else torch.empty(2, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _abi.check(lib.gnf_grevnet_from_f32(C.byref(csr.desc), C.byref(flow), _abi.ptr(src), src.stride(0) if n else d,
                                                _abi.ptr(out), d, d, direction, _abi.ptr(sums), _abi.ptr(ws), ws_bytes,
                                                _abi.stream_ptr(dev)), "gnf_grevnet_from_f32")
        return out, sums

    # ---- the reference's methods --------------------------------------------------------------
    def f(self, x, sums_out=None):
        """gnn.py:304-341: data -> latent.  Returns (GraphsTuple with nodes = z, log_det_jacobian)."""
        z, sums = self._run(x, _abi.GNF_FORWARD, sums_out)
        self.last_sums = sums
        self._last_z = z
        return x.replace(nodes=z), sums[0].to(torch.float32)

    def g(self, z):
        """gnn.py:343-373: latent -> data (sampling direction)."""
        x, _ = self._run(z, _abi.GNF_INVERSE)
        return z.replace(nodes=x)

    def log_prob(self, x):
        """gnn.py:375-377 with the prior the drivers use (standard normal, run_grevnet.py:292-294)."""
        from .flow import log_prob_terms
        return log_prob_terms(self, x)["log_prob_xs"]

    def _build(self, input, inverse=True):
        """gnn.py:379-381: inverse=True is the NORMALISING direction f; inverse=False is g."""
        func = self.f if inverse else self.g
        return func(input)

    __call__ = _build
