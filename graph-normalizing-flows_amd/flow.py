"""Log-prob assembly and sampling entry of the GRevNet drivers
(/root/reference/run_grevnet.py:290-311; same at train_grevnet_with_data.py:346-355, 397-414).

The two batch-wide reductions (sum(s) over all coupling half-steps, sum(z^2)) come out of
gnf_grevnet_f32 as device fp64 scalars; what is left here is scalar arithmetic, kept on the device
so that nothing synchronises until the caller reads a value.
"""
import ctypes as C
import math

import torch

from . import _abi

LN_2PI = math.log(2.0 * math.pi)


def gauss_sumsq(z):
    """sum_{n,j} z[n,j]^2 as a device fp64 0-d tensor (kernel D alone, gnf_gauss_sumsq_f32)."""
    lib = _abi.lib()
    if z.device.type != "cuda":
        raise _abi.GnfError("gauss_sumsq runs on a HIP device only (no CPU path)")
    z = z.to(torch.float32)
    if z.stride(1) != 1:
        z = z.contiguous()
    n, d = z.shape
    out = torch.empty(1, dtype=torch.float64, device=z.device)
    ws = torch.empty(8 * 1024, dtype=torch.uint8, device=z.device)
    with torch.cuda.device(z.device):
        _abi.check(lib.gnf_gauss_sumsq_f32(_abi.ptr(z), n, d, z.stride(0), _abi.ptr(out), _abi.ptr(ws),
                                           ws.numel(), _abi.stream_ptr(z.device)), "gnf_gauss_sumsq_f32")
    return out[0]


def forward_shard_sums(grevnet, graph, out=None):
    """Lean forward for throughput loops and multi-GPU shards: runs f and leaves
    [log_det_jacobian, sum(z^2), num_nodes] in a 3-element device fp64 tensor with NO further device
    work (that vector is exactly what one all-reduce sums over ranks; `log_prob_from_sums` finishes
    the arithmetic of run_grevnet.py:292-302 on the host).  Returns (z_nodes, sums3)."""
    n = graph.nodes.shape[0]
    if out is None:
        out = torch.zeros(3, dtype=torch.float64, device=graph.nodes.device)
        out[2] = float(n)
    z, _ = grevnet._run(graph, _abi.GNF_FORWARD, out)
    grevnet.last_sums = out
    return z, out


def log_prob_from_sums(sums3, d):
    """Host arithmetic of run_grevnet.py:292-302 on (all-reduced) [logdet, sum z^2, N] (python floats)."""
    logdet, sumsq, n = float(sums3[0]), float(sums3[1]), float(sums3[2])
    log_prob_zs = -0.5 * sumsq - 0.5 * d * LN_2PI * n
    log_prob_xs = log_prob_zs + logdet
    return {"log_prob_zs": log_prob_zs, "log_det_jacobian": logdet, "log_prob_xs": log_prob_xs,
            "total_loss": -log_prob_xs, "num_nodes": n, "loss_per_node": -log_prob_xs / n,
            "log_prob_xs_per_node": log_prob_xs / n, "log_prob_zs_per_node": log_prob_zs / n,
            "log_det_jacobian_per_node": logdet / n}


def log_prob_terms(grevnet, graph):
    """run_grevnet.py:290-302.  Returns a dict of 0-d device tensors (fp64 internally; the `*_f32`
    entries are the fp32 scalars the TF graph would log) plus the transformed graph:
      log_prob_zs = sum_n MVN(0,I).log_prob(z_n) = -0.5*sum(z^2) - D/2*ln(2pi)*N
      log_prob_xs = log_prob_zs + log_det_jacobian ; total_loss = -log_prob_xs ; *_per_node = * / sum(n_node)
    """
    z_graph, _ = grevnet(graph, inverse=True)
    sums = grevnet.last_sums                      # device fp64 [2]: logdet, sum z^2
    n, d = z_graph.nodes.shape
    logdet = sums[0]
    log_prob_zs = -0.5 * sums[1] - 0.5 * d * LN_2PI * n
    log_prob_xs = log_prob_zs + logdet
    num_nodes = float(n)                          # tf.cast(tf.reduce_sum(n_node), tf.float32)
    out = {
        "z_graph": z_graph,
        "log_det_jacobian": logdet,
        "log_prob_zs": log_prob_zs,
        "log_prob_xs": log_prob_xs,
        "total_loss": -log_prob_xs,
        "num_nodes": num_nodes,
        "loss_per_node": -log_prob_xs / num_nodes,
        "log_prob_xs_per_node": log_prob_xs / num_nodes,
        "log_prob_zs_per_node": log_prob_zs / num_nodes,
        "log_det_jacobian_per_node": logdet / num_nodes,
        # the three batch-wide sums a multi-GPU shard all-reduces
        "shard_sums": torch.stack([log_prob_zs, logdet, torch.tensor(num_nodes, dtype=torch.float64,
                                                                     device=sums.device)]),
    }
    return out


def sample(grevnet, graph, generator=None):
    """run_grevnet.py:304-311: z ~ N(0, I) of shape [sum(n_node), D]; x = grevnet(graph.replace(nodes=z),
    inverse=False).nodes; also MVN.log_prob(z) per node.  torch.randn is the sampler."""
    n, d = graph.nodes.shape
    z = torch.randn(n, d, dtype=torch.float32, device=graph.nodes.device, generator=generator)
    sample_log_prob = -0.5 * (z.double() ** 2).sum(dim=1) - 0.5 * d * LN_2PI
    top = grevnet(graph.replace(nodes=z), inverse=False)
    return {"sample": z, "sample_log_prob": sample_log_prob, "grevnet_top": top, "grevnet_top_nodes": top.nodes}


def scaled_hacky_sigmoid_l2(*_a, **_k):
    """Token for the distance function of loss.py:45-53 (the only one pred_adj is used with on this
    path); the arithmetic runs inside gnf_pred_adj_f32."""
    raise NotImplementedError("token only: pass it as distance_fn to pred_adj")


def pred_adj(gnn_output, distance_fn=scaled_hacky_sigmoid_l2, max_nodes_per_graph=None):
    """loss.py:154-159 for the sampling path (train_grevnet_with_data.py:415-416): edge probabilities
    sigmoid(10 * (1 - ||z_i - z_j||^2 / sqrt(D))) between the nodes of each graph, zero diagonal.  The
    reference returns a dense block-diagonal-masked [N, N] matrix; this returns the list of per-graph
    [n_g, n_g] blocks (views of one device buffer), which is what its consumer slices out
    (train_grevnet_with_data.py:538-540).  `adjacency = block > 0.5` gives the sampled graphs."""
    if distance_fn is not scaled_hacky_sigmoid_l2:
        raise NotImplementedError("pred_adj supports distance_fn=scaled_hacky_sigmoid_l2 (loss.py:45-53)")
    lib = _abi.lib()
    z = gnn_output.nodes
    if z.device.type != "cuda":
        raise _abi.GnfError("pred_adj runs on a HIP device only (no CPU path)")
    z = z.to(torch.float32)
    if z.stride(1) != 1:
        z = z.contiguous()
    n_node_host = gnn_output.n_node.cpu().tolist()          # sizes the output (the reference syncs here too)
    b = len(n_node_host)
    total = sum(n * n for n in n_node_host)
    largest = max(n_node_host) if b else 0
    cap = int(max_nodes_per_graph) if max_nodes_per_graph is not None else largest
    if cap < largest:   # the launch covers `cap` rows per graph: a smaller bound would leave blocks unwritten
        raise ValueError(f"max_nodes_per_graph={cap} is below the largest graph of the batch ({largest} nodes)")
    dev = z.device
    out = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
    off = torch.empty(b + 1, dtype=torch.int64, device=dev)
    ws_bytes = lib.gnf_pred_adj_workspace_bytes(b)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    nn = gnn_output.n_node.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        _abi.check(lib.gnf_pred_adj_f32(_abi.ptr(z), z.stride(0), z.shape[1], _abi.ptr(nn), b, cap, _abi.ptr(out),
                                        _abi.ptr(off), _abi.ptr(ws), ws_bytes, _abi.stream_ptr(dev)),
                   "gnf_pred_adj_f32")
    blocks, o = [], 0
    for n in n_node_host:
        blocks.append(out[o:o + n * n].view(n, n))
        o += n * n
    return blocks
