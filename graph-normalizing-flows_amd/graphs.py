"""GraphsTuple container + batching + CSR topology (host mirror of what the reference takes from
graph_nets: `gn.graphs.GraphsTuple`, `gn.utils_np.data_dicts_to_graphs_tuple`,
`gn.utils_np.networkxs_to_graphs_tuple`; used at /root/reference/grevnet_synthetic_data.py:45-47,
graph_data.py:122, train_grevnet_with_data.py:265-271).

Tensors are torch tensors (any device); indices are int32 as in graph_nets.  The receiver-sorted
CSR the kernels consume is derived data: it is built on the device by `gnf_build_csr` (or on the
host with a stable argsort for CPU-side tooling) and cached per (senders, receivers) pair so that
`graph.replace(nodes=...)` (gnn.py:307-308, run_grevnet.py:308) never rebuilds it.
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _abi

_FIELDS = ("nodes", "edges", "receivers", "senders", "globals", "n_node", "n_edge")


class GraphsTuple(collections.namedtuple("GraphsTuple", _FIELDS)):
    """Same field names and order as graph_nets' GraphsTuple; `.replace(**kw)` returns a copy."""
    __slots__ = ()

    def replace(self, **kwargs):
        return self._replace(**kwargs)

    def map(self, fn, fields=_FIELDS):
        return self._replace(**{k: fn(getattr(self, k)) for k in fields if getattr(self, k) is not None})

    def to(self, device):
        return self.map(lambda t: t.to(device) if isinstance(t, torch.Tensor) else t)


def data_dicts_to_graphs_tuple(data_dicts, device=None):
    """Concatenate per-graph dicts {nodes, senders, receivers[, n_node, n_edge, edges, globals]} into
    one block-diagonal batch, offsetting sender/receiver ids by the cumulative node count
    (what gn.utils_np.data_dicts_to_graphs_tuple does; grevnet_synthetic_data.py:28-47)."""
    nodes, senders, receivers, n_node, n_edge = [], [], [], [], []
    off = 0
    for d in data_dicts:
        x = np.asarray(d["nodes"], dtype=np.float32)
        n = int(d.get("n_node", x.shape[0]))
        s = np.asarray(d["senders"], dtype=np.int64)
        r = np.asarray(d["receivers"], dtype=np.int64)
        nodes.append(x)
        senders.append(s + off)
        receivers.append(r + off)
        n_node.append(n)
        n_edge.append(len(s))
        off += n
    cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros((0,), dt))
    e_total = int(sum(n_edge))
    # the unused all-zero fields are created where they will live (no 4 E-byte upload per batch for nothing)
    zdev = device if device is not None else "cpu"
    g = GraphsTuple(
        nodes=torch.from_numpy(np.concatenate(nodes, axis=0)) if nodes else torch.zeros(0, 0),
        edges=torch.zeros(e_total, dtype=torch.float32, device=zdev),          # unused zeros, as in the reference
        receivers=torch.from_numpy(cat(receivers, np.int32)),
        senders=torch.from_numpy(cat(senders, np.int32)),
        globals=torch.zeros(len(n_node), dtype=torch.float32, device=zdev),   # unused zeros
        n_node=torch.tensor(n_node, dtype=torch.int32),
        n_edge=torch.tensor(n_edge, dtype=torch.int32))
    return g.to(device) if device is not None else g


def check_graphs_tuple(graph):
    """Host-side structural check of a hand-built GraphsTuple (batches made by data_dicts_to_graphs_tuple are
    valid by construction): graph g's edges are the g-th n_edge slice of senders / receivers and both endpoints lie
    inside graph g's node range - the block-diagonal layout gnf_build_csr relies on (it indexes per-graph LDS
    histograms with `receiver - node_offset[g]` and does NOT re-check on the device).  Raises ValueError.
    Synchronises (copies the index tensors to the host): call it once per dataset, not per step."""
    n_node = graph.n_node.detach().cpu().numpy().astype(np.int64)
    n_edge = graph.n_edge.detach().cpu().numpy().astype(np.int64)
    s = graph.senders.detach().cpu().numpy().astype(np.int64)
    r = graph.receivers.detach().cpu().numpy().astype(np.int64)
    if n_node.sum() != graph.nodes.shape[0] or n_edge.sum() != len(s) or len(s) != len(r):
        raise ValueError(f"GraphsTuple: sum(n_node)={n_node.sum()} vs {graph.nodes.shape[0]} nodes, "
                         f"sum(n_edge)={n_edge.sum()} vs {len(s)} senders / {len(r)} receivers")
    lo = np.repeat(np.concatenate([[0], np.cumsum(n_node)[:-1]]), n_edge)
    hi = np.repeat(np.cumsum(n_node), n_edge)
    bad = (s < lo) | (s >= hi) | (r < lo) | (r >= hi)
    if bad.any():
        e = int(np.flatnonzero(bad)[0])
        raise ValueError(f"GraphsTuple: edge {e} ({int(s[e])} -> {int(r[e])}) leaves its graph's node range "
                         f"[{int(lo[e])}, {int(hi[e])}): {int(bad.sum())} such edges")
    return True


def graphs_tuple_from_edge_lists(n_node, n_edge, senders_local, receivers_local, graph_ids, nodes,
                                 device=None):
    """Batch graphs stored as concatenated LOCAL edge lists (the data/*.npz format)."""
    eoff = np.concatenate([[0], np.cumsum(n_edge)])
    noff = 0
    dicts = []
    for gid in graph_ids:
        lo, hi = int(eoff[gid]), int(eoff[gid + 1])
        n = int(n_node[gid])
        dicts.append({"nodes": nodes[noff:noff + n], "senders": senders_local[lo:hi],
                      "receivers": receivers_local[lo:hi], "n_node": n})
        noff += n
    return data_dicts_to_graphs_tuple(dicts, device)


# ----------------------------------------------------------------------------------------------
# CSR
# ----------------------------------------------------------------------------------------------
class Csr:
    """Receiver-sorted CSR on one device + the ctypes descriptor handed to the kernels."""

    def __init__(self, rowptr, col, n_nodes, n_edges):
        self.rowptr, self.col = rowptr, col
        self.n_nodes, self.n_edges = int(n_nodes), int(n_edges)
        self.desc = _abi.GnfCsr(_abi.ptr(rowptr).value, _abi.ptr(col).value if n_edges else 0,
                                self.n_nodes, self.n_edges)


def build_csr_host(senders, receivers, n_nodes):
    """numpy reference construction: stable sort of the edge list by receiver."""
    s = np.asarray(senders, dtype=np.int64)
    r = np.asarray(receivers, dtype=np.int64)
    order = np.argsort(r, kind="stable")
    col = s[order].astype(np.int32)
    rowptr = np.zeros(n_nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=n_nodes), out=rowptr[1:])
    return rowptr, col


def build_csr_device(graph, by_sender=False):
    """gnf_build_csr on the GraphsTuple's own device tensors (no host round trip).
    by_sender=True groups the same edges by SENDER (row u lists the receivers of u's out-edges): the
    transposed topology the backward pass of the aggregation needs (gnf_grevnet_backward_f32)."""
    lib = _abi.lib()
    dev = graph.senders.device
    if dev.type != "cuda":
        raise _abi.GnfError("build_csr_device needs the GraphsTuple on a HIP device; there is no CPU path")
    n = int(graph.nodes.shape[0])
    e = int(graph.senders.shape[0])
    b = int(graph.n_node.shape[0])
    snd = graph.senders.to(torch.int32).contiguous()
    rcv = graph.receivers.to(torch.int32).contiguous()
    if by_sender:
        snd, rcv = rcv, snd
    nn = graph.n_node.to(torch.int32).contiguous()
    ne = graph.n_edge.to(torch.int32).contiguous()
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    ws_bytes = lib.gnf_csr_workspace_bytes(b, n)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _abi.check(lib.gnf_build_csr(_abi.ptr(snd), _abi.ptr(rcv), _abi.ptr(nn), _abi.ptr(ne), b, n, e,
                                     _abi.ptr(rowptr), _abi.ptr(col), _abi.ptr(ws), ws_bytes,
                                     _abi.stream_ptr(dev)), "gnf_build_csr")
    return Csr(rowptr, col, n, e)


_CSR_CACHE = collections.OrderedDict()
_CSR_CACHE_MAX = 16


def csr_of(graph, by_sender=False):
    """CSR of a GraphsTuple, cached on the identity of its senders/receivers tensors: the index tensors of a
    batch must not be edited in place afterwards (make a new tensor, or call clear_csr_cache())."""
    key = (graph.senders.data_ptr(), graph.receivers.data_ptr(), int(graph.senders.shape[0]),
           int(graph.nodes.shape[0]), str(graph.senders.device), bool(by_sender))
    hit = _CSR_CACHE.get(key)
    if hit is not None:
        _CSR_CACHE.move_to_end(key)
        return hit[0]
    csr = build_csr_device(graph, by_sender)
    # keep the index tensors alive so the data_ptr key cannot be recycled while cached
    _CSR_CACHE[key] = (csr, graph.senders, graph.receivers)
    while len(_CSR_CACHE) > _CSR_CACHE_MAX:
        _CSR_CACHE.popitem(last=False)
    return csr


def clear_csr_cache():
    _CSR_CACHE.clear()
