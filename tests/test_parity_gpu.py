"""Parity tests proper: the HIP path (through the C ABI) vs the CPU oracle on identical inputs, and vs
the committed golden fixtures.  Run on a real MI355X:  python -m pytest tests -m gpu

Tolerances (stated, fp32 path vs float64 oracle):
  per-node log-prob  |delta| <= 1e-4        (BASELINE.json north_star)
  z elementwise      atol 2e-4 + rtol 2e-4  (fp32 rounding through 2*T*K chained GEMMs + exp)
  round trip         max|g(f(x)) - x| <= 5e-5 on the full-size configs (measured 2e-6 .. 6e-6; the CPU float32 restatement of
                     the reference has 2.4e-6 on the config-2 batch; tests/test_fullsize_gpu.py derives the bound from it)
  integer work (CSR) bit-exact
"""
import ctypes as C
import os
from functools import partial

import numpy as np
import pytest
import torch

from helpers import ATTN_GOLDEN_CASES, GOLDEN_CASES, graph_from_arrays, load_golden, make_product_grevnet
from oracle import gnf_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from gnf_amd import _abi
    _abi.lib()   # raises if libgnf_hip.so is missing: no silent fallback


def _oracle(hp, s, r, n):
    return O.Fp64Dense(s, r, n, agg=hp["agg"], combine=hp["combine"], epsilon=hp["epsilon"],
                       activation=hp["activation"])


def _batch(dataset, ids):
    n_node, n_edge, sl, rl = dataset
    return O.batch_graphs(n_node, n_edge, sl, rl, ids)


def _run_forward(net, graph):
    from gnf_amd.flow import log_prob_terms
    out = log_prob_terms(net, graph)
    torch.cuda.synchronize()
    return out


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
@pytest.mark.parametrize("name", GOLDEN_CASES + ATTN_GOLDEN_CASES)
def test_forward_matches_golden(name, fused):
    g = load_golden(name)
    net = make_product_grevnet(g, g["params"])
    net.fused = fused
    graph = graph_from_arrays(g["n_node"], g["n_edge"], g["senders"], g["receivers"], g["x"], DEV)
    out = _run_forward(net, graph)
    z = out["z_graph"].nodes.cpu().numpy()
    np.testing.assert_allclose(z, g["z"], atol=2e-4, rtol=2e-4)
    assert abs(float(out["log_det_jacobian"]) - float(g["logdet"])) <= 1e-4 * max(1.0, g["x"].shape[0])
    assert abs(float(out["log_prob_xs_per_node"]) - float(g["log_prob_xs_per_node"])) <= 1e-4
    # the input graph is untouched (TF ops are functional)
    np.testing.assert_array_equal(graph.nodes.cpu().numpy(), g["x"])
    # inverse: g(z) gives x back
    x_back = net(out["z_graph"], inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(x_back, g["x_roundtrip"], atol=5e-4, rtol=5e-4)


HP_DEFAULT = dict(D=64, latent=256, K=5, T=8, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
                  weight_sharing=False)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
def test_config2_shape_vs_oracle(community_medium, fused):
    """community_medium batch (reduced to 16 graphs so the fp64 dense oracle takes seconds), full
    BASELINE hyper-parameters: D=64, L=256, K=5, T=8, avg_then_mlp, leaky_relu."""
    hp = dict(HP_DEFAULT)
    rng = np.random.default_rng(12345)
    ids = rng.choice(168, size=16, replace=True)
    nn, ne, s, r = _batch(community_medium, ids)
    n = int(nn.sum())
    x = rng.standard_normal((n, hp["D"])).astype(np.float32)
    p = O.make_grevnet_params(99, hp["D"] // 2, hp["latent"], hp["K"], hp["T"], final_scale=0.25)
    ref = _oracle(hp, s, r, n).log_prob(x, p, hp["T"])
    net = make_product_grevnet(hp, p)
    net.fused = fused
    out = _run_forward(net, graph_from_arrays(nn, ne, s, r, x, DEV))
    assert abs(float(out["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= 1e-4
    assert abs(float(out["log_prob_zs_per_node"]) - ref["log_prob_zs_per_node"]) <= 1e-4
    assert abs(float(out["log_det_jacobian_per_node"]) - ref["log_det_jacobian_per_node"]) <= 1e-4
    np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), ref["z"], atol=5e-4, rtol=5e-4)


SHAPES = [
    # D, latent, K, T, agg, combine, eps, act, ws   (odd reference-default widths: D=2 run_grevnet.py:39,
    # D=100 run_gnn.py:111, D=200 train_grevnet_with_data.py:117; widths that are not multiples of 16)
    (2, 16, 3, 2, "mean", "agg", 1.0, "leaky_relu", False),
    (2, 256, 5, 3, "mean", "agg", 1.0, "leaky_relu", False),
    (100, 48, 2, 2, "sum", "agg", 0.5, "relu", False),
    (200, 40, 3, 1, "mean", "concat", 0.0, "relu", True),
    (6, 20, 1, 2, "sum", "concat", 0.0, "leaky_relu", False),     # K = 1: a single Linear layer
    (32, 128, 4, 4, "mean", "agg", 1.0, "leaky_relu", True),
    (16, 8, 8, 1, "mean", "agg", 1.0, "leaky_relu", False),       # K = GNF_MAX_LAYERS
    # train_grevnet_with_data.py:111-117 defaults (latent 2048, 3 layers, D = 200): too wide for the
    # LDS-resident kernel, both settings run the layered path's matrix-core GEMM
    (200, 2048, 3, 2, "mean", "agg", 1.0, "leaky_relu", False),
    (70, 1100, 2, 1, "sum", "concat", 0.0, "relu", False),        # ragged in every GEMM dimension
]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"D{s[0]}_L{s[1]}_K{s[2]}_T{s[3]}_{s[4]}_{s[5]}" for s in SHAPES])
def test_shape_generic(grid_small, shape, fused):
    d, latent, k, t, agg, combine, eps, act, ws = shape
    hp = dict(D=d, latent=latent, K=k, T=t, agg=agg, combine=combine, epsilon=eps, activation=act,
              weight_sharing=ws)
    nn, ne, s, r = _batch(grid_small, list(range(12)))          # all 12 graphs, ragged sizes 4..20
    n = int(nn.sum())
    rng = np.random.default_rng(d * 1000 + latent)
    x = rng.standard_normal((n, d)).astype(np.float32)
    p = O.make_grevnet_params(d + k, d // 2, latent, k, t, combine=combine, weight_sharing=ws,
                              final_scale=0.3 if agg == "mean" else 0.1)
    o = _oracle(hp, s, r, n)
    ref = o.log_prob(x, p, t, ws)
    net = make_product_grevnet(hp, p)
    net.fused = fused
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    out = _run_forward(net, graph)
    assert abs(float(out["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= 1e-4
    np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), ref["z"], atol=3e-4, rtol=3e-4)
    # inverse on an independent latent sample vs the oracle's g
    zs = rng.standard_normal((n, d)).astype(np.float32)
    xg = net(graph.replace(nodes=torch.as_tensor(zs).to(DEV)), inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(xg, o.g(zs, p, t, ws), atol=3e-4, rtol=3e-4)


def test_fully_connected_topology(community_medium):
    """train_grevnet_with_data.py topology: every ordered pair incl. self (utils.py:164-183)."""
    from gnf_amd.datasets import senders_receivers
    n_node = community_medium[0][[3, 50, 77]]
    s, r, ne = senders_receivers(n_node)
    n = int(n_node.sum())
    hp = dict(HP_DEFAULT, D=16, latent=64, K=3, T=2)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, 16)).astype(np.float32)
    p = O.make_grevnet_params(5, 8, 64, 3, 2, final_scale=0.3)
    ref = _oracle(hp, s, r, n).log_prob(x, p, 2)
    out = _run_forward(make_product_grevnet(hp, p), graph_from_arrays(n_node, ne, s, r, x, DEV))
    assert abs(float(out["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= 1e-4


def test_isolated_nodes_and_missing_self_loops():
    """Nodes without incoming edges: empty segment -> 0 for sum AND mean (unsorted_segment_mean
    divides by max(count, 1))."""
    s = np.array([0, 0, 2], np.int32)
    r = np.array([1, 2, 1], np.int32)          # node 0 and node 3 receive nothing
    n = 4
    for agg in ("sum", "mean"):
        hp = dict(HP_DEFAULT, D=4, latent=8, K=2, T=2, agg=agg)
        x = np.random.default_rng(1).standard_normal((n, 4)).astype(np.float32)
        p = O.make_grevnet_params(1, 2, 8, 2, 2, final_scale=0.5)
        ref = _oracle(hp, s, r, n).log_prob(x, p, 2)
        for fused in (True, False):
            net = make_product_grevnet(hp, p)
            net.fused = fused
            out = _run_forward(net, graph_from_arrays([4], [3], s, r, x, DEV))
            assert abs(float(out["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= 1e-4
            np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), ref["z"], atol=1e-4, rtol=1e-4)


def test_empty_batch():
    hp = dict(HP_DEFAULT, D=4, latent=8, K=2, T=1)
    net = make_product_grevnet(hp, O.make_grevnet_params(1, 2, 8, 2, 1))
    g = graph_from_arrays(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32),
                          np.zeros(0, np.int32), np.zeros((0, 4), np.float32), DEV)
    zg, ld = net(g, inverse=True)
    torch.cuda.synchronize()
    assert zg.nodes.shape == (0, 4) and float(ld) == 0.0
    assert float(net.last_sums[1]) == 0.0


def test_additivity_over_shards_on_device(community_medium):
    """The property multi-GPU sharding relies on: batch sums = sum of per-shard sums."""
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.sharding import assemble_from_sums, shard_graph_ids
    hp = dict(HP_DEFAULT, D=16, latent=32, K=3, T=2)
    p = O.make_grevnet_params(3, 8, 32, 3, 2, final_scale=0.4)
    rng = np.random.default_rng(2)
    ids = rng.choice(168, size=12, replace=True)
    n_node, n_edge, sl, rl = community_medium
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, ids)
    n = int(nn.sum())
    x = rng.standard_normal((n, 16)).astype(np.float32)
    net = make_product_grevnet(hp, p)
    full = log_prob_terms(net, graph_from_arrays(nn, ne, s, r, x, DEV))
    noff = np.concatenate([[0], np.cumsum(nn)])
    total = torch.zeros(3, dtype=torch.float64, device=DEV)
    for shard in shard_graph_ids(nn, ne, 3):
        rows = np.concatenate([np.arange(noff[i], noff[i + 1]) for i in shard])
        n2, e2, s2, r2 = O.batch_graphs(n_node, n_edge, sl, rl, ids[shard])
        total += log_prob_terms(net, graph_from_arrays(n2, e2, s2, r2, x[rows], DEV))["shard_sums"]
    asm = assemble_from_sums(total)
    assert abs(float(asm["log_prob_xs_per_node"]) - float(full["log_prob_xs_per_node"])) <= 1e-6
    ref = _oracle(hp, s, r, n).log_prob(x, p, 2)
    assert abs(float(asm["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= 1e-4


def test_round_trip_at_full_size(community_medium):
    """Size-independent property at BASELINE config 2's full size (batch 64, T=8, D=64, L=256, K=5):
    g(f(x)) = x, f(g(z)) = z; and the fused and layered kernels agree with each other."""
    hp = dict(HP_DEFAULT)
    rng = np.random.default_rng(12345)
    ids = rng.choice(168, size=64, replace=True)
    nn, ne, s, r = _batch(community_medium, ids)
    n = int(nn.sum())
    x = rng.standard_normal((n, 64)).astype(np.float32)
    p = O.make_grevnet_params(99, 32, 256, 5, 8, final_scale=0.25)
    net = make_product_grevnet(hp, p)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    zg, ld = net(graph, inverse=True)
    back = net(zg, inverse=False).nodes
    assert float((back - graph.nodes).abs().max()) <= 5e-5
    z2, _ = net(net(graph, inverse=False), inverse=True)
    assert float((z2.nodes - graph.nodes).abs().max()) <= 5e-5
    lay = make_product_grevnet(hp, p)
    lay.fused = False
    zl, ldl = lay(graph, inverse=True)
    torch.cuda.synchronize()
    assert abs(float(ld) - float(ldl)) / n <= 2e-6
    assert float((zl.nodes - zg.nodes).abs().max()) <= 5e-5
    # reproducibility: bitwise identical on a re-run (fixed-order reductions, no atomics)
    zg2, ld2 = net(graph, inverse=True)
    assert torch.equal(zg2.nodes, zg.nodes) and float(ld2) == float(ld)


# ------------------------------------------------------------------------------------------------
# single entry points
# ------------------------------------------------------------------------------------------------
def test_device_csr_build_is_bit_exact(community_medium, grid_small):
    from gnf_amd.graphs import build_csr_device, build_csr_host
    from gnf_amd.datasets import senders_receivers
    cases = []
    rng = np.random.default_rng(0)
    nn, ne, s, r = _batch(community_medium, rng.choice(210, size=40, replace=True))
    cases.append((nn, ne, s, r))
    cases.append(_batch(grid_small, list(range(12))))
    n_node = np.array([100, 1, 37], np.int32)                 # FC incl. a 1-node graph; 10^4-edge graph
    s2, r2, ne2 = senders_receivers(n_node)
    cases.append((n_node, ne2, s2.astype(np.int32), r2.astype(np.int32)))
    # shuffled edge order inside each graph (unsorted segment ids) + a graph with zero edges
    nn3 = np.array([5, 3, 4], np.int32)
    ne3 = np.array([7, 0, 5], np.int32)
    s3 = np.array([4, 0, 2, 2, 1, 3, 0, 8, 11, 9, 10, 8], np.int32)
    r3 = np.array([1, 1, 0, 4, 1, 3, 0, 10, 8, 8, 8, 11], np.int32)
    cases.append((nn3, ne3, s3, r3))
    # beyond the LDS budget of the fast path: one graph with > 16384 edges (FC, n = 130) and one with
    # > 2048 nodes (a 2100-node ring, both directions + self loops, edge order shuffled)
    s4, r4, ne4 = senders_receivers(np.array([130, 3], np.int32))
    cases.append((np.array([130, 3], np.int32), ne4, s4.astype(np.int32), r4.astype(np.int32)))
    m = 2100
    ring_s = np.concatenate([np.arange(m), np.arange(m), (np.arange(m) + 1) % m])
    ring_r = np.concatenate([np.arange(m), (np.arange(m) + 1) % m, np.arange(m)])
    perm = np.random.default_rng(9).permutation(len(ring_s))
    cases.append((np.array([m], np.int32), np.array([len(ring_s)], np.int32),
                  ring_s[perm].astype(np.int32), ring_r[perm].astype(np.int32)))
    # rows longer than a wave (rank pass in several strides) with the edge list shuffled and 300 duplicated edges
    s5, r5, _ = senders_receivers(np.array([90], np.int32))
    dup = np.random.default_rng(4).integers(0, len(s5), size=300)
    s5, r5 = np.concatenate([s5, s5[dup]]), np.concatenate([r5, r5[dup]])
    perm5 = np.random.default_rng(5).permutation(len(s5))
    cases.append((np.array([90], np.int32), np.array([len(s5)], np.int32), s5[perm5].astype(np.int32),
                  r5[perm5].astype(np.int32)))
    for nn, ne, s, r in cases:
        n = int(nn.sum())
        g = graph_from_arrays(nn, ne, s, r, np.zeros((n, 2), np.float32), DEV)
        csr = build_csr_device(g)
        torch.cuda.synchronize()
        rowptr, col = build_csr_host(s, r, n)
        assert np.array_equal(csr.rowptr.cpu().numpy(), rowptr)
        assert np.array_equal(csr.col.cpu().numpy()[:len(s)], col)


@pytest.mark.parametrize("agg", ["sum", "mean"])
@pytest.mark.parametrize("h", [1, 32, 50, 300])
def test_aggregate_kernel_alone(community_medium, agg, h):
    from gnf_amd import _abi
    from gnf_amd.graphs import csr_of
    nn, ne, s, r = _batch(community_medium, [1, 2, 3, 200])
    n = int(nn.sum())
    x = np.random.default_rng(h).standard_normal((n, h)).astype(np.float32)
    g = graph_from_arrays(nn, ne, s, r, x, DEV)
    csr = csr_of(g)
    out = torch.empty(n, h, device=DEV)
    _abi.check(_abi.lib().gnf_aggregate_f32(C.byref(csr.desc), _abi.ptr(g.nodes), h, h,
                                            _abi.GNF_AGG_MEAN if agg == "mean" else _abi.GNF_AGG_SUM,
                                            _abi.ptr(out), h, _abi.stream_ptr()), "gnf_aggregate_f32")
    o = O.Fp64Dense(s, r, n, agg=agg, epsilon=0.0)
    want = o.adj @ x.astype(np.float64)
    if agg == "mean":
        want = want / o.deg
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=1e-5, rtol=1e-5)


def _hub_graphs(seed):
    """Graphs that exercise kernel A's hub-row path (gnf_layered.hip): hubs of 40 - 400 in-edges, a slice of 256 rows with
    more long rows than a front workgroup takes (the rest stay with the regular waves), a complete graph (a DENSE slice: more
    than 16 long rows, left alone), a long row in the batch's last, partial slice, duplicate edges."""
    rng = np.random.default_rng(seed)
    sizes, S, R = [], [], []
    def graph(n, hubs):
        s = list(range(n)) + [int(v) for v in rng.integers(0, n, 2 * n)]
        r = list(range(n)) + [int(v) for v in rng.integers(0, n, 2 * n)]
        for hub, deg in hubs:
            src = rng.integers(0, n, deg)
            s += [int(v) for v in src]
            r += [hub] * deg
        sizes.append(n), S.append(np.array(s, np.int32)), R.append(np.array(r, np.int32))
    graph(300, [(0, 400), (7, 41)])                                   # one hub far beyond a chunk, one just over the line
    graph(200, [(i, 33 + 5 * i) for i in range(9)])                   # 9 long rows in one slice: the front takes the first 4
    n_c = 40                                                          # complete graph: every row has 40 > 32 edges
    sizes.append(n_c), S.append(np.repeat(np.arange(n_c, dtype=np.int32), n_c)), R.append(np.tile(np.arange(n_c, dtype=np.int32), n_c))
    graph(75, [(74, 120)])                                            # the batch's last row is a hub
    nn = np.array(sizes, np.int32)
    ne = np.array([len(v) for v in S], np.int32)
    off = np.concatenate([[0], np.cumsum(nn)[:-1]]).astype(np.int32)
    return nn, ne, np.concatenate([v + o for v, o in zip(S, off)]), np.concatenate([v + o for v, o in zip(R, off)])


@pytest.mark.parametrize("agg", ["sum", "mean"])
@pytest.mark.parametrize("h", [3, 8, 32, 128, 300])
def test_aggregate_kernel_hub_rows(agg, h):
    """Rows of more than 32 edges: front workgroups, edges split over the lane groups, partial sums in group order
    (every lane width: scalar lanes for H = 3, 2 .. 64 lanes per row, several feature slices per lane for H = 300)."""
    from gnf_amd import _abi
    from gnf_amd.graphs import csr_of
    nn, ne, s, r = _hub_graphs(h)
    n = int(nn.sum())
    x = np.random.default_rng(h).standard_normal((n, h)).astype(np.float32)
    g = graph_from_arrays(nn, ne, s, r, x, DEV)
    csr = csr_of(g)
    out = torch.full((n, h), float("nan"), device=DEV)
    _abi.check(_abi.lib().gnf_aggregate_f32(C.byref(csr.desc), _abi.ptr(g.nodes), h, h,
                                            _abi.GNF_AGG_MEAN if agg == "mean" else _abi.GNF_AGG_SUM,
                                            _abi.ptr(out), h, _abi.stream_ptr()), "gnf_aggregate_f32")
    deg = np.bincount(r, minlength=n).astype(np.float64)
    want = np.zeros((n, h))
    np.add.at(want, r, x.astype(np.float64)[s])
    if agg == "mean":
        want = want / np.maximum(deg, 1.0)[:, None]
    got = out.cpu().numpy()
    assert not np.isnan(got).any()                                     # every row written exactly by someone
    np.testing.assert_allclose(got, want, atol=3e-5 * (1.0 if agg == "mean" else 20.0), rtol=1e-5)
    out2 = torch.empty_like(out)                                       # deterministic: a second launch is bitwise the first
    _abi.check(_abi.lib().gnf_aggregate_f32(C.byref(csr.desc), _abi.ptr(g.nodes), h, h,
                                            _abi.GNF_AGG_MEAN if agg == "mean" else _abi.GNF_AGG_SUM,
                                            _abi.ptr(out2), h, _abi.stream_ptr()), "gnf_aggregate_f32")
    assert torch.equal(out, out2)


@pytest.mark.parametrize("combine", ["agg", "concat"])
def test_gnn_module_on_hub_graphs(combine):
    """The eps * x + agg and [x || agg] forms of kernel A (what the large-batch kernel and the layered path read) on the
    hub graphs, through a make_gnn_fn() product with layers too wide for the fused kernels (the layered path)."""
    from gnf_amd import gnn
    nn, ne, s, r = _hub_graphs(11)
    n = int(nn.sum())
    x = np.random.default_rng(5).standard_normal((n, 12)).astype(np.float32)
    in0 = 24 if combine == "concat" else 12
    layers = O.make_mlp_params(np.random.default_rng(6), in0, 1100, 12, 2)
    mk = partial(gnn.make_mlp_model, 1100, 12, 2, gnn.leaky_relu)
    mod = gnn.avg_concat_then_mlp_gnn(mk) if combine == "concat" else gnn.avg_then_mlp_gnn(mk, 0.5)
    mod._node_block._mlp.set_params(layers)
    out = mod(graph_from_arrays(nn, ne, s, r, x, DEV))
    want = O.Fp64Dense(s, r, n, agg="mean", combine=combine, epsilon=0.5).gnn(x.astype(np.float64), layers)
    np.testing.assert_allclose(out.nodes.cpu().numpy(), want, atol=2e-4, rtol=2e-4)


def test_gnn_module_call_alone(grid_small):
    """A make_gnn_fn() product called like the reference calls it: module(GraphsTuple) -> GraphsTuple."""
    from gnf_amd import gnn
    nn, ne, s, r = _batch(grid_small, [6, 2])
    n = int(nn.sum())
    x = np.random.default_rng(3).standard_normal((n, 5)).astype(np.float32)
    layers = O.make_mlp_params(np.random.default_rng(4), 10, 24, 7, 3)
    mod = gnn.sum_concat_then_mlp_gnn(partial(gnn.make_mlp_model, 24, 7, 3, gnn.leaky_relu))
    mod._node_block._mlp.set_params(layers)
    out = mod(graph_from_arrays(nn, ne, s, r, x, DEV))
    want = O.Fp64Dense(s, r, n, agg="sum", combine="concat").gnn(x.astype(np.float64), layers)
    np.testing.assert_allclose(out.nodes.cpu().numpy(), want, atol=1e-4, rtol=1e-4)
    assert out.senders is not None and out.nodes.shape == (n, 7)


def test_coupling_half_entry_point_accumulates_logdet(grid_small):
    from gnf_amd import _abi
    from gnf_amd.graphs import csr_of
    nn, ne, s, r = _batch(grid_small, [6])
    n = int(nn.sum())
    hp = dict(HP_DEFAULT, D=8, latent=16, K=2, T=1)
    p = O.make_grevnet_params(8, 4, 16, 2, 1, final_scale=0.5)
    x = np.random.default_rng(8).standard_normal((n, 8)).astype(np.float32)
    net = make_product_grevnet(hp, p)
    g = graph_from_arrays(nn, ne, s, r, x, DEV)
    flow = net._flow(4, torch.device(DEV))
    csr = csr_of(g)
    lib = _abi.lib()
    buf = g.nodes.clone()
    acc = torch.full((1,), 10.0, dtype=torch.float64, device=DEV)     # pre-loaded: must be ADDED to
    ws_bytes = lib.gnf_workspace_bytes(n, 8, C.byref(flow))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    spec = flow.gnn
    _abi.check(lib.gnf_coupling_half_f32(C.byref(csr.desc), C.byref(flow.s_nets[0]), C.byref(flow.t_nets[0]),
                                         C.byref(spec), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 16),
                                         8, 4, 0, _abi.ptr(acc), _abi.ptr(ws), ws_bytes, _abi.stream_ptr()),
               "gnf_coupling_half_f32")
    o = _oracle(hp, s, r, n)
    x0, x1 = x[:, :4].astype(np.float64), x[:, 4:].astype(np.float64)
    sv = o.gnn(x0, p["s"][0][0])
    tv = o.gnn(x0, p["t"][0][0])
    got = buf.cpu().numpy()
    np.testing.assert_allclose(got[:, 4:], x1 * np.exp(sv) + tv, atol=1e-5, rtol=1e-5)
    np.testing.assert_array_equal(got[:, :4], x[:, :4])               # conditioning half untouched
    assert abs(float(acc[0]) - (10.0 + sv.sum())) < 1e-4
    # and the inverse direction restores the input
    _abi.check(lib.gnf_coupling_half_f32(C.byref(csr.desc), C.byref(flow.s_nets[0]), C.byref(flow.t_nets[0]),
                                         C.byref(spec), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 16),
                                         8, 4, 1, None, _abi.ptr(ws), ws_bytes, _abi.stream_ptr()),
               "gnf_coupling_half_f32")
    np.testing.assert_allclose(buf.cpu().numpy(), x, atol=1e-5, rtol=1e-5)


def test_gauss_sumsq_kernel_alone():
    from gnf_amd.flow import gauss_sumsq
    z = np.random.default_rng(6).standard_normal((1234, 10)).astype(np.float32)
    got = float(gauss_sumsq(torch.as_tensor(z).to(DEV)))
    assert abs(got - float((z.astype(np.float64) ** 2).sum())) < 1e-8 * got
    zt = torch.as_tensor(z).to(DEV)[:, 2:8]                            # strided view (ld = 10, D = 6)
    assert abs(float(gauss_sumsq(zt)) - float((z[:, 2:8].astype(np.float64) ** 2).sum())) < 1e-6


def test_sampling_entry(grid_small):
    from gnf_amd.flow import sample
    nn, ne, s, r = _batch(grid_small, [6, 7])
    n = int(nn.sum())
    hp = dict(HP_DEFAULT, D=8, latent=16, K=3, T=2)
    p = O.make_grevnet_params(2, 4, 16, 3, 2, final_scale=0.5)
    net = make_product_grevnet(hp, p)
    gen = torch.Generator(device=DEV).manual_seed(1)
    out = sample(net, graph_from_arrays(nn, ne, s, r, np.zeros((n, 8), np.float32), DEV), generator=gen)
    z = out["sample"].cpu().numpy()
    want = _oracle(hp, s, r, n).g(z, p, 2)
    np.testing.assert_allclose(out["grevnet_top_nodes"].cpu().numpy(), want, atol=2e-4, rtol=2e-4)
    from scipy.stats import multivariate_normal
    np.testing.assert_allclose(out["sample_log_prob"].cpu().numpy(),
                               multivariate_normal(np.zeros(8), np.eye(8)).logpdf(z), atol=1e-5)


# ------------------------------------------------------------------------------------------------
# edge-list attention GNN (DMSelfAttentionMLP, gnn.py:385-553): the reference drivers' default make_gnn_fn
# ------------------------------------------------------------------------------------------------
ATTN_SHAPES = [
    # D, latent, K, T, heads, kq, v, C, concat, kq_div, residual, ws
    (64, 256, 5, 2, 8, 10, 10, 80, True, False, False, False),     # reference defaults (run_grevnet.py:59-80) at D=64
    (2, 32, 3, 2, 8, 10, 10, 80, True, False, False, False),       # node_embedding_dim default 2 -> H = 1
    (20, 48, 2, 1, 3, 7, 5, 20, False, True, True, True),          # no concat, scaled logits, residual, shared
    (12, 16, 1, 2, 1, 32, 32, 9, True, True, False, False),        # single head at the kq / v limits, K = 1
]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
@pytest.mark.parametrize("shape", ATTN_SHAPES, ids=[f"D{s[0]}_L{s[1]}_h{s[4]}_kq{s[5]}_v{s[6]}_C{s[7]}" for s in ATTN_SHAPES])
def test_attention_gnn_vs_oracle(grid_small, community_medium, shape, fused):
    d, latent, k, t, nh, kq, vd, c, concat, div, res, ws = shape
    akw = dict(num_heads=nh, kq_dim=kq, v_dim=vd, out_dim=c, concat=concat, kq_dim_division=div, residual=res)
    hp = dict(D=d, latent=latent, K=k, T=t, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=ws, attn=akw)
    nn, ne, s, r = _batch(grid_small, list(range(12))) if d != 64 else _batch(community_medium, [3, 77, 150, 9])
    n = int(nn.sum())
    rng = np.random.default_rng(d * 100 + nh)
    x = (rng.standard_normal((n, d)) * (0.3 if res else 1.0)).astype(np.float32)
    p = O.make_attn_grevnet_params(d + nh, d // 2, latent, k, t, weight_sharing=ws, final_scale=0.3, **akw)
    o = O.Fp64Dense(s, r, n, activation="relu")
    ref = o.log_prob(x, p, t, ws)
    net = make_product_grevnet(hp, p)
    net.fused = fused
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    out = _run_forward(net, graph)
    assert abs(float(out["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= 1e-4
    np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), ref["z"], atol=3e-4, rtol=3e-4)
    zs = (rng.standard_normal((n, d)) * (0.3 if res else 1.0)).astype(np.float32)
    xg = net(graph.replace(nodes=torch.as_tensor(zs).to(DEV)), inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(xg, o.g(zs, p, t, ws), atol=3e-4, rtol=3e-4)


ATTN_LN_SHAPES = [
    # D, latent, K, T, heads, kq, v, C, concat, residual, ws          (all layer_norm=True, gnn.py:550-552)
    (64, 256, 5, 2, 8, 10, 10, 80, True, True, False),     # --attn_layer_norm --attn_residual on the reference defaults
    (20, 48, 2, 1, 3, 7, 5, 20, False, False, True),       # no concat, no residual, shared nets
    (2, 32, 3, 2, 8, 10, 10, 80, True, True, False),       # H = 1: one feature -> variance 0 -> s = t = ln_beta
    (300, 64, 2, 1, 2, 4, 4, 8, True, True, False),        # H = 150 > two wave widths per row
]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
@pytest.mark.parametrize("shape", ATTN_LN_SHAPES, ids=[f"D{s[0]}_L{s[1]}_h{s[4]}_C{s[7]}" for s in ATTN_LN_SHAPES])
def test_attention_layer_norm_vs_oracle(grid_small, community_medium, shape, fused):
    """DMSelfAttentionMLP(layer_norm=True) (run_grevnet.py:80 --attn_layer_norm): forward, log-det and sampling
    direction vs the fp64 oracle, through the fused MLP kernel (one net per workgroup + normalisation + coupling
    launches) and through the layered path."""
    d, latent, k, t, nh, kq, vd, c, concat, res, ws = shape
    akw = dict(num_heads=nh, kq_dim=kq, v_dim=vd, out_dim=c, concat=concat, kq_dim_division=False, residual=res,
               layer_norm=True)
    hp = dict(D=d, latent=latent, K=k, T=t, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=ws, attn=akw)
    nn, ne, s, r = _batch(grid_small, list(range(12))) if d != 64 else _batch(community_medium, [3, 77, 150, 9])
    n = int(nn.sum())
    rng = np.random.default_rng(d * 100 + nh)
    x = rng.standard_normal((n, d)).astype(np.float32)
    p = O.make_attn_grevnet_params(d + nh, d // 2, latent, k, t, weight_sharing=ws, final_scale=0.3, **akw)
    o = O.Fp64Dense(s, r, n, activation="relu")
    ref = o.log_prob(x, p, t, ws)
    net = make_product_grevnet(hp, p)
    net.fused = fused
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    out = _run_forward(net, graph)
    # the normalised s is O(1) per feature, so |z| grows to ~e^3 over the flow: the bound is the error the fp32 CPU
    # restatement makes on the same inputs (x 6) next to the usual 1e-4
    o32 = O.Fp32Gather(s, r, n, activation="relu")
    r32 = o32.log_prob(o32.to_t(x), o32.prep_params(p), t, ws)
    tol = 1e-4 + 6.0 * abs(r32["log_prob_xs_per_node"] - ref["log_prob_xs_per_node"])
    assert abs(float(out["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= tol
    np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), ref["z"], atol=3e-4, rtol=3e-4)
    zs = rng.standard_normal((n, d)).astype(np.float32)
    xg = net(graph.replace(nodes=torch.as_tensor(zs).to(DEV)), inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(xg, o.g(zs, p, t, ws), atol=3e-4, rtol=3e-4)
    back = net(graph.replace(nodes=torch.as_tensor(xg).to(DEV)), inverse=True)[0].nodes.cpu().numpy()
    np.testing.assert_allclose(back, zs, atol=3e-4, rtol=3e-4)


def test_attention_module_with_layer_norm_called_alone():
    """module(GraphsTuple) with layer_norm=True and a free MLP output width (7): the normalisation runs over the
    MLP's output width, not H; first-connect initialisation is gamma = 1, beta = 0 (snt.LayerNorm)."""
    from gnf_amd import gnn
    s = np.array([0, 0, 2, 1], np.int32)
    r = np.array([1, 2, 1, 1], np.int32)
    n, h = 4, 6
    x = np.random.default_rng(0).standard_normal((n, h)).astype(np.float32)
    net = O.make_attn_net_params(np.random.default_rng(1), h, 16, 2, num_heads=4, kq_dim=3, v_dim=2, out_dim=5)
    net["mlp"] = O.make_mlp_params(np.random.default_rng(2), h + 5, 16, 7, 2)
    mod = gnn.dm_self_attn_gnn(kq_dim=3, v_dim=2, make_mlp_fn=partial(gnn.make_mlp_model, 16, 7, 2, gnn.relu),
                               num_heads=4, concat_heads_output_dim=5, layer_norm=True)
    graph = graph_from_arrays([4], [4], s, r, x, DEV)
    mod(graph)                                           # first connection creates the variables
    np.testing.assert_array_equal(mod.attn_params["ln_gamma"].cpu().numpy(), np.ones(7, np.float32))
    np.testing.assert_array_equal(mod.attn_params["ln_beta"].cpu().numpy(), np.zeros(7, np.float32))
    net["attn"].update(layer_norm=True, ln_gamma=np.linspace(0.5, 1.5, 7).astype(np.float32),
                       ln_beta=np.linspace(-0.3, 0.3, 7).astype(np.float32))
    mod.set_attn_params(net["attn"])
    mod._mlp.set_params(net["mlp"])
    out = mod(graph)
    want = O.Fp64Dense(s, r, n, activation="relu").gnn(x.astype(np.float64), net)
    np.testing.assert_allclose(out.nodes.cpu().numpy(), want, atol=1e-4, rtol=1e-4)


def test_attention_module_call_alone_and_isolated_nodes():
    """module(GraphsTuple) -> GraphsTuple for a dm_self_attn_gnn product; a node without incoming edges
    gets attended value 0 (gnn.py:403)."""
    from gnf_amd import gnn
    s = np.array([0, 0, 2, 1], np.int32)
    r = np.array([1, 2, 1, 1], np.int32)          # nodes 0 and 3 receive nothing; node 1 has 3 incoming
    n, h = 4, 6
    x = np.random.default_rng(0).standard_normal((n, h)).astype(np.float32)
    net = O.make_attn_net_params(np.random.default_rng(1), h, 16, 2, num_heads=4, kq_dim=3, v_dim=2, out_dim=5)
    net["mlp"] = O.make_mlp_params(np.random.default_rng(2), h + 5, 16, 7, 2)     # free output width
    mod = gnn.dm_self_attn_gnn(kq_dim=3, v_dim=2, make_mlp_fn=partial(gnn.make_mlp_model, 16, 7, 2, gnn.relu),
                               num_heads=4, concat_heads_output_dim=5)
    mod.set_attn_params(net["attn"])
    mod._mlp.set_params(net["mlp"])
    out = mod(graph_from_arrays([4], [4], s, r, x, DEV))
    want = O.Fp64Dense(s, r, n, activation="relu").gnn(x.astype(np.float64), net)
    np.testing.assert_allclose(out.nodes.cpu().numpy(), want, atol=1e-4, rtol=1e-4)


def test_pred_adj_decoder_after_sampling(community_medium):
    """pred_adj(grevnet(sample, inverse=False), scaled_hacky_sigmoid_l2) -> per-graph edge
    probabilities (loss.py:45-53,131-159; train_grevnet_with_data.py:397-416), incl. a 1-node graph and a
    launch bound (max_nodes_per_graph) larger than any graph; d = 1100: rows wider than the kernel's LDS tile."""
    from gnf_amd.flow import pred_adj, scaled_hacky_sigmoid_l2
    rng = np.random.default_rng(4)
    n_node = np.array([17, 1, 40, 33], np.int32)
    n = int(n_node.sum())
    for d in (2, 64, 200, 1100):
        z = (rng.standard_normal((n, d)) * (0.7 if d < 1000 else 0.12)).astype(np.float32)
        g = graph_from_arrays(n_node, np.zeros(4, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), z, DEV)
        for cap in (None, 64):
            blocks = pred_adj(g, distance_fn=scaled_hacky_sigmoid_l2, max_nodes_per_graph=cap)
            want = O.pred_adj_blocks(z, n_node)
            assert [tuple(b.shape) for b in blocks] == [(17, 17), (1, 1), (40, 40), (33, 33)]
            for b, w in zip(blocks, want):
                np.testing.assert_allclose(b.cpu().numpy(), w, atol=2e-5, rtol=1e-4)
                assert float(torch.diagonal(b).abs().max()) == 0.0
    # thresholded adjacency is symmetric (what train_grevnet_with_data.py:532-540 turns into graphs)
    adj = (blocks[2] > 0.5)
    assert bool((adj == adj.T).all())


# ---- batch-norm bijector (TFP-0.7 semantics restated, unpinned) ---------------------------
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
def test_batch_norm_flow_matches_golden(fused):
    from helpers import BN_GOLDEN_CASES
    g = load_golden(BN_GOLDEN_CASES[0])
    hp = {k: g[k] for k in ("D", "latent", "K", "T", "agg", "combine", "epsilon", "activation", "weight_sharing")}
    net = make_product_grevnet(hp, g["params"])
    assert net.use_batch_norm
    net.fused = fused
    graph = graph_from_arrays(g["n_node"], g["n_edge"], g["senders"], g["receivers"], g["x"], DEV)
    out = _run_forward(net, graph)
    n = g["x"].shape[0]
    np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), g["z"], atol=3e-4, rtol=3e-4)
    assert abs(float(out["log_det_jacobian"]) - float(g["logdet"])) <= 1e-4 * n
    assert abs(float(out["log_prob_xs_per_node"]) - float(g["log_prob_xs_per_node"])) <= 1e-4
    # the batch moments every bijector saw come back for the moving-average update
    for half in range(2):
        for i in range(g["T"]):
            bn = net.bns[half][i]
            np.testing.assert_allclose(bn.batch_mean.cpu().numpy(), g[f"bn_{half}_{i}_batch_mean"], atol=2e-5)
            np.testing.assert_allclose(bn.batch_variance.cpu().numpy(), g[f"bn_{half}_{i}_batch_variance"],
                                       rtol=2e-5, atol=2e-5)
    # sampling direction: de-normalisation with the MOVING statistics
    x_back = net(out["z_graph"], inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(x_back, g["x_roundtrip"], atol=1e-3, rtol=1e-3)
    # input untouched
    np.testing.assert_array_equal(graph.nodes.cpu().numpy(), g["x"])


def test_batch_norm_round_trip_after_moving_average_converges(community_medium):
    """With the moving statistics set to the batch moments g(f(x)) = x; update_moving_statistics moves them
    there geometrically (momentum 0.99), as UPDATE_OPS does during training."""
    hp = dict(HP_DEFAULT, D=8, latent=32, K=3, T=2)
    nn, ne, s, r = _batch(community_medium, [3, 50, 77, 12])
    n = int(nn.sum())
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((n, 8)) * 3 + 2).astype(np.float32)
    p = O.make_grevnet_params(21, 4, 32, 3, 2, final_scale=0.4)
    p["bn"] = O.make_bn_params(22, 4, 2)
    net = make_product_grevnet(hp, p)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    zg, _ = net(graph, inverse=True)
    bn = net.bns[0][0]
    mm0 = bn.moving_mean.clone()
    bn.update_moving_statistics()
    torch.testing.assert_close(bn.moving_mean, mm0 * 0.99 + bn.batch_mean * 0.01)
    for half in net.bns:
        for b in half:
            b.moving_mean.copy_(b.batch_mean)
            b.moving_variance.copy_(b.batch_variance)
    back = net(zg, inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(back, x, atol=2e-4, rtol=2e-4)
    # isolated statistics: a single-node-feature column that is constant has variance 0 -> epsilon keeps it finite
    xc = x.copy()
    xc[:, 0] = 1.0
    out = _run_forward(net, graph_from_arrays(nn, ne, s, r, xc, DEV))
    assert np.isfinite(float(out["log_prob_xs_per_node"]))


# ---- BASELINE configs 4 and 5 at full size: size-independent properties ------------------------------------
def test_config4_full_size_inverse_round_trip():
    """protein stand-in, batch 256 (~77k nodes, (2,2) workgroup shape): g then f returns z; f then g returns x;
    fused and layered kernels agree; per-shard sums add up to the batch sums."""
    from gnf_amd import datasets as D
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.graphs import data_dicts_to_graphs_tuple
    from gnf_amd.sharding import shard_graph_ids
    hp = dict(HP_DEFAULT)
    pool = D.synthetic_protein(256, seed=12345)
    rng = np.random.default_rng(7)
    feats = lambda n: rng.standard_normal((n, 64)).astype(np.float32)
    graph = data_dicts_to_graphs_tuple(pool.data_dicts(np.arange(256), feats), DEV)
    p = O.make_grevnet_params(99, 32, 256, 5, 8, final_scale=0.25)
    net = make_product_grevnet(hp, p)
    xg = net(graph, inverse=False)                       # sampling direction first (config 4)
    zb, _ = net(xg, inverse=True)
    assert float((zb.nodes - graph.nodes).abs().max()) <= 5e-5
    full = log_prob_terms(net, graph)
    lay = make_product_grevnet(hp, p)
    lay.fused = False
    flay = log_prob_terms(lay, graph)
    assert abs(float(full["log_prob_xs_per_node"]) - float(flay["log_prob_xs_per_node"])) <= 2e-5
    # additivity over 4 shards of whole graphs
    nn = graph.n_node.cpu().numpy()
    ne = graph.n_edge.cpu().numpy()
    total = torch.zeros(3, dtype=torch.float64, device=DEV)
    rng = np.random.default_rng(7)                       # same features again, graph by graph
    allx = graph.nodes.cpu().numpy()
    noff = np.concatenate([[0], np.cumsum(nn)])
    for shard in shard_graph_ids(nn, ne, 4):
        it = iter([allx[noff[i]:noff[i + 1]] for i in shard])
        sub = data_dicts_to_graphs_tuple(pool.data_dicts(shard, lambda n: next(it)), DEV)
        total += log_prob_terms(net, sub)["shard_sums"]
    assert abs(float(total[0] + total[1]) / float(total[2]) - float(full["log_prob_xs_per_node"])) <= 1e-5


def test_config5_full_size_properties():
    """citeseer/ego stand-in, batch 128, node-dim 256, 16-step flow: round trip, fused vs layered, re-run
    bitwise identical."""
    from gnf_amd import datasets as D
    from gnf_amd.graphs import data_dicts_to_graphs_tuple
    hp = dict(HP_DEFAULT, D=256, T=16)
    pool = D.synthetic_ego(128, seed=12345)
    rng = np.random.default_rng(8)
    graph = data_dicts_to_graphs_tuple(pool.data_dicts(np.arange(128), lambda n: rng.standard_normal((n, 256)).astype(np.float32)), DEV)
    p = O.make_grevnet_params(99, 128, 256, 5, 16, final_scale=0.25)
    net = make_product_grevnet(hp, p)
    zg, ld = net(graph, inverse=True)
    back = net(zg, inverse=False).nodes
    assert float((back - graph.nodes).abs().max()) <= 1e-4
    lay = make_product_grevnet(hp, p)
    lay.fused = False
    zl, ldl = lay(graph, inverse=True)
    n = graph.nodes.shape[0]
    assert abs(float(ld) - float(ldl)) / n <= 5e-6
    assert float((zl.nodes - zg.nodes).abs().max()) <= 1e-4
    zg2, ld2 = net(graph, inverse=True)
    assert torch.equal(zg2.nodes, zg.nodes) and float(ld2) == float(ld)


def test_rccl_backend_initialises_and_reduces_on_this_box():
    """One-rank RCCL communicator on the real GPU: the exact calls of bench.py's N > 1 path (init_process_group
    with the "nccl" backend bound to the device, the 3 x fp64 all-reduce, barrier, the MAX all-reduce of the elapsed
    time).  Multi-rank behaviour is covered by the gloo world_size-2 tests; this one proves the RCCL library loads
    and runs collectives in this environment."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        sums = torch.tensor([1.5, -2.25, 42.0], dtype=torch.float64, device=dev)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.barrier()
        t = torch.tensor([0.125], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        grad = torch.arange(1000, dtype=torch.float32, device=dev)
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)          # the flat gradient all-reduce of the training step
        torch.cuda.synchronize()
        assert sums.tolist() == [1.5, -2.25, 42.0] and float(t[0]) == 0.125 and float(grad[999]) == 999.0
    finally:
        dist.destroy_process_group()
