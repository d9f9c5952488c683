"""Host-side logic of the drop-in surface (CPU): GraphsTuple batching, CSR construction, datasets,
the reference-shaped class structure, parameter plumbing, sharding."""
from functools import partial

import numpy as np
import pytest
import torch

from gnf_amd import gnn
from gnf_amd.datasets import (EdgeListDataset, GraphDataset, fully_connected_edges, senders_receivers,
                              synthetic_ego, synthetic_protein, with_fully_connected_topology)
from gnf_amd.graphs import GraphsTuple, build_csr_host, data_dicts_to_graphs_tuple
from gnf_amd.sharding import shard_graph_ids
from oracle import gnf_oracle as O


def test_graphs_tuple_fields_and_replace():
    assert GraphsTuple._fields == ("nodes", "edges", "receivers", "senders", "globals", "n_node", "n_edge")
    g = data_dicts_to_graphs_tuple([
        {"nodes": np.ones((2, 4)), "senders": [0, 1, 0], "receivers": [0, 1, 1]},
        {"nodes": np.zeros((3, 4)), "senders": [0, 1, 2], "receivers": [2, 1, 0]}])
    assert g.nodes.shape == (5, 4) and g.nodes.dtype == torch.float32
    assert g.senders.tolist() == [0, 1, 0, 2, 3, 4] and g.receivers.tolist() == [0, 1, 1, 4, 3, 2]
    assert g.senders.dtype == torch.int32
    assert g.n_node.tolist() == [2, 3] and g.n_edge.tolist() == [3, 3]
    h = g.replace(nodes=g.nodes * 2)
    assert h.senders is g.senders and float(h.nodes.sum()) == 16.0 and float(g.nodes.sum()) == 8.0


def test_batching_matches_oracle_batcher(community_medium):
    n_node, n_edge, sl, rl = community_medium
    ds = EdgeListDataset(n_node, n_edge, sl, rl)
    ids = [5, 5, 100, 17]
    g = data_dicts_to_graphs_tuple(ds.data_dicts(ids, lambda n: np.zeros((n, 2), np.float32)))
    nn, ne, s, r = O.batch_graphs(n_node, n_edge, sl, rl, ids)
    assert g.senders.numpy().tolist() == s.tolist() and g.receivers.numpy().tolist() == r.tolist()
    assert g.n_node.numpy().tolist() == nn.tolist() and g.n_edge.numpy().tolist() == ne.tolist()


def test_build_csr_host_is_stable_receiver_sort():
    rng = np.random.default_rng(0)
    n, e = 13, 90
    s = rng.integers(0, n, e)
    r = rng.integers(0, n - 2, e)          # nodes n-2, n-1 receive nothing (empty rows)
    rowptr, col = build_csr_host(s, r, n)
    assert rowptr[0] == 0 and rowptr[-1] == e and rowptr.dtype == np.int32
    for v in range(n):
        want = [int(s[i]) for i in range(e) if r[i] == v]      # original edge order
        assert col[rowptr[v]:rowptr[v + 1]].tolist() == want
    assert rowptr[n - 1] == rowptr[n] == rowptr[n - 2]


def test_graph_dataset_semantics():
    ds = GraphDataset("graph_rnn_community_medium", 6, seed=1)
    assert len(ds.all) == 210 and len(ds.train_ids) == 168          # graph_data.py:77-78
    g = ds.get_next_train_batch(16)
    assert g.nodes.shape[1] == 6 and int(g.n_node.sum()) == g.nodes.shape[0]
    assert int(g.n_edge.sum()) == g.senders.shape[0]
    # one self loop per node, inserted first (graph_data.py:40-44)
    s, r = g.senders.numpy(), g.receivers.numpy()
    assert int((s == r).sum()) == g.nodes.shape[0]
    # every edge stays inside its graph (block diagonal)
    off = np.concatenate([[0], np.cumsum(g.n_node.numpy())])
    gid_of = np.repeat(np.arange(16), g.n_node.numpy())
    assert (gid_of[s] == gid_of[r]).all()
    # symmetric (to_directed)
    pairs = set(zip(s.tolist(), r.tolist()))
    assert all((b, a) in pairs for a, b in pairs)


def test_fully_connected_topology_order_and_count():
    a, b = fully_connected_edges(3)
    assert a.tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 2] and b.tolist() == [0, 1, 2, 0, 1, 2, 0, 1, 2]  # utils.py:138-143
    s, r, ne = senders_receivers([2, 3])
    assert ne.tolist() == [4, 9] and s.max() == 4 and len(s) == 13
    assert s[4:].min() == 2 and r[4:].min() == 2
    ds = with_fully_connected_topology(EdgeListDataset([2, 3], [1, 1], [0, 0], [0, 0]))
    assert ds.n_edge.tolist() == [4, 9]


def test_synthetic_stand_ins_shape():
    p = synthetic_protein(4, seed=3)
    assert (p.n_node >= 100).all() and (p.n_node <= 500).all()
    deg = p.n_edge.sum() / p.n_node.sum()
    assert 3.5 < deg < 8
    e = synthetic_ego(4, seed=3)
    assert (e.n_node >= 50).all() and (e.n_node < 400).all()
    for ds in (p, e):
        n, s, r = ds.graph(0)
        assert (s[:n] == np.arange(n)).all() and (r[:n] == np.arange(n)).all()   # self loops first
        assert s.max() < n and r.max() < n


def test_reference_shaped_structure():
    mk_mlp = partial(gnn.make_mlp_model, 32, 8 / 2, 5, gnn.leaky_relu, 0.01, 0.1)   # run_grevnet.py:175-180
    mk = partial(gnn.avg_then_mlp_gnn, mk_mlp, 1.0)
    net = gnn.GRevNet(mk, 3, 8, use_batch_norm=False, weight_sharing=False)
    assert len(net.s) == 2 and len(net.s[0]) == 3 and len(net.t[1]) == 3            # gnn.py:288-296
    assert isinstance(net.s[0][0], gnn.NodeBlockGNN)
    blk = net.s[0][0]._node_block
    assert isinstance(blk, gnn.AggThenMLPBlock) and blk.epsilon == 1.0
    assert blk._mlp.layer_sizes == [32, 32, 32, 32, 4] and blk._mlp.alpha == pytest.approx(0.2)
    shared = gnn.GRevNet(mk, 3, 8, weight_sharing=True)
    assert len(shared.s) == 2 and isinstance(shared.s[0], gnn.NodeBlockGNN)          # gnn.py:284-286
    bn_net = gnn.GRevNet(mk, 3, 8, use_batch_norm=True)                            # gnn.py:298-299
    assert bn_net.use_batch_norm and len(bn_net.bns) == 2 and len(bn_net.bns[1]) == 3
    assert isinstance(bn_net.bns[0][0], gnn.BatchNormBijector) and bn_net.bns[0][0].epsilon == 1e-3
    assert len(net.bns[0]) == 3 and not net.use_batch_norm                          # created even when unused
    c = gnn.sum_concat_then_mlp_gnn(mk_mlp)._node_block
    assert isinstance(c, gnn.ConcatThenMLPBlock) and c.in_dim(4) == 8
    assert gnn.EDGE_BLOCK_OPT == {"use_edges": False, "use_receiver_nodes": False, "use_sender_nodes": True,
                                  "use_globals": False}
    with pytest.raises(ValueError):
        gnn.AggThenMLPBlock("max", mk_mlp, 1.0)


def test_lazy_init_and_param_roundtrip():
    gnn.set_random_seed(7)
    m = gnn.make_mlp_model(16, 4, 3, gnn.relu, 0.01, 0.1)
    assert m.params is None
    m.ensure_built(6, "cpu")
    assert [tuple(w.shape) for w, _ in m.params] == [(6, 16), (16, 16), (16, 4)]
    w0 = m.params[0][0]
    std = np.sqrt(2.0 / (6 + 16)) / 0.87962566103423978
    assert float(w0.abs().max()) <= 2 * std + 1e-6                                   # truncated at 2 sigma
    assert float(m.params[0][1].abs().max()) <= 0.2 + 1e-6
    with pytest.raises(ValueError):
        m.ensure_built(7, "cpu")
    p = O.make_grevnet_params(1, 4, 16, 3, 2)
    mk = partial(gnn.avg_then_mlp_gnn, partial(gnn.make_mlp_model, 16, 4, 3, gnn.leaky_relu), 1.0)
    net = gnn.GRevNet(mk, 2, 8).set_params(p)
    q = net.get_params()
    for kind in ("s", "t"):
        for half in range(2):
            for i in range(2):
                for (w, b), (w2, b2) in zip(p[kind][half][i], q[kind][half][i]):
                    assert np.array_equal(w, w2) and np.array_equal(b, b2)
    with pytest.raises(ValueError):
        net.mlps("s")[0].set_params([(np.zeros((4, 16)), np.zeros(16))])            # wrong layer count


def test_sharding_is_balanced_and_complete(community_medium):
    n_node, n_edge, _, _ = community_medium
    rng = np.random.default_rng(12345)
    ids = rng.choice(168, size=512, replace=True)
    nn, ne = n_node[ids], n_edge[ids]
    shards = shard_graph_ids(nn, ne, 8)
    allpos = np.sort(np.concatenate(shards))
    assert allpos.tolist() == list(range(512))
    loads = np.array([nn[s].sum() + 0.05 * ne[s].sum() for s in shards])
    assert loads.max() / loads.mean() < 1.01
    assert shard_graph_ids(nn, ne, 8)[3].tolist() == shards[3].tolist()             # deterministic
    one = shard_graph_ids(nn[:3], ne[:3], 8)                                         # fewer graphs than ranks
    assert sum(len(s) for s in one) == 3


# ---- embedding-chunk format of the data-backed trainer (train_grevnet_with_data.py:145-271) ----------
def _write_chunks(tmp_path, sizes_per_file, d=6, seed=0):
    from gnf_amd import datasets as D
    rng = np.random.default_rng(seed)
    allz, alln = [], []
    for k, sizes in enumerate(sizes_per_file):
        z = rng.standard_normal((int(np.sum(sizes)), d))
        D.write_embedding_chunk(str(tmp_path / f"chunk_{k}.pkl"), z, sizes)
        allz.append(z)
        alln.append(np.asarray(sizes, np.int32))
    return allz, alln


def test_fixed_chunk_reader_batches_and_file_rollover(tmp_path):
    import os
    from gnf_amd import datasets as D
    _write_chunks(tmp_path, [[3, 4, 5, 2, 6], [2, 2, 7, 1]])
    ds = D.GrevnetDatasetFixed(str(tmp_path), 2)
    order = os.listdir(str(tmp_path))
    import pickle
    first = pickle.load(open(tmp_path / order[0], "rb"))
    z, n = ds.train_batch()
    np.testing.assert_array_equal(n, first[1][:2])
    np.testing.assert_array_equal(z, first[0][:int(first[1][:2].sum())])
    z, n = ds.train_batch()
    np.testing.assert_array_equal(n, first[1][2:4])
    if len(first[1]) == 5:      # the odd graph at the end of a chunk is dropped, next file opened
        second = pickle.load(open(tmp_path / order[1], "rb"))
        z, n = ds.train_batch()
        np.testing.assert_array_equal(n, second[1][:2])
        np.testing.assert_array_equal(z, second[0][:int(second[1][:2].sum())])


def test_variable_chunk_reader_respects_max_nodes(tmp_path):
    from gnf_amd import datasets as D
    allz, alln = _write_chunks(tmp_path, [[3, 4, 5, 2, 6, 1, 1]])
    ds = D.GrevnetDatasetVariable(str(tmp_path), max_nodes=10)
    z, n = ds.train_batch()
    np.testing.assert_array_equal(n, [3, 4])            # 3 + 4 + 5 >= 10 stops before the third graph
    assert z.shape == (7, 6)
    z, n = ds.train_batch()
    np.testing.assert_array_equal(n, [5, 2])            # 5 + 2 + 6 >= 10
    np.testing.assert_array_equal(z, allz[0][7:14])


def test_transform_example_is_fully_connected_with_self_loops():
    from gnf_amd import datasets as D
    z = np.arange(5 * 4, dtype=np.float32).reshape(5, 4)
    g = D.transform_example(z, [2, 3])
    assert g.n_edge.tolist() == [4, 9]
    pairs = set(zip(g.senders.tolist(), g.receivers.tolist()))
    assert pairs == {(a, b) for a in (0, 1) for b in (0, 1)} | {(a, b) for a in (2, 3, 4) for b in (2, 3, 4)}
    assert g.nodes.shape == (5, 4) and g.edges.shape[0] == 13


def test_synthetic_datasets_mirror_the_reference_generators():
    """grevnet_synthetic_data.py: complete digraph with self loops (n^2 edges, the self loop last in every
    row as networkx lists it), moons / mixture-of-Gaussians features, every dataset name of DATASETS_MAP."""
    import random
    from gnf_amd import grevnet_synthetic_data as S
    assert set(S.DATASETS_MAP) == {"moons_100", "moons_10", "moons_6", "mom_6_10", "mom_6_10_20", "mog_4", "mog_6",
                                   "mog_9", "mog_4_rotate", "mog_4_6", "mog_4_9"}
    s, r = S.fully_connected_edge_list(3)
    assert list(zip(s.tolist(), r.tolist())) == [(0, 1), (0, 2), (0, 0), (1, 0), (1, 2), (1, 1), (2, 0), (2, 1), (2, 2)]
    random.seed(1)
    np.random.seed(1)
    dd = S.DATASETS_MAP["mom_6_10"].get_next_batch_data_dicts(5)
    assert all(d["n_node"] in (6, 10) and d["n_edge"] == d["n_node"] ** 2 and d["nodes"].shape == (d["n_node"], 2)
               for d in dd)
    g = S.DATASETS_MAP["mog_4"].get_next_batch(3)
    assert g.nodes.shape == (12, 2) and g.n_edge.tolist() == [16, 16, 16]
    assert int(g.senders.max()) == 11 and int(g.receivers.min()) == 0      # global node ids after batching
    # mixture of Gaussians: one point near each of the four offsets
    pts = g.nodes[:4].numpy()
    assert sorted(np.sign(pts).astype(int).tolist()) == sorted([[-1, 1], [1, 1], [-1, -1], [1, -1]])


def test_overfit_graph_dataset_subset_rules():
    """graph_data.py:125-177: smallest graphs of the train split (or the first graph of each requested size),
    cycled up to max(num_graphs, train_batch_size)."""
    from gnf_amd import datasets as D
    full = D.GraphDataset("graph_rnn_community_medium", 4)
    sizes = sorted(full.train_n_nodes())
    ds = D.OverfitGraphDataset("graph_rnn_community_medium", 3, 8, 4)
    assert len(ds.train_ids) == 8 and sorted(set(ds.train_n_nodes())) == sorted(set(sizes[:3]))
    assert ds.full_n_nodes() == ds.train_n_nodes() == ds.test_n_nodes()
    g = ds.get_next_train_batch(5)
    assert set(g.n_node.tolist()) <= set(sizes[:3]) and g.nodes.shape[1] == 4
    want = [sizes[0], sizes[-1]]
    ds2 = D.OverfitGraphDataset("graph_rnn_community_medium", 2, 2, 4, graph_sizes=want)
    assert ds2.train_n_nodes() == want


def test_make_moons_is_sklearns_stream():
    """grevnet_synthetic_data.py:50-56 draws its features with sklearn.datasets.make_moons(..., random_state=seed);
    the mirror restates that generator (same point set, same RandomState draws in the same order) to shed sklearn's
    per-call parameter validation - the arrays must be IDENTICAL for every size / seed / noise."""
    datasets = pytest.importorskip("sklearn.datasets")
    from gnf_amd import grevnet_synthetic_data as G
    for n in (4, 6, 7, 33, 100):
        for seed in (0, 5, 123456, 2 ** 31 - 1):
            for noise in (0.05, None, 0.3):
                a = G.make_moons(n, noise, seed)
                b = datasets.make_moons(n_samples=n, shuffle=True, noise=noise, random_state=seed)[0]
                assert np.array_equal(a, b), (n, seed, noise)


def test_check_graphs_tuple_rejects_cross_graph_edges_and_bad_counts():
    """gnf_build_csr trusts the block-diagonal layout; check_graphs_tuple is the host-side validation for
    hand-built batches."""
    import torch
    from gnf_amd.graphs import GraphsTuple, check_graphs_tuple, data_dicts_to_graphs_tuple
    good = data_dicts_to_graphs_tuple([{"nodes": np.zeros((3, 2)), "senders": [0, 1, 2], "receivers": [1, 2, 0]},
                                       {"nodes": np.zeros((2, 2)), "senders": [0, 1], "receivers": [1, 0]}])
    assert check_graphs_tuple(good)
    cross = good.replace(senders=torch.tensor([0, 1, 2, 3, 2], dtype=torch.int32))      # last edge: node 2 is in graph 0
    with pytest.raises(ValueError, match="leaves its graph"):
        check_graphs_tuple(cross)
    short = good.replace(n_edge=torch.tensor([3, 1], dtype=torch.int32))
    with pytest.raises(ValueError, match="sum\\(n_edge\\)"):
        check_graphs_tuple(short)


def test_chunk_readers_open_their_first_file_in_the_constructor(tmp_path):
    """file_ind exists before the first batch, an empty directory and an oversized graph fail at construction, and the
    short last batch of the last file is handed out (documented difference to train_grevnet_with_data.py:183-234)."""
    from gnf_amd import datasets as D
    with pytest.raises(IndexError):
        D.GrevnetDatasetVariable(str(tmp_path), 50)
    rng = np.random.default_rng(0)
    for k, sizes in enumerate(([10, 20, 5], [30, 4])):
        D.write_embedding_chunk(str(tmp_path / f"chunk{k}.pkl"), rng.standard_normal((sum(sizes), 3)), sizes)
    ds = D.GrevnetDatasetVariable(str(tmp_path), 32, sort_files=True)
    assert ds.file_ind == 0
    got = []
    while True:
        try:
            got.append(ds.train_batch()[1].tolist())
        except IndexError:
            break
    assert got == [[10, 20], [5], [30], [4]] and ds.file_ind == 1
    with pytest.raises(ValueError):
        D.GrevnetDatasetVariable(str(tmp_path), 15, sort_files=True)          # the 20-node graph of the FIRST chunk
    late = D.GrevnetDatasetVariable(str(tmp_path), 25, sort_files=True)       # the 30-node graph sits in the second chunk:
    assert late.train_batch()[1].tolist() == [10] and late.train_batch()[1].tolist() == [20] and late.train_batch()[1].tolist() == [5]
    with pytest.raises(ValueError):                                          # found when that chunk is opened
        late.train_batch()
    fx = D.GrevnetDatasetFixed(str(tmp_path), 2, sort_files=True)
    assert fx.file_ind == 0 and fx.n_node.tolist() == [10, 20, 5]             # the open chunk's sizes (reference: self.n_node = d[1])
    assert fx.train_batch()[1].tolist() == [10, 20] and fx.train_batch()[1].tolist() == [30, 4]
    assert fx.n_node.tolist() == [30, 4]


def test_example_driver_defaults_are_the_reference_flags():
    """examples/train_grevnet_with_data.py's argument defaults = train_grevnet_with_data.py:40-46, 86, 100-117."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_grevnet_with_data.py")).read()
    want = {"attn_type": '"dm_attn"', "attn_kq_dim": "64", "attn_v_dim": "64", "attn_num_heads": "1",
            "attn_concat_heads_output_dim": "64", "node_embedding_dim": "200", "latent_dim": "2048", "num_layers": "3",
            "num_coupling_layers": "10", "bias_init_stddev": "0.3", "adam_beta2": "0.999", "train_batch_size": "32"}
    for flag, val in want.items():
        m = re.search(r'add_argument\("--%s",[^\n]*default=([^,)\s]+)' % flag, src)
        assert m and m.group(1) == val, (flag, m and m.group(1))


def test_mlp_stash_budget_rule():
    """train.mlp_stash_within_budget: the MLP-row stash is taken only within the caller's cap or, with no
    cap, within a fraction of the device memory that is free - a 20 GB stash of a 30 k-node wide-net batch must not turn a
    step that trains through the recomputing walk into a device OOM."""
    from gnf_amd.train import mlp_stash_within_budget
    gb = 1 << 30
    assert mlp_stash_within_budget(2 * gb, 200 * gb)[0]
    ok, why = mlp_stash_within_budget(20 * gb, 30 * gb)
    assert not ok and "free" in why
    assert mlp_stash_within_budget(20 * gb, 30 * gb, free_fraction=0.9)[0]
    assert mlp_stash_within_budget(20 * gb, 1 * gb, max_bytes=32 * gb)[0]       # an explicit cap is the caller's decision
    ok, why = mlp_stash_within_budget(20 * gb, 200 * gb, max_bytes=8 * gb)
    assert not ok and "mlp_stash_max_bytes" in why
