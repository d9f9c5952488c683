"""Harness-side data for the hot path (no arithmetic of the path itself):

  * GraphDataset       - the reference's GraphRNN-pickle datasets after graph_data.py:33-50 preprocessing
                         (to_directed + one self loop per node first), read from the committed
                         data/*.npz edge lists (tools/convert_datasets.py), with the reference's batch
                         semantics (graph_data.py:61-122): train split = first int(0.8*G) graphs,
                         graphs drawn uniformly WITH replacement (the `self.index` quirk at
                         graph_data.py:119 leaves train_index at 0), fresh N(0, scale^2) node features
                         per batch (graph_data.py:24-27).
  * fully_connected_edges / senders_receivers - the complete-graph + self-loop topology of
                         grevnet_synthetic_data.py:17-21 and utils.py:164-183 (sender-major order).
  * synthetic stand-ins - protein / citeseer(ego) datasets are absent from the reference
                         (.MISSING_LARGE_BLOBS); BASELINE.md section 4 fixes the generators used instead.
"""
import os

import numpy as np

from .graphs import data_dicts_to_graphs_tuple

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")

# reference name (graph_data.py:283-300 FILENAME_MAP key) -> committed edge-list file
FILENAME_MAP = {
    "graph_rnn_grid": "grid.npz",
    "graph_rnn_ego_small": "citeseer_small.npz",
    "graph_rnn_community_small": "caveman_small.npz",
    "graph_rnn_community_medium": "community_medium.npz",
    "graph_rnn_grid_small": "grid_small.npz",   # no key in the reference map; BASELINE config 1
}


class EdgeListDataset:
    """A list of graphs as concatenated local edge lists."""

    def __init__(self, n_node, n_edge, senders, receivers):
        self.n_node = np.asarray(n_node, np.int32)
        self.n_edge = np.asarray(n_edge, np.int32)
        self.senders = np.asarray(senders, np.int32)
        self.receivers = np.asarray(receivers, np.int32)
        self.eoff = np.concatenate([[0], np.cumsum(self.n_edge)]).astype(np.int64)

    def __len__(self):
        return len(self.n_node)

    def graph(self, gid):
        lo, hi = self.eoff[gid], self.eoff[gid + 1]
        return int(self.n_node[gid]), self.senders[lo:hi], self.receivers[lo:hi]

    def data_dicts(self, graph_ids, nodes_fn):
        out = []
        for gid in graph_ids:
            n, s, r = self.graph(gid)
            out.append({"nodes": nodes_fn(n), "senders": s, "receivers": r, "n_node": n})
        return out

    @staticmethod
    def load(path):
        d = np.load(path)
        return EdgeListDataset(d["n_node"], d["n_edge"], d["senders"], d["receivers"])


class GraphDataset:
    def __init__(self, dataset_name, node_embedding_dim, gaussian_scale=1.0, seed=12345):
        self.all = EdgeListDataset.load(os.path.join(DATA_DIR, FILENAME_MAP[dataset_name]))
        g = len(self.all)
        self.train_ids = np.arange(0, int(0.8 * g))        # graph_data.py:77-78
        self.test_ids = np.arange(int(0.8 * g), g)
        self.dim = int(node_embedding_dim)
        self.scale = float(gaussian_scale)
        self.rng = np.random.default_rng(seed)              # run_grevnet.py:108 default seed

    def _features(self, n):
        return self.rng.normal(scale=self.scale, size=(n, self.dim)).astype(np.float32)

    def sample_ids(self, batch_size, split="train"):
        ids = self.train_ids if split == "train" else self.test_ids
        return self.rng.choice(ids, size=batch_size, replace=True)

    def get_next_train_batch(self, batch_size, device=None):
        return data_dicts_to_graphs_tuple(self.all.data_dicts(self.sample_ids(batch_size), self._features), device)

    def get_next_test_batch(self, batch_size, device=None):
        return data_dicts_to_graphs_tuple(self.all.data_dicts(self.sample_ids(batch_size, "test"), self._features),
                                          device)

    def get_random_test_batch(self, batch_size, device=None):            # graph_data.py:109-111
        return self.get_next_test_batch(batch_size, device)

    def full_n_nodes(self):                                              # graph_data.py:89-96
        return [int(n) for n in self.all.n_node]

    def train_n_nodes(self):
        return [int(self.all.n_node[i]) for i in self.train_ids]

    def test_n_nodes(self):
        return [int(self.all.n_node[i]) for i in self.test_ids]


class OverfitGraphDataset(GraphDataset):
    """graph_data.py:125-203: train on a handful of graphs.  The train split is sorted by node count; either
    the `num_graphs` smallest graphs, or the first graph of every size listed in `graph_sizes`, repeated
    cyclically up to max(num_graphs, train_batch_size) entries (subset_graphs, graph_data.py:157-177); batches are
    then drawn from that list exactly like GraphDataset's (uniformly with replacement, graph_data.py:194-203).
    full / train / test_n_nodes all describe the subset, as in the reference (graph_data.py:147-154)."""

    def __init__(self, dataset_name, num_graphs, train_batch_size, node_embedding_dim, graph_sizes=None,
                 gaussian_scale=1.0, seed=12345):
        super().__init__(dataset_name, node_embedding_dim, gaussian_scale, seed)
        order = sorted(self.train_ids.tolist(), key=lambda i: int(self.all.n_node[i]))   # stable, like list.sort
        if graph_sizes:
            first = {}
            for i in order:
                first.setdefault(int(self.all.n_node[i]), i)
            subset = [first[int(sz)] for sz in graph_sizes]                              # KeyError like the reference
        else:
            subset = order[:int(num_graphs)]
        want = max(int(num_graphs), int(train_batch_size))
        self.train_ids = np.array([subset[k % len(subset)] for k in range(want)], dtype=np.int64)

    def full_n_nodes(self):
        return self.train_n_nodes()

    def test_n_nodes(self):
        return self.train_n_nodes()


def fully_connected_edges(n):
    """All ordered pairs (a, b) incl. self, sender-major (utils.py:138-143 order; same edge SET as
    grevnet_synthetic_data.py:17-21)."""
    a = np.repeat(np.arange(n, dtype=np.int32), n)
    b = np.tile(np.arange(n, dtype=np.int32), n)
    return a, b


def senders_receivers(n_node):
    """utils.py:164-183 for a whole batch: returns (senders, receivers, n_edge = n_node**2) with global ids."""
    s, r, off = [], [], 0
    for n in n_node:
        a, b = fully_connected_edges(int(n))
        s.append(a + off)
        r.append(b + off)
        off += int(n)
    return np.concatenate(s), np.concatenate(r), np.asarray(n_node, np.int32) ** 2


def with_fully_connected_topology(ds):
    """Same node counts, complete-graph topology (train_grevnet_with_data.py:237-244 transform_example)."""
    S, R = [], []
    for n in ds.n_node:
        a, b = fully_connected_edges(int(n))
        S.append(a)
        R.append(b)
    return EdgeListDataset(ds.n_node, ds.n_node.astype(np.int64) ** 2, np.concatenate(S), np.concatenate(R))


def _knn_graph(rng, n, k):
    """Random geometric k-NN graph in the unit square, symmetrised, + self loops first
    (the data shape of graph_data.py:33-50)."""
    pts = rng.random((n, 2))
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nbr = np.argsort(d2, axis=1)[:, :k]
    adj = np.zeros((n, n), bool)
    adj[np.repeat(np.arange(n), k), nbr.ravel()] = True
    adj |= adj.T
    s, r = np.nonzero(adj)
    idx = np.arange(n)
    return np.concatenate([idx, s]).astype(np.int32), np.concatenate([idx, r]).astype(np.int32)


def synthetic_protein(num_graphs, seed=12345):
    """Stand-in for the absent protein dataset: n ~ U{100..500}, geometric k-NN
    graph with mean degree about 5, + self loops."""
    rng = np.random.default_rng(seed)
    nn, ne, S, R = [], [], [], []
    for _ in range(num_graphs):
        n = int(rng.integers(100, 501))
        s, r = _knn_graph(rng, n, 3)
        nn.append(n); ne.append(len(s)); S.append(s); R.append(r)
    return EdgeListDataset(nn, ne, np.concatenate(S), np.concatenate(R))


def synthetic_ego(num_graphs, seed=12345):
    """Stand-in for the absent citeseer (ego) dataset: n ~ U{50..399}; node 0 is
    the ego hub (linked to a third of the nodes), the rest grows by preferential attachment with 2
    links per new node (mean degree about 5-6), + self loops first."""
    rng = np.random.default_rng(seed)
    nn, ne, S, R = [], [], [], []
    for _ in range(num_graphs):
        n = int(rng.integers(50, 400))
        deg = np.zeros(n)
        edges = set()
        for v in range(1, n):
            if v % 3 == 1:
                edges.add((0, v)); deg[0] += 1; deg[v] += 1
            w = deg[:v] + 1.0
            for u in rng.choice(v, size=min(2, v), replace=False, p=w / w.sum()):
                e = (int(u), v)
                if e not in edges:
                    edges.add(e); deg[u] += 1; deg[v] += 1
        und = np.array(sorted(edges), dtype=np.int32).reshape(-1, 2)
        idx = np.arange(n, dtype=np.int32)
        s = np.concatenate([idx, und[:, 0], und[:, 1]])
        r = np.concatenate([idx, und[:, 1], und[:, 0]])
        nn.append(n); ne.append(len(s)); S.append(s); R.append(r)
    return EdgeListDataset(nn, ne, np.concatenate(S), np.concatenate(R))


# ----------------------------------------------------------------------------------------------
# On-disk embedding chunks of the data-backed trainer
#   writer: generate_grevnet_training_data.py:89-97,116-120   reader: train_grevnet_with_data.py:145-234
# A chunk is pickle.dump((node_embeddings[sum(n_node), D] float, n_node[B] int32)); a directory of chunks
# is consumed in os.listdir order.
# ----------------------------------------------------------------------------------------------
def write_embedding_chunk(path, node_embeddings, n_node):
    """generate_grevnet_training_data.py:91-92."""
    import pickle
    node_embeddings = np.asarray(node_embeddings)
    n_node = np.asarray(n_node, np.int32)
    if node_embeddings.ndim != 2 or int(n_node.sum()) != node_embeddings.shape[0]:
        raise ValueError("node_embeddings must be [sum(n_node), D]")
    with open(path, "wb") as f:
        pickle.dump((node_embeddings, n_node), f)


def _read_embedding_chunk(path):
    import pickle
    with open(path, "rb") as f:
        node_embeddings, n_node = pickle.load(f)[:2]
    return node_embeddings, np.asarray(n_node)


def plan_fixed_batches(n_node, batch_size):
    """Graph ranges [lo, hi) of one chunk for a fixed number of graphs per batch: whole batches only - the graphs
    left over at the end of a chunk are dropped (train_grevnet_with_data.py:163-176 moves on to the next file as
    soon as the current one cannot supply `train_batch_size` more graphs)."""
    whole = len(n_node) // batch_size
    return [(k * batch_size, (k + 1) * batch_size) for k in range(whole)]


def plan_variable_batches(n_node, max_nodes):
    """Graph ranges [lo, hi) of one chunk for batches of consecutive graphs holding FEWER than max_nodes nodes
    (strict, train_grevnet_with_data.py:221): a batch closes in front of the first graph that would reach the
    limit; the range that runs into the end of the chunk is emitted short (:201-219)."""
    out, lo, total = [], 0, 0
    for g, n in enumerate(n_node):
        if total + int(n) >= max_nodes:
            out.append((lo, g))
            lo, total = g, 0
            # the graph that did not fit opens the next batch - unless it cannot fit in ANY batch
            if int(n) >= max_nodes:
                raise ValueError(f"graph {g} has {int(n)} nodes: never fits under max_nodes={max_nodes} "
                                 "(the reference would hand out empty batches forever)")
        total += int(n)
    out.append((lo, len(n_node)))
    return out


class _ChunkBatches:
    """Batches out of a directory of embedding chunks: the files are visited once in order, each chunk is cut by
    `plan` into graph ranges, a batch is (node_embeddings[rows of the range], n_node[range]).  Running past the
    last file raises IndexError, like the reference's `self.files[self.file_ind]`.
    The first chunk is opened (and cut) by the constructor, as the reference's readers do in theirs
    (train_grevnet_with_data.py:150-157, 188-195): an empty or unreadable directory, or a graph that can never fit under
    max_nodes, fails there and not in the middle of an epoch; `file_ind` exists from the start.
    One deliberate difference: the short LAST batch of the LAST file is handed out (the variable-size reader of the
    reference loses it - it raises IndexError while opening the file after the last one, before returning the batch)."""

    def __init__(self, directory, files, plan):
        import os
        self._paths = [os.path.join(directory, f) for f in files]
        self.files = list(files)
        self._plan = plan
        self.file_ind = 0
        if not self._paths:
            raise IndexError(f"{type(self).__name__}: no training files in {directory!r}")
        self._first = self._cut(self._paths[0])
        self._batches = self._generate()

    def _cut(self, path):
        emb, n_node = _read_embedding_chunk(path)
        row_end = np.concatenate([[0], np.cumsum(n_node)])
        self.n_node = n_node      # the chunk just opened, like the reference readers' self.n_node = d[1] (:157, 172)
        return emb, n_node, row_end, self._plan(n_node)

    def _generate(self):
        for ind, path in enumerate(self._paths):
            self.file_ind = ind
            emb, n_node, row_end, ranges = self._first if ind == 0 else self._cut(path)
            self._first = None
            for lo, hi in ranges:
                yield emb[row_end[lo]:row_end[hi]], n_node[lo:hi]

    def train_batch(self):
        try:
            return next(self._batches)
        except StopIteration:
            raise IndexError(f"{type(self).__name__}: out of training files (the reference raises IndexError too)") from None


def _chunk_files(directory, sort_files):
    import os
    # os.listdir order, like the reference: unspecified, it changes from one directory to the next.
    # sort_files=True (not in the reference) makes a run reproducible.
    return sorted(os.listdir(directory)) if sort_files else os.listdir(directory)


class GrevnetDatasetFixed(_ChunkBatches):
    """train_grevnet_with_data.py:145-180: `train_batch_size` graphs per batch, chunk tails dropped;
    `train_epochs` is FLAGS.train_epochs (the file list is repeated that many times)."""

    def __init__(self, train_data_dir, train_batch_size, train_epochs=1, sort_files=False):
        self.train_batch_size = int(train_batch_size)
        super().__init__(train_data_dir, _chunk_files(train_data_dir, sort_files) * int(train_epochs),
                         lambda n_node: plan_fixed_batches(n_node, self.train_batch_size))


class GrevnetDatasetVariable(_ChunkBatches):
    """train_grevnet_with_data.py:183-234: consecutive graphs up to (not reaching) max_nodes per batch."""

    def __init__(self, train_data_dir, max_nodes, sort_files=False):
        self.max_nodes = int(max_nodes)
        super().__init__(train_data_dir, _chunk_files(train_data_dir, sort_files),
                         lambda n_node: plan_variable_batches(n_node, self.max_nodes))


def transform_example(node_embeddings, n_node, device=None):
    """train_grevnet_with_data.py:237-271: a (node_embeddings, n_node) batch -> GraphsTuple with the complete
    topology incl. self loops (utils.py:164-183), n_edge = n_node**2, zero edges / globals."""
    import torch
    from .graphs import GraphsTuple
    n_node = np.asarray(n_node, np.int32)
    s, r, n_edge = senders_receivers(n_node)
    g = GraphsTuple(nodes=torch.as_tensor(np.asarray(node_embeddings, np.float32)),
                    edges=torch.zeros(len(s), dtype=torch.float32),
                    receivers=torch.as_tensor(r.astype(np.int32)), senders=torch.as_tensor(s.astype(np.int32)),
                    globals=torch.zeros(len(n_node), dtype=torch.float32),
                    n_node=torch.as_tensor(n_node), n_edge=torch.as_tensor(n_edge.astype(np.int32)))
    return g.to(device) if device is not None else g
