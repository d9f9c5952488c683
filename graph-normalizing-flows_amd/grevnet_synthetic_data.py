"""Synthetic 2-D datasets of the GRevNet driver (/root/reference/grevnet_synthetic_data.py:14-124): every
example is a fully connected directed graph WITH self loops whose node features are points of two moons
/ a mixture of Gaussians.  Same names, same `SyntheticDataset.get_next_batch_data_dicts(batch_size)` data
dicts (grevnet_synthetic_data.py:28-43); `get_next_batch` returns this package's GraphsTuple (on `device`).
networkx is not needed: the complete digraph's edge list is written directly, in networkx's edge order
(complete_graph(create_using=DiGraph) + add_edges_from(zip(r, r)): adjacency iteration, self loop last in
every row)."""
import functools
import random
from functools import partial

import numpy as np

from .graphs import data_dicts_to_graphs_tuple

MAX_SEED = 2 ** 32 - 1
GAUSSIAN_MEAN = [0, 0]
GAUSSIAN_COV = [[1, 0], [0, 1]]


@functools.lru_cache(maxsize=64)
def _fully_connected_edge_list(num_nodes):
    m = np.arange(num_nodes, dtype=np.int32)
    grid = np.tile(m, (num_nodes, 1))
    off_diag = grid[grid != m[:, None]].reshape(num_nodes, num_nodes - 1)
    receivers = np.concatenate([off_diag, m[:, None]], axis=1).ravel()      # the self loop was added last
    senders = np.repeat(m, num_nodes)
    senders.setflags(write=False)
    receivers.setflags(write=False)
    return senders, receivers


def fully_connected_edge_list(num_nodes):
    """Edges of fully_connected_nx_graph (grevnet_synthetic_data.py:17-21) in `g.edges()` order: for every node u
    its out-edges to v != u ascending, then the self loop (cached per size; read-only arrays)."""
    return _fully_connected_edge_list(int(num_nodes))


class SyntheticDataset:
    def __init__(self, graph_generator_fn):
        self.graph_generator_fn = graph_generator_fn

    def get_next_batch_data_dicts(self, batch_size):
        data_dicts = []
        for _ in range(batch_size):
            num_nodes, node_features = self.graph_generator_fn()
            s, r = fully_connected_edge_list(num_nodes)
            data_dicts.append({"n_node": num_nodes, "n_edge": len(s), "senders": s, "receivers": r,
                               "nodes": node_features, "globals": 0, "edges": np.zeros(len(s))})
        return data_dicts

    def get_next_batch(self, batch_size, device=None):
        return data_dicts_to_graphs_tuple(self.get_next_batch_data_dicts(batch_size), device)


@functools.lru_cache(maxsize=None)
def _moons_template(n_samples):
    """The noise-free, unshuffled point set of sklearn.datasets.make_moons(n_samples): outer half circle first,
    then the inner one (read-only [n, 2] float64)."""
    n_out = n_samples // 2
    n_in = n_samples - n_out
    t_out, t_in = np.linspace(0, np.pi, n_out), np.linspace(0, np.pi, n_in)
    x = np.vstack([np.append(np.cos(t_out), 1 - np.cos(t_in)), np.append(np.sin(t_out), 1 - np.sin(t_in) - 0.5)]).T
    x.setflags(write=False)
    return x


def make_moons(n_samples, noise, seed):
    """sklearn.datasets.make_moons(n_samples, shuffle=True, noise=noise, random_state=seed)[0], restated: the same
    point set, the same RandomState draws in the same order (one shuffle of arange(n), then one normal(scale=noise)
    of shape [n, 2]) - identical arrays (tests/test_host_logic_cpu.py checks against sklearn), without sklearn's
    parameter validation, which cost 1 ms per graph (32 graphs per batch: more host time than the GPU needs for the
    whole training iteration)."""
    rs = np.random.RandomState(seed)
    idx = np.arange(n_samples)
    rs.shuffle(idx)
    x = _moons_template(int(n_samples))[idx]
    if noise is not None:
        x += rs.normal(scale=noise, size=x.shape)
    return x


def moons_sample(n_samples, noise=0.05):
    # grevnet_synthetic_data.py:50-56: make_moons(..., random_state=random.randrange(MAX_SEED))[0].astype(np.float32)
    return n_samples, make_moons(n_samples, noise, random.randrange(MAX_SEED)).astype(np.float32)


def mom_sample(n_samples_choices, noise=0.05):
    return moons_sample(int(np.random.choice(n_samples_choices)), noise=noise)


def mog_sample(offsets_choices, rotate=False):
    offsets = random.choice(offsets_choices).copy()
    num_nodes = len(offsets)
    np.random.shuffle(offsets)
    features = np.random.multivariate_normal(GAUSSIAN_MEAN, GAUSSIAN_COV, num_nodes).astype(np.float32) + offsets
    if rotate:
        angle = np.random.random() * np.pi
        rot_mat = [[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]]
        features = np.transpose(np.matmul(rot_mat, np.transpose(features))).astype(np.float32)
    return num_nodes, features


OFFSETS_4 = np.array([[-5, 5], [5, 5], [-5, -5], [5, -5]]).astype(np.float32)
OFFSETS_6 = np.array([[-5, 5], [5, 5], [-5, -5], [5, -5], [15, 5], [15, -5]]).astype(np.float32)
OFFSETS_9 = np.array([[-5, 5], [5, 5], [-5, -5], [5, -5], [15, 5], [15, -5], [-5, 15], [5, 15],
                      [15, 15]]).astype(np.float32)

DATASETS_MAP = {
    "moons_100": SyntheticDataset(partial(moons_sample, n_samples=100)),
    "moons_10": SyntheticDataset(partial(moons_sample, n_samples=10)),
    "moons_6": SyntheticDataset(partial(moons_sample, n_samples=6)),
    "mom_6_10": SyntheticDataset(partial(mom_sample, n_samples_choices=[6, 10])),
    "mom_6_10_20": SyntheticDataset(partial(mom_sample, n_samples_choices=[6, 10, 20])),
    "mog_4": SyntheticDataset(partial(mog_sample, offsets_choices=[OFFSETS_4])),
    "mog_6": SyntheticDataset(partial(mog_sample, offsets_choices=[OFFSETS_6])),
    "mog_9": SyntheticDataset(partial(mog_sample, offsets_choices=[OFFSETS_9])),
    "mog_4_rotate": SyntheticDataset(partial(mog_sample, offsets_choices=[OFFSETS_4], rotate=True)),
    "mog_4_6": SyntheticDataset(partial(mog_sample, offsets_choices=[OFFSETS_4, OFFSETS_6])),
    "mog_4_9": SyntheticDataset(partial(mog_sample, offsets_choices=[OFFSETS_4, OFFSETS_9])),
}
