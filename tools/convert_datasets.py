#!/usr/bin/env python3
"""Convert the reference's GraphRNN pickles into neutral .npz edge lists.

Runs ONLY in the build container (needs /root/reference + networkx).  The GPU box never sees
/root/reference, pickle or networkx: it reads the committed data/*.npz files.

What is replicated (data preparation, not code): the preprocessing of
/root/reference/graph_data.py:33-50 (`convert_nx_repr`) applied after `.to_directed()`
(graph_data.py:80-85): nodes relabelled 0..n-1 in node-iteration order, ONE self loop per node
inserted first, then every directed edge.  Edge order is the `DiGraph.edges()` iteration order,
which is what graph_nets' `networkxs_to_graphs_tuple` (graph_data.py:122) consumes.

Output arrays per dataset (all graphs of the pickle, in file order; the reference's train split is
the first int(0.8*G) graphs, graph_data.py:77-78):
  n_node[G] int32, n_edge[G] int32, senders[sum n_edge] int32, receivers[sum n_edge] int32
with LOCAL (per-graph, 0-based) node ids.
"""
import os
import pickle
import sys

import networkx as nx
import numpy as np

REF = "/root/reference/training_graphs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")

DATASETS = {
    "grid_small": "GraphRNN_RNN_grid_small_4_64_train_0.dat",
    "community_medium": "GraphRNN_RNN_community_medium_4_128_train_0.dat",
    "caveman_small": "GraphRNN_RNN_caveman_small_4_64_train_0.dat",
    "citeseer_small": "GraphRNN_RNN_citeseer_small_4_64_train_0.dat",
    "grid": "GraphRNN_RNN_grid_4_128_train_0.dat",
}


def preprocess(g):
    """to_directed + relabel + self-loop-first, as graph_data.py:33-50, 80-85 prepare the data."""
    g = g.to_directed()
    new = nx.DiGraph()
    index = {}
    for i, node in enumerate(g.nodes()):
        index[node] = i
        new.add_node(i)
        new.add_edge(i, i)
    for u, v in g.edges():
        new.add_edge(index[u], index[v])
    e = list(new.edges())
    s = np.array([a for a, _ in e], dtype=np.int32)
    r = np.array([b for _, b in e], dtype=np.int32)
    return new.number_of_nodes(), s, r


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in DATASETS.items():
        with open(os.path.join(REF, fn), "rb") as f:
            graphs = pickle.load(f)
        n_node, n_edge, S, R = [], [], [], []
        for g in graphs:
            n, s, r = preprocess(g)
            n_node.append(n)
            n_edge.append(len(s))
            S.append(s)
            R.append(r)
        out = os.path.join(OUT, name + ".npz")
        np.savez_compressed(out,
                            n_node=np.array(n_node, np.int32),
                            n_edge=np.array(n_edge, np.int32),
                            senders=np.concatenate(S),
                            receivers=np.concatenate(R))
        nn, ne = np.array(n_node), np.array(n_edge)
        print(f"{name}: G={len(graphs)} n_node min/mean/max={nn.min()}/{nn.mean():.2f}/{nn.max()} "
              f"n_edge min/mean/max={ne.min()}/{ne.mean():.1f}/{ne.max()} total_e={ne.sum()} -> {out}")


if __name__ == "__main__":
    sys.exit(main())
