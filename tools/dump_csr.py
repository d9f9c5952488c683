#!/usr/bin/env python3
"""Developer tool: write the receiver-sorted CSR of a bench batch to a flat binary file for the stand-alone kernel
probes under tools/probes/ (numpy only, no GPU):  int64 N, int64 E, int32 rowptr[N+1], int32 col[E].
    python tools/dump_csr.py ego128 /tmp/ego128.csr        (the config-5 per-GPU batch)
    python tools/dump_csr.py protein256 /tmp/protein256.csr (config 4)
    python tools/dump_csr.py community64 /tmp/community64.csr (config 2)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnf_amd import datasets as D  # noqa: E402


def batch(name):
    if name == "ego128":
        pool = D.synthetic_ego(128)
        ids = np.arange(128)
    elif name == "protein256":
        pool = D.synthetic_protein(256)
        ids = np.arange(256)
    elif name == "community64":
        ds = D.GraphDataset("graph_rnn_community_medium", 64)
        pool, ids = ds.all, ds.sample_ids(64)
    else:
        raise SystemExit("unknown batch " + name)
    S, R, off = [], [], 0
    for g in ids:
        n, s, r = pool.graph(int(g))
        S.append(s.astype(np.int64) + off)
        R.append(r.astype(np.int64) + off)
        off += n
    return off, np.concatenate(S), np.concatenate(R)


def main():
    name, path = sys.argv[1], sys.argv[2]
    n, s, r = batch(name)
    order = np.argsort(r, kind="stable")       # stable within a receiver = the edge order of the list
    col = s[order].astype(np.int32)
    rowptr = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(r, minlength=n), out=rowptr[1:])
    deg = np.diff(rowptr)
    with open(path, "wb") as f:
        np.array([n, len(col)], np.int64).tofile(f)
        rowptr.tofile(f)
        col.tofile(f)
    print(f"{name}: N={n} E={len(col)} in-degree mean {deg.mean():.2f} max {deg.max()} "
          f"rows>16: {(deg > 16).sum()} rows>32: {(deg > 32).sum()} rows>64: {(deg > 64).sum()}")


if __name__ == "__main__":
    main()
