"""graph-normalizing-flows_amd: MI355X (gfx950) implementation of the GRevNet forward / inverse +
log-det hot path of jliu/graph-normalizing-flows behind the reference's own gnn.py call surface.

Import as `gnf_amd` (the directory name has a hyphen; gnf_amd.py at the repo root is the loader):
    from gnf_amd.gnn import GRevNet, avg_then_mlp_gnn, make_mlp_model, leaky_relu
"""
from . import _abi
from .graphs import GraphsTuple, data_dicts_to_graphs_tuple, build_csr_host, csr_of

__all__ = ["GraphsTuple", "data_dicts_to_graphs_tuple", "build_csr_host", "csr_of", "_abi"]
