"""Developer check: one big batch (thousands of graphs, ~170k nodes) through forward / inverse / a training step:
grid limits, 64-bit indexing, workspace sizes.  Size-independent properties only (round trip, finite values,
fused == layered log-prob)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import gnf_oracle as O
from helpers import make_product_grevnet
from gnf_amd import datasets as D
from gnf_amd.flow import log_prob_terms
from gnf_amd.graphs import data_dicts_to_graphs_tuple
from gnf_amd.train import GRevNetTrainer

g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ds = D.GraphDataset("graph_rnn_community_medium", 64, seed=1)
ids = ds.sample_ids(g)
rng = np.random.default_rng(0)
graph = data_dicts_to_graphs_tuple(ds.all.data_dicts(ids, lambda n: rng.standard_normal((n, 64)).astype(np.float32)), "cuda:0")
n, e = graph.nodes.shape[0], graph.senders.shape[0]
print(f"{g} graphs, {n} nodes, {e} edges")
for bn in (False, True):
    hp = dict(D=64, latent=256, K=5, T=8, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu", weight_sharing=False)
    p = O.make_grevnet_params(99, 32, 256, 5, 8, final_scale=0.25)
    if bn:
        p["bn"] = O.make_bn_params(5, 32, 8)
    net = make_product_grevnet(hp, p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = log_prob_terms(net, graph)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lp = float(out["log_prob_xs_per_node"])
    lay = make_product_grevnet(hp, p); lay.fused = False
    lpl = float(log_prob_terms(lay, graph)["log_prob_xs_per_node"])
    if bn:
        for half in net.bns:
            for b in half:
                b.moving_mean.copy_(b.batch_mean); b.moving_variance.copy_(b.batch_variance)
    back = net(out["z_graph"], inverse=False).nodes
    rt = float((back - graph.nodes).abs().max())
    tr = GRevNetTrainer(net, lr=1e-5, use_lr_decay=False)
    tr.overlap_weight_grads = os.environ.get("NO_OVERLAP") is None
    v = tr.step(graph); torch.cuda.synchronize()
    t2 = time.perf_counter(); v = tr.step(graph); torch.cuda.synchronize(); t3 = time.perf_counter()
    rec = float((v["reconstruction"] - graph.nodes).abs().max())
    print(f"bn={bn}: log-prob/node {lp:.5f} (layered {lpl:.5f}), round trip {rt:.2e}, forward {1e3*(t1-t0):.1f} ms (first call), "
          f"train step {1e3*(t3-t2):.1f} ms, reconstruction {rec:.2e}, grad finite {bool(torch.isfinite(tr.grad).all())}, "
          f"{n*8/(t3-t2)/1e6:.1f} M node-updates/s training")
    assert np.isfinite(lp) and abs(lp - lpl) < 1e-4 and rt < 5e-3 and rec < 5e-3
print("ok")
