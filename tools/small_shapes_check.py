"""Developer check: training gradients vs the oracle on small / odd shapes (D = 2: one feature per half; a 7-node and a
5-node batch; K = 1; single-node graphs) through whatever backward path the environment selects (GNF_BWD_GENERIC=1 for
the GEMM path): thin 1 x 8 / 8 x 1 wave layouts of the dW kernel, single-chunk launches, partial tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import gnf_oracle as O
from helpers import graph_from_arrays, make_product_grevnet
from gnf_amd.train import GRevNetTrainer
from gnf_amd.datasets import senders_receivers
rng = np.random.default_rng(5)
for (D, L, K, T, ng) in ((2, 256, 5, 3, 40), (2, 64, 2, 2, 1), (4, 40, 1, 2, 3), (6, 300, 3, 2, 90)):
    n_node = rng.integers(1, 9, size=ng).astype(np.int32)
    s, r, ne = senders_receivers(n_node)
    n = int(n_node.sum())
    x = (rng.standard_normal((n, D)) * 0.7).astype(np.float32)
    kw = dict(agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu")
    p = O.make_grevnet_params(21, D // 2, L, K, T, final_scale=0.3)
    ref = O.loss_and_grads(s, r, n, x, p, T, **kw)
    net = make_product_grevnet(dict(D=D, latent=L, K=K, T=T, weight_sharing=False, **kw), p)
    tr = GRevNetTrainer(net)
    out = tr.loss_and_grads(graph_from_arrays(n_node, ne, s, r, x, "cuda:0"))
    torch.cuda.synchronize()
    got = tr.named_gradients()
    worst = 0.0
    gmax = 0.0
    for kind in "st":
        for a_net, b_net in zip(got[kind][0] + got[kind][1], ref["grads"][kind][0] + ref["grads"][kind][1]):
            for (aw, ab), (bw, bb) in zip(a_net, b_net):
                for a, b in ((aw, bw), (ab, bb)):
                    gmax = max(gmax, float(np.abs(b).max()))
    for kind in "st":
        for a_net, b_net in zip(got[kind][0] + got[kind][1], ref["grads"][kind][0] + ref["grads"][kind][1]):
            for (aw, ab), (bw, bb) in zip(a_net, b_net):
                for a, b in ((aw, bw), (ab, bb)):
                    worst = max(worst, float(np.abs(a - b).max()) / (float(np.abs(b).max()) + 1e-3 * gmax))
    print(f"D={D} L={L} K={K} T={T} n={n}: loss err {abs(float(out['loss_per_node']) - ref['total_loss']/n):.2e} worst grad rel err {worst:.2e}")
