"""Gradient parity of one mid-size batch under whatever dW launch shape the library options force (GNF_OPTIONS=
"dw_grouped=1,..." -> gnf_set_option through the binding; a script so that tests/test_train_gpu.py can run every
shape in its own process).  argv[1]: comma list of "ws" (weight sharing), "serial" (no auxiliary dW stream), "nostash"
(the backward walk recomputes the MLP rows), "big" (3 300 nodes: more than 192 tiles, the walk whose dW GEMMs go to the
auxiliary stream).  Ragged layer widths exercise partial 128 x 128 tiles, thin strips (1 x 8 / 8 x 1 wave layouts),
several node chunks per job and cheap units riding behind the costly ones; weight sharing exercises the
accumulating reduce.  Prints 'dw-modes-ok' on success."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import gnf_oracle as O
from helpers import graph_from_arrays, make_product_grevnet
from gnf_amd.train import GRevNetTrainer
from gnf_amd.datasets import senders_receivers

flags = sys.argv[1].split(",") if len(sys.argv) > 1 else []
ws = "ws" in flags
rng = np.random.default_rng(5)
n_node = rng.integers(9, 17, size=264 if "big" in flags else 72).astype(np.int32)   # ~900 nodes: 4 node chunks in the plan
s, r, ne = senders_receivers(n_node)
n = int(n_node.sum())
D, L, K, T = 48, 200, 3, 2                                      # widths 24 -> 200 -> 200 -> 24
x = (rng.standard_normal((n, D)) * 0.7).astype(np.float32)
kw = dict(agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu")
p = O.make_grevnet_params(21, D // 2, L, K, T, final_scale=0.3, weight_sharing=ws)
ref = O.loss_and_grads(s, r, n, x, p, T, weight_sharing=ws, **kw)
net = make_product_grevnet(dict(D=D, latent=L, K=K, T=T, weight_sharing=ws, **kw), p)
tr = GRevNetTrainer(net)
tr.overlap_weight_grads = "serial" not in flags
tr.stash_mlp_rows = "nostash" not in flags
assert ("big" in flags) == (n > 192 * 16)
out = tr.loss_and_grads(graph_from_arrays(n_node, ne, s, r, x, "cuda:0"))
torch.cuda.synchronize()
assert abs(float(out["loss_per_node"]) - ref["total_loss"] / n) <= 1e-4
got = tr.named_gradients()
worst = 0.0
for kind in "st":
    a_nets = got[kind] if ws else got[kind][0] + got[kind][1]
    b_nets = ref["grads"][kind] if ws else ref["grads"][kind][0] + ref["grads"][kind][1]
    for a_net, b_net in zip(a_nets, b_nets):
        for (aw, ab), (bw, bb) in zip(a_net, b_net):
            for a, b in ((aw, bw), (ab, bb)):
                err = float(np.abs(a - b).max()) / (float(np.abs(b).max()) + 1e-12)
                worst = max(worst, err)
assert worst <= 1e-3, worst
print(f"dw-modes-ok worst relative error {worst:.2e}")
