"""include/gnf.h promises that every entry point is legal inside hipGraph stream capture (launches only on the
caller's stream, no allocation, no host synchronisation).  Here the whole-flow entry points are captured with
torch.cuda.graph (hipStreamBeginCapture on torch's current stream, which is the stream the binding hands to the
library), replayed, and compared BITWISE with the eager call - also after the inputs changed in place, which is
what a replayed graph is for.  gnf_grevnet_backward_f32 forks its weight-gradient GEMMs onto a second stream with
events: under capture that is a cross-stream fork / join inside one graph.
"""
import numpy as np
import pytest
import torch

from helpers import graph_from_arrays, make_product_grevnet
from oracle import gnf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HP = dict(D=64, latent=256, K=5, T=8, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
          weight_sharing=False)


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from gnf_amd import _abi
    _abi.lib()


def _config2(community_medium, graphs=64, seed=12345):
    rng = np.random.default_rng(seed)
    ids = rng.choice(168, size=graphs, replace=True)
    nn, ne, s, r = O.batch_graphs(*community_medium, ids)
    n = int(nn.sum())
    x = rng.standard_normal((n, 64)).astype(np.float32)
    x2 = rng.standard_normal((n, 64)).astype(np.float32)
    p = O.make_grevnet_params(99, 32, 256, 5, 8, final_scale=0.25)
    return graph_from_arrays(nn, ne, s, r, x, DEV), torch.as_tensor(x2).to(DEV), p


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
def test_forward_and_inverse_replay_bitwise(community_medium, fused):
    from gnf_amd.flow import forward_shard_sums
    graph, x2, p = _config2(community_medium)
    net = make_product_grevnet(HP, p)
    net.fused = fused
    # eager reference on both inputs (also the first launches: CSR / flow caches, per-device kernel attributes)
    x1 = graph.nodes.clone()
    eager = {}
    for name, x in (("x1", x1), ("x2", x2)):
        graph.nodes.copy_(x)
        z, s3 = forward_shard_sums(net, graph)
        back = net(graph.replace(nodes=z), inverse=False).nodes
        eager[name] = (z.clone(), s3.clone(), back.clone())
    graph.nodes.copy_(x1)
    torch.cuda.synchronize()
    sums = torch.zeros(3, dtype=torch.float64, device=DEV)
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        z_c, _ = forward_shard_sums(net, graph, sums)            # gnf_grevnet_f32(GNF_FORWARD)
        back_c = net(graph.replace(nodes=z_c), inverse=False).nodes   # gnf_grevnet_f32(GNF_INVERSE)
    for name, x in (("x1", x1), ("x2", x2), ("x1", x1)):
        graph.nodes.copy_(x)
        z_c.zero_(), back_c.zero_(), sums.zero_()
        cg.replay()
        torch.cuda.synchronize()
        z_e, s_e, b_e = eager[name]
        assert torch.equal(z_c, z_e), name
        assert torch.equal(sums[:2], s_e[:2]), name
        assert torch.equal(back_c, b_e), name


@pytest.mark.parametrize("graphs", [48, 90], ids=["merged_launch", "forked_dw_stream"])
def test_backward_replay_bitwise(community_medium, graphs):
    """The training step under capture: the walk of batches of up to 192 tiles (one launch per half-step carrying the
    previous half-step's weight-gradient GEMMs) and the one larger batches take (dW GEMMs forked to the auxiliary stream
    with per-call events)."""
    _backward_replay(community_medium, graphs)


def _backward_replay(community_medium, graphs):
    from gnf_amd.graphs import csr_of
    from gnf_amd.train import GRevNetTrainer
    graph, x2, p = _config2(community_medium, graphs=graphs, seed=7)
    assert (graph.nodes.shape[0] > 192 * 16) == (graphs > 48)   # 192 tiles: where the merged launch ends
    net = make_product_grevnet(HP, p)
    tr = GRevNetTrainer(net)
    assert tr.overlap_weight_grads
    x1 = graph.nodes.clone()
    eager = {}
    for name, x in (("x1", x1), ("x2", x2)):
        graph.nodes.copy_(x)
        out = tr.loss_and_grads(graph)                             # also warms caches, workspace, aux stream
        torch.cuda.synchronize()
        eager[name] = (tr.grad.clone(), out["total_loss"].clone(), out["reconstruction"].clone())
    assert float((eager["x1"][0] - eager["x2"][0]).abs().max()) > 0
    graph.nodes.copy_(x1)
    csr_of(graph), csr_of(graph, by_sender=True)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        out_c = tr.loss_and_grads(graph)                           # gnf_grevnet_f32 + gnf_grevnet_backward_f32
    for name, x in (("x2", x2), ("x1", x1)):
        graph.nodes.copy_(x)
        tr.grad.zero_()
        cg.replay()
        torch.cuda.synchronize()
        g_e, loss_e, rec_e = eager[name]
        assert torch.equal(out_c["total_loss"], loss_e), name
        assert torch.equal(out_c["reconstruction"], rec_e), name
        assert torch.equal(tr.grad, g_e), name
