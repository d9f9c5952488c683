"""Training step on the GPU: gnf_grevnet_backward_f32 / gnf_adam_f32 / clippers /
gnf_pack_flow through the product's GRevNetTrainer vs the CPU oracle (torch-autograd float64 restatement,
pinned by finite differences in test_oracle.py).

Tolerances: a gradient tensor is compared with atol = 3e-4 * max|g| of that tensor (+1e-5) - fp32 GEMMs
reducing over thousands of nodes against a float64 reference; Adam / clipping are elementwise fp32:
rtol 2e-6.  The deep (T = 8, K = 5) case states a second bound: relu / leaky_relu are not differentiable
at 0, so a pre-activation within fp32 rounding of 0 can take the other branch of act' than the float64
oracle does and shifts every earlier layer's gradient of that one net by O(1e-3) of its scale (measured:
2 nets of 32 at 4e-3 .. 7e-3; a float32 autograd run of the oracle itself shows the same 7e-3 worst case);
there, 90 % of the tensors must meet 1e-3 and all of them 2e-2."""
import numpy as np
import pytest
import torch

from helpers import graph_from_arrays, make_product_grevnet
from oracle import gnf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from gnf_amd import _abi
    _abi.lib()


def _batch(dataset, ids):
    n_node, n_edge, sl, rl = dataset
    return O.batch_graphs(n_node, n_edge, sl, rl, ids)


def _flat(grads, ws):
    for kind in ("s", "t"):
        nets = grads[kind] if ws else grads[kind][0] + grads[kind][1]
        for q, net in enumerate(nets):
            for j, (w, b) in enumerate(net):
                yield f"{kind}[{q}].W{j}", w
                yield f"{kind}[{q}].b{j}", b


def _check_grads(got, ref, ws, scale=3e-4):
    # a gradient that is zero by cancellation (the last bias of a t-net in front of a batch-norm bijector: the
    # bijector removes the mean again) is judged against the flow's overall gradient scale
    gmax = max(float(np.abs(b).max()) for _, b in _flat(ref, ws))
    for (name, a), (_, b) in zip(_flat(got, ws), _flat(ref, ws)):
        tol = scale * float(np.abs(b).max()) + 1e-5 + 1e-6 * gmax
        err = float(np.abs(a - b).max())
        assert err <= tol, f"{name}: max err {err:.3e} > {tol:.3e} (max|g| {np.abs(b).max():.3e})"


CASES = [
    # D, latent, K, T, agg, combine, eps, act, ws
    (8, 32, 3, 2, "mean", "agg", 1.0, "leaky_relu", False),
    (6, 20, 2, 2, "sum", "concat", 0.0, "relu", False),
    (12, 48, 4, 3, "mean", "concat", 0.0, "leaky_relu", True),      # weight sharing: uses summed
    (10, 24, 1, 2, "sum", "agg", 0.5, "leaky_relu", False),         # K = 1: a single Linear layer
    (64, 256, 5, 2, "mean", "agg", 1.0, "leaky_relu", False),       # BASELINE widths
    (100, 70, 3, 1, "mean", "agg", 1.0, "relu", False),             # ragged in every GEMM dimension
]


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "gemm"])
@pytest.mark.parametrize("case", CASES, ids=[f"D{c[0]}_L{c[1]}_K{c[2]}_T{c[3]}_{c[4]}_{c[5]}{'_ws' if c[8] else ''}" for c in CASES])
def test_gradients_vs_oracle(grid_small, case, fused):
    """fused: the LDS-resident backward kernel (gnf_fused_bwd.hip, needs the packed weights); gemm: the generic
    GEMM building blocks (what layers too wide for LDS run)."""
    from gnf_amd.train import GRevNetTrainer
    d, latent, k, t, agg, combine, eps, act, ws = case
    hp = dict(D=d, latent=latent, K=k, T=t, agg=agg, combine=combine, epsilon=eps, activation=act, weight_sharing=ws)
    nn, ne, s, r = _batch(grid_small, list(range(12)))
    n = int(nn.sum())
    rng = np.random.default_rng(d * 7 + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    p = O.make_grevnet_params(d + k, d // 2, latent, k, t, combine=combine, weight_sharing=ws,
                              final_scale=0.3 if agg == "mean" else 0.1)
    ref = O.loss_and_grads(s, r, n, x, p, t, ws, agg=agg, combine=combine, epsilon=eps, activation=act)
    net = make_product_grevnet(hp, p)
    net.fused = fused
    tr = GRevNetTrainer(net)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    out = tr.loss_and_grads(graph)
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    # reversible backprop rebuilds the input on its way back
    np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=2e-4, rtol=2e-4)
    _check_grads(tr.named_gradients(), ref["grads"], ws)
    # calling it again overwrites (does not accumulate into) the gradient
    tr.loss_and_grads(graph)
    torch.cuda.synchronize()
    _check_grads(tr.named_gradients(), ref["grads"], ws)


def test_gradients_config2_batch(community_medium):
    """community_medium batch of 32 graphs, BASELINE hyper-parameters (D=64, L=256, K=5, T=8)."""
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=64, latent=256, K=5, T=8, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    rng = np.random.default_rng(77)
    nn, ne, s, r = _batch(community_medium, rng.choice(168, size=32, replace=True))
    n = int(nn.sum())
    x = rng.standard_normal((n, 64)).astype(np.float32)
    p = O.make_grevnet_params(99, 32, 256, 5, 8, final_scale=0.25)
    ref = O.loss_and_grads(s, r, n, x, p, 8, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu")
    tr = GRevNetTrainer(make_product_grevnet(hp, p))
    out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
    torch.cuda.synchronize()
    assert abs(float(out["loss_per_node"]) - ref["total_loss"] / n) <= 1e-4
    np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=2e-3, rtol=2e-3)
    rel = [float(np.abs(a - b).max() / np.abs(b).max())
           for (_, a), (_, b) in zip(_flat(tr.named_gradients(), False), _flat(ref["grads"], False))]
    assert max(rel) <= 2e-2, max(rel)
    assert np.mean(np.array(rel) <= 1e-3) >= 0.9, sorted(rel)[-40:]


@pytest.mark.parametrize("flavour", ["plain", "batch_norm", "weight_sharing_concat"])
def test_mlp_row_stash_matches_the_recomputing_walk(community_medium, flavour):
    """GnfFlow.mlp_stash (ABI v8): the training forward leaves the rows the backward walk would recompute; the gradients,
    the loss and the reconstructed input agree with the fully reversible walk (stash_mlp_rows=False) to rounding, and
    with the oracle.  The stash is only offered where the library uses it (gnf_mlp_stash_bytes)."""
    import ctypes as C
    from gnf_amd import _abi
    from gnf_amd.train import GRevNetTrainer
    combine = "concat" if flavour == "weight_sharing_concat" else "agg"
    ws = flavour == "weight_sharing_concat"
    hp = dict(D=24, latent=96, K=4, T=3, agg="mean", combine=combine, epsilon=1.0, activation="leaky_relu", weight_sharing=ws)
    nn, ne, s, r = _batch(community_medium, [3, 50, 77, 12, 100, 5, 9, 130])
    n = int(nn.sum())
    x = np.random.default_rng(4).standard_normal((n, hp["D"])).astype(np.float32)
    p = O.make_grevnet_params(31, hp["D"] // 2, hp["latent"], hp["K"], hp["T"], combine=combine, weight_sharing=ws, final_scale=0.3)
    if flavour == "batch_norm":
        p["bn"] = O.make_bn_params(32, hp["D"] // 2, hp["T"])
    ref = O.loss_and_grads(s, r, n, x, p, hp["T"], ws, agg="mean", combine=combine, epsilon=1.0, activation="leaky_relu")
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    res = {}
    for stash in (True, False):
        net = make_product_grevnet(hp, p)
        tr = GRevNetTrainer(net)
        tr.stash_mlp_rows = stash
        out = tr.loss_and_grads(graph)
        torch.cuda.synchronize()
        assert (tr._mlp_stash is not None) == stash
        if stash:
            flow = net._flow(hp["D"] // 2, torch.device(DEV))
            assert tr._mlp_stash.numel() == _abi.lib().gnf_mlp_stash_bytes(n, hp["D"], C.byref(flow)) > 0
        assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
        np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=2e-4, rtol=2e-4)
        _check_grads(tr.named_gradients(), ref["grads"], ws)
        res[stash] = (tr.grad.detach().cpu().numpy().copy(), float(out["total_loss"]))
    scale = float(np.abs(res[False][0]).max())
    assert float(np.abs(res[True][0] - res[False][0]).max()) <= 2e-5 * scale
    assert abs(res[True][1] - res[False][1]) <= 1e-9 * abs(res[False][1])      # the same forward arithmetic


@pytest.mark.parametrize("gnn_kind", ["avg_then_mlp", "dm_self_attn"])
@pytest.mark.parametrize("graphs", [10, 30, 52, 73])
def test_merged_launch_thin_units_over_batch_sizes(community_medium, graphs, gnn_kind):
    """The merged backward + dW launch of the stash walk hands the thin layers' weight-gradient units to its backward-tile
    workgroups (WideGemmT.light_on_tiles, round 6): about 25 / 75 / 130 / 185 tiles beside 224 / 160 / 96 / 64 dW
    workgroups, thin jobs cut for the tile count.  Reference: the same walk with the units left on the dW workgroups
    (option dw_thin_on_dw, the round-5 dealing, pinned to the oracle by the tests above) - same stash, same activations,
    only the cut of the node axis of the thin layers' sums differs: gradients equal to rounding."""
    from gnf_amd import _abi
    from gnf_amd.train import GRevNetTrainer
    rng = np.random.default_rng(graphs)
    nn, ne, s, r = _batch(community_medium, rng.choice(168, size=graphs, replace=False))
    n = int(nn.sum())
    assert (n + 15) // 16 <= 192
    if gnn_kind == "dm_self_attn":
        attn = dict(num_heads=8, kq_dim=10, v_dim=10, out_dim=80, concat=True, kq_dim_division=True, residual=False)
        hp = dict(D=64, latent=256, K=5, T=2, agg="mean", combine="agg", epsilon=0.0, activation="relu", weight_sharing=False,
                  attn=attn)
        p = O.make_attn_grevnet_params(7, 32, 256, 5, 2, weight_sharing=False, final_scale=0.3, **attn)
    else:
        hp = dict(D=64, latent=256, K=5, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu", weight_sharing=False)
        p = O.make_grevnet_params(7, 32, 256, 5, 2, final_scale=0.3)
    x = rng.standard_normal((n, 64)).astype(np.float32)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    res = {}
    try:
        for on_dw in (0, 1):
            _abi.set_option("dw_thin_on_dw", on_dw)
            tr = GRevNetTrainer(make_product_grevnet(hp, p))
            out = tr.loss_and_grads(graph)
            torch.cuda.synchronize()
            assert tr._mlp_stash is not None
            np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
            res[on_dw] = (tr.grad.detach().cpu().numpy().copy(), float(out["total_loss"]))
    finally:
        _abi.set_option("dw_thin_on_dw", 0)
    assert np.isfinite(res[0][0]).all()
    scale = float(np.abs(res[1][0]).max())
    assert float(np.abs(res[0][0] - res[1][0]).max()) <= 1e-5 * scale
    assert res[0][1] == res[1][1]


def test_mlp_row_stash_is_not_offered_where_it_would_not_be_used(community_medium, grid_small):
    """gnf_mlp_stash_bytes: 0 for batches of more than 256 16-node tiles and for blocks that end in snt.LayerNorm (their
    half-steps run one net per workgroup); attention nets without it do get one."""
    import ctypes as C
    from gnf_amd import _abi
    lib = _abi.lib()
    nn, ne, s, r = _batch(grid_small, [0, 1])
    for ln, want in ((False, True), (True, False)):
        attn = dict(num_heads=2, kq_dim=3, v_dim=4, out_dim=6, concat=True, kq_dim_division=True, residual=False, layer_norm=ln)
        hp_a = dict(D=8, latent=32, K=2, T=1, agg="mean", combine="agg", epsilon=0.0, activation="relu", weight_sharing=False, attn=attn)
        net_a = make_product_grevnet(hp_a, O.make_attn_grevnet_params(5, 4, 32, 2, 1, **attn))
        net_a(graph_from_arrays(nn, ne, s, r, np.zeros((int(nn.sum()), 8), np.float32), DEV))
        assert (lib.gnf_mlp_stash_bytes(500, 8, C.byref(net_a._flow(4, torch.device(DEV)))) > 0) == want
    hp = dict(D=16, latent=64, K=3, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu", weight_sharing=False)
    net = make_product_grevnet(hp, O.make_grevnet_params(6, 8, 64, 3, 2))
    net(graph_from_arrays(nn, ne, s, r, np.zeros((int(nn.sum()), 16), np.float32), DEV))
    flow = net._flow(8, torch.device(DEV))
    assert lib.gnf_mlp_stash_bytes(256 * 16, 16, C.byref(flow)) > 0
    assert lib.gnf_mlp_stash_bytes(256 * 16 + 1, 16, C.byref(flow)) == 0


def test_isolated_nodes_and_directed_edges():
    """A directed, asymmetric edge list with isolated nodes: the backward aggregation runs over the by-sender
    CSR, which is NOT the receiver CSR here."""
    from gnf_amd.train import GRevNetTrainer
    s = np.array([0, 0, 2, 4, 4, 4], np.int32)
    r = np.array([1, 2, 1, 0, 1, 4], np.int32)      # node 3 isolated; node 4 has a self loop
    n = 5
    for agg in ("sum", "mean"):
        hp = dict(D=4, latent=8, K=2, T=2, agg=agg, combine="agg", epsilon=1.0, activation="leaky_relu",
                  weight_sharing=False)
        x = np.random.default_rng(1).standard_normal((n, 4)).astype(np.float32)
        p = O.make_grevnet_params(1, 2, 8, 2, 2, final_scale=0.5)
        ref = O.loss_and_grads(s, r, n, x, p, 2, agg=agg, combine="agg", epsilon=1.0, activation="leaky_relu")
        tr = GRevNetTrainer(make_product_grevnet(hp, p))
        tr.loss_and_grads(graph_from_arrays([5], [6], s, r, x, DEV))
        torch.cuda.synchronize()
        _check_grads(tr.named_gradients(), ref["grads"], False)


def test_adam_and_clipping_vs_oracle():
    import ctypes as C
    from gnf_amd import _abi
    lib = _abi.lib()
    rng = np.random.default_rng(3)
    n = 100003
    w0, g0 = rng.standard_normal(n).astype(np.float32), (rng.standard_normal(n) * 3).astype(np.float32)
    w, g = torch.tensor(w0, device=DEV), torch.tensor(g0, device=DEV)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    wr, mr, vr = w0.astype(np.float64), np.zeros(n), np.zeros(n)
    import math
    st = _abi.stream_ptr(torch.device(DEV))
    for t in (1, 2, 3):
        lr_t = 1e-2 * math.sqrt(1 - 0.9 ** t) / (1 - 0.9 ** t)
        _abi.check(lib.gnf_adam_f32(_abi.ptr(w), _abi.ptr(g), _abi.ptr(m), _abi.ptr(v), n, lr_t, 0.9, 0.9, 1e-8, st), "adam")
        wr, mr, vr = O.adam_step(wr, g0, mr, vr, t, 1e-2, 0.9, 0.9, 1e-8)
    torch.cuda.synchronize()
    np.testing.assert_allclose(w.cpu().numpy(), wr, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(m.cpu().numpy(), mr, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(v.cpu().numpy(), vr, rtol=2e-6, atol=1e-7)
    # clip by value
    gc = g.clone()
    _abi.check(lib.gnf_clip_by_value_f32(_abi.ptr(gc), n, -1.0, 5.0, st), "clip value")
    np.testing.assert_array_equal(gc.cpu().numpy(), np.clip(g0, -1.0, 5.0))
    # clip by norm per tensor (three tensors of very different sizes, one of them empty)
    bounds = np.array([0, 10, 10, 70000, n], np.int64)
    off = torch.tensor(bounds, device=DEV)
    nt = len(bounds) - 1
    cws = torch.empty(lib.gnf_clip_workspace_bytes(nt), dtype=torch.uint8, device=DEV)
    for ws_ptr, ws_bytes in ((None, 0), (_abi.ptr(cws), cws.numel())):   # one workgroup per tensor / two-pass sliced
        gn = g.clone()
        _abi.check(lib.gnf_clip_by_norm_f32(_abi.ptr(gn), _abi.ptr(off), nt, 10.0, ws_ptr, ws_bytes, st), "clip norm")
        got = gn.cpu().numpy()
        for a, b in zip(bounds[:-1], bounds[1:]):
            if b > a:
                np.testing.assert_allclose(got[a:b], O.clip_by_norm(g0[a:b], 10.0), rtol=3e-6, atol=1e-7)
    with pytest.raises(_abi.GnfError, match="workspace"):
        _abi.check(lib.gnf_clip_by_norm_f32(_abi.ptr(gn), _abi.ptr(off), nt, 10.0, _abi.ptr(cws), 8, st), "clip norm")


def test_training_loop_reduces_loss_and_keeps_packed_weights_fresh(grid_small):
    """A few iterations of run_grevnet.py:440-447: the loss goes down, and after every step the fused kernels
    (which read the re-packed matrix-core copy) agree with the layered kernels (which read W / b)."""
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=8, latent=32, K=3, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    nn, ne, s, r = _batch(grid_small, list(range(12)))
    n = int(nn.sum())
    x = (np.random.default_rng(0).standard_normal((n, 8)) * 2 + 1).astype(np.float32)
    p = O.make_grevnet_params(4, 4, 32, 3, 2, final_scale=0.3)
    net = make_product_grevnet(hp, p)
    tr = GRevNetTrainer(net, lr=3e-3, use_lr_decay=False, clip_gradient_by_norm=True, clip_gradient_norm=50.0)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    losses = []
    for _ in range(30):
        losses.append(float(tr.step(graph)["loss_per_node"]))
    assert losses[-1] < losses[0] - 0.05, losses
    fused = float(log_prob_terms(net, graph)["log_prob_xs_per_node"])
    net.fused = False
    layered = float(log_prob_terms(net, graph)["log_prob_xs_per_node"])
    assert abs(fused - layered) <= 1e-5
    # one oracle step from the same start reproduces the first update
    ref = O.loss_and_grads(s, r, n, x, p, 2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu")
    net2 = make_product_grevnet(hp, p)
    tr2 = GRevNetTrainer(net2, lr=3e-3, use_lr_decay=False)
    tr2.step(graph)
    torch.cuda.synchronize()
    w_new = net2.get_params()["s"][0][0][0][0]
    g = ref["grads"]["s"][0][0][0][0]
    w_ref, _, _ = O.adam_step(p["s"][0][0][0][0], g, 0 * g, 0 * g, 1, 3e-3, 0.9, 0.9, 1e-8)
    big = np.abs(g) > 1e-3 * np.abs(g).max()      # where |g| ~ 0 the first Adam step is sign-sensitive
    np.testing.assert_allclose(w_new[big], w_ref[big], atol=2e-5)


def _flat_attn(grads, ws):
    for kind in ("s", "t"):
        nets = grads[kind] if ws else grads[kind][0] + grads[kind][1]
        for q, net in enumerate(nets):
            for key in ("wq", "wk", "wv", "wo", "ln_gamma", "ln_beta"):
                if key in net["attn"]:
                    yield f"{kind}[{q}].{key}", net["attn"][key]
            for j, (w, b) in enumerate(net["mlp"]):
                yield f"{kind}[{q}].W{j}", w
                yield f"{kind}[{q}].b{j}", b


ATTN_TRAIN_CASES = [
    # D, latent, K, T, heads, kq, v, C, concat, division, residual, weight sharing
    (64, 64, 3, 2, 8, 10, 10, 80, True, False, False, False),     # the drivers' default head geometry
    (8, 24, 2, 2, 3, 4, 5, 6, False, True, False, True),          # no concat, scaled logits, weight sharing
    (12, 32, 3, 1, 2, 7, 3, 9, True, False, True, False),         # residual
]


@pytest.mark.parametrize("stash", [True, False], ids=["stash", "recompute"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "gemm"])
@pytest.mark.parametrize("case", ATTN_TRAIN_CASES, ids=[f"D{c[0]}_h{c[4]}_kq{c[5]}_v{c[6]}_C{c[7]}" for c in ATTN_TRAIN_CASES])
def test_attention_gradients_vs_oracle(community_medium, case, fused, stash):
    """The drivers' default GNN (run_grevnet.py:56,199-211) trains: gradients of wq, wk, wv, wo and of the MLP
    behind the attention front-end vs the autograd oracle (itself pinned by finite differences) - with the
    front-end stashed by the forward pass (GnfFlow.attn_stash, the default) and recomputed by the backward walk."""
    from gnf_amd.train import GRevNetTrainer
    d, latent, k, t, nh, kq, vd, c, concat, div, res, ws = case
    attn = dict(num_heads=nh, kq_dim=kq, v_dim=vd, out_dim=c, concat=concat, kq_dim_division=div, residual=res)
    hp = dict(D=d, latent=latent, K=k, T=t, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=ws, attn=attn)
    nn, ne, s, r = _batch(community_medium, [3, 50, 77, 12, 100])
    n = int(nn.sum())
    rng = np.random.default_rng(d + nh)
    x = (rng.standard_normal((n, d)) * (0.3 if res else 1.0)).astype(np.float32)
    p = O.make_attn_grevnet_params(d + 1, d // 2, latent, k, t, weight_sharing=ws, final_scale=0.3, **attn)
    ref = O.loss_and_grads(s, r, n, x, p, t, ws, activation="relu")
    net = make_product_grevnet(hp, p)
    net.fused = fused
    tr = GRevNetTrainer(net)
    tr.stash_attention = stash
    tr.stash_mlp_rows = stash          # "recompute": the fully reversible walk (neither stash)
    out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
    torch.cuda.synchronize()
    assert (tr._stash is not None) == stash
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
    for (name, a), (_, b) in zip(_flat_attn(tr.named_gradients(), ws), _flat_attn(ref["grads"], ws)):
        tol = 5e-4 * float(np.abs(b).max()) + 1e-5
        err = float(np.abs(a - b).max())
        assert err <= tol, f"{name}: max err {err:.3e} > {tol:.3e} (max|g| {np.abs(b).max():.3e})"


ATTN_LN_TRAIN_CASES = [
    # D, latent, K, T, heads, kq, v, C, concat, residual, weight sharing       (all with layer_norm=True, gnn.py:550-552)
    (64, 64, 3, 2, 8, 10, 10, 80, True, True, False),     # --attn_layer_norm --attn_residual on the default head geometry
    (8, 24, 2, 3, 3, 4, 5, 6, False, False, True),        # no concat, no residual, weight sharing (gradients accumulate)
    (260, 32, 1, 1, 2, 4, 4, 8, True, True, False),       # H = 130 > 2 wave widths; K = 1 (normalisation right behind layer 0)
]


@pytest.mark.parametrize("stash", [True, False], ids=["stash", "recompute"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "gemm"])
@pytest.mark.parametrize("case", ATTN_LN_TRAIN_CASES, ids=[f"D{c[0]}_h{c[4]}_C{c[7]}" for c in ATTN_LN_TRAIN_CASES])
def test_attention_layer_norm_gradients_vs_oracle(community_medium, case, fused, stash):
    """DMSelfAttentionMLP(layer_norm=True): gradients of ln_gamma / ln_beta and of everything in front of the
    normalisation vs the autograd oracle (pinned by finite differences in tests/test_oracle.py)."""
    from gnf_amd.train import GRevNetTrainer
    d, latent, k, t, nh, kq, vd, c, concat, res, ws = case
    attn = dict(num_heads=nh, kq_dim=kq, v_dim=vd, out_dim=c, concat=concat, kq_dim_division=False, residual=res,
                layer_norm=True)
    hp = dict(D=d, latent=latent, K=k, T=t, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=ws, attn=attn)
    nn, ne, s, r = _batch(community_medium, [3, 50, 77, 12, 100])
    n = int(nn.sum())
    rng = np.random.default_rng(d + nh)
    x = rng.standard_normal((n, d)).astype(np.float32)
    p = O.make_attn_grevnet_params(d + 1, d // 2, latent, k, t, weight_sharing=ws, final_scale=0.3, **attn)
    ref = O.loss_and_grads(s, r, n, x, p, t, ws, activation="relu")
    net = make_product_grevnet(hp, p)
    net.fused = fused
    tr = GRevNetTrainer(net)
    tr.stash_attention = stash
    tr.stash_mlp_rows = stash          # "recompute": the fully reversible walk (neither stash)
    out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
    torch.cuda.synchronize()
    # (the normalised s is O(1) per feature: |z| and the loss are large, the bound is relative to them)
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n + 1e-6 * abs(ref["total_loss"])
    np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
    names = []
    for (name, a), (_, b) in zip(_flat_attn(tr.named_gradients(), ws), _flat_attn(ref["grads"], ws)):
        names.append(name)
        tol = 5e-4 * float(np.abs(b).max()) + 1e-5
        err = float(np.abs(a - b).max())
        assert err <= tol, f"{name}: max err {err:.3e} > {tol:.3e} (max|g| {np.abs(b).max():.3e})"
    assert any(nm.endswith("ln_gamma") for nm in names) and any(nm.endswith("ln_beta") for nm in names)


def test_layer_norm_parameters_train(community_medium):
    """ln_gamma / ln_beta sit in the optimiser's arena: a few Adam steps move them and reduce the loss."""
    from gnf_amd.train import GRevNetTrainer
    attn = dict(num_heads=4, kq_dim=6, v_dim=5, out_dim=12, concat=True, kq_dim_division=False, residual=True,
                layer_norm=True)
    hp = dict(D=16, latent=48, K=3, T=2, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=False, attn=attn)
    nn, ne, s, r = _batch(community_medium, [3, 50, 77, 12, 100])
    n = int(nn.sum())
    x = (np.random.default_rng(0).standard_normal((n, 16)) * 2 + 1).astype(np.float32)
    net = make_product_grevnet(hp, None)          # Sonnet-style first-connect initialisation: gamma = 1, beta = 0
    tr = GRevNetTrainer(net, lr=2e-3, use_lr_decay=False)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    first = float(tr.step(graph)["total_loss"])
    blk = net.blocks("s")[0]
    np.testing.assert_array_less(0.0, np.abs(blk.attn_params["ln_gamma"].cpu().numpy() - 1.0).max())
    for _ in range(30):
        last = float(tr.step(graph)["total_loss"])
    assert last < first
    assert np.abs(blk.attn_params["ln_beta"].cpu().numpy()).max() > 0.0


@pytest.mark.parametrize("rows,dense", [(16, 0), (32, 0), (64, 0), (3264, 0), (32, 1), (64, 1)],
                         ids=["16", "32", "64", "32s64r", "32-dense-two-launches", "64-dense-two-launches"])
def test_attention_backward_row_tile_sizes(community_medium, rows, dense):
    """The attention backward's edge passes exist for 64-, 32- and 16-row tiles, k_attn_fwd_rows for 64 and 32 (the library
    picks by batch size and mean degree; the option attn_bwd_rows forces one; 32-row tiles split every row's
    edges over two lanes), and on sparse batches as ONE launch (k_attn_bwd_edges: sender tiles, which then carry the
    receivers' softmax statistics and delta in their LDS window, and receiver tiles; 3264 = 32-row sender / 64-row
    receiver tiles) or as two (what dense batches - here complete graphs - take; 16-row tiles always): the same loss and
    gradients through each.  attn_kernel=1 keeps the sparse batch on the rows kernels in the forward pass too."""
    from gnf_amd import _abi
    from gnf_amd.datasets import senders_receivers
    from gnf_amd.train import GRevNetTrainer
    attn = dict(num_heads=4, kq_dim=6, v_dim=5, out_dim=12, concat=True, kq_dim_division=True, residual=False)
    hp = dict(D=12, latent=32, K=2, T=2, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=False, attn=attn)
    if dense:
        nn = np.array([30, 41, 26, 64, 35], np.int32)
        s, r, ne = senders_receivers(nn)
        assert len(s) >= 24 * int(nn.sum())
    else:
        nn, ne, s, r = _batch(community_medium, [3, 50, 77, 12, 100])
    n = int(nn.sum())
    x = np.random.default_rng(8).standard_normal((n, 12)).astype(np.float32)
    p = O.make_attn_grevnet_params(13, 6, 32, 2, 2, final_scale=0.3, **attn)
    ref = O.loss_and_grads(s, r, n, x, p, 2, activation="relu")
    _abi.set_option("attn_bwd_rows", rows)
    _abi.set_option("attn_kernel", 1)
    try:
        tr = GRevNetTrainer(make_product_grevnet(hp, p))
        out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
        torch.cuda.synchronize()
    finally:
        _abi.set_option("attn_bwd_rows", 0)
        _abi.set_option("attn_kernel", 0)
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    for (name, a), (_, b) in zip(_flat_attn(tr.named_gradients(), False), _flat_attn(ref["grads"], False)):
        assert np.abs(a - b).max() <= 5e-4 * np.abs(b).max() + 1e-5, name


def test_attention_backward_window_wider_than_lds():
    """A sparse batch whose graphs are too large for the edge passes' LDS row window (one 1500-node graph with random
    long-range edges next to a small one): the tiles read their rows from global memory instead, and in the one-launch
    form the sender tiles then form delta = <dagg, attended> per edge themselves (nothing a receiver tile of the same
    launch writes may be read) - same gradients, both forms."""
    from gnf_amd import _abi
    from gnf_amd.train import GRevNetTrainer
    attn = dict(num_heads=4, kq_dim=6, v_dim=5, out_dim=12, concat=True, kq_dim_division=True, residual=False)
    hp = dict(D=8, latent=16, K=2, T=1, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=False, attn=attn)
    rng = np.random.default_rng(77)
    sizes = [1500, 9]
    s_l, r_l, ne, off = [], [], [], 0
    for m in sizes:
        a = rng.integers(0, m, size=3 * m)
        b = rng.integers(0, m, size=3 * m)
        pairs = {(int(i), int(i)) for i in range(m)}
        for u, v in zip(a, b):
            pairs.add((int(u), int(v)))
            pairs.add((int(v), int(u)))
        pairs = sorted(pairs)
        s_l.append(np.array([u for u, _ in pairs], np.int32) + off)
        r_l.append(np.array([v for _, v in pairs], np.int32) + off)
        ne.append(len(pairs))
        off += m
    nn, ne = np.array(sizes, np.int32), np.array(ne, np.int32)
    s, r = np.concatenate(s_l), np.concatenate(r_l)
    n = int(nn.sum())
    assert len(s) < 24 * n     # sparse: the one-launch form
    x = rng.standard_normal((n, 8)).astype(np.float32)
    p = O.make_attn_grevnet_params(5, 4, 16, 2, 1, final_scale=0.3, **attn)
    ref = O.loss_and_grads(s, r, n, x, p, 1, activation="relu")
    for rows in (0, 16):   # 0: the one-launch form; 16-row tiles: two launches
        _abi.set_option("attn_bwd_rows", rows)
        try:
            tr = GRevNetTrainer(make_product_grevnet(hp, p))
            out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
            torch.cuda.synchronize()
        finally:
            _abi.set_option("attn_bwd_rows", 0)
        assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
        for (name, a), (_, b) in zip(_flat_attn(tr.named_gradients(), False), _flat_attn(ref["grads"], False)):
            assert np.abs(a - b).max() <= 5e-4 * np.abs(b).max() + 1e-5, (rows, name)


def test_attention_backward_statistics_from_the_edge_tiled_forward():
    """The backward walk that RECOMPUTES the attention front-end (no stash) runs the two-launch forward kernels; on a
    sparse batch that is k_attn_agg, which takes a workgroup's edges 256 at a time - here 64 receiver rows of mean degree
    ~20 per workgroup, several edge tiles each (max pass first) - and leaves the softmax statistics and attended values
    the single-sweep backward kernels read."""
    from gnf_amd.train import GRevNetTrainer
    attn = dict(num_heads=2, kq_dim=5, v_dim=3, out_dim=7, concat=True, kq_dim_division=False, residual=False)
    hp = dict(D=6, latent=16, K=2, T=1, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=False, attn=attn)
    rng = np.random.default_rng(31)
    sizes = [260, 40]
    s_l, r_l, ne, off = [], [], [], 0
    for m in sizes:
        pairs = {(i, i) for i in range(m)}
        for u, v in zip(rng.integers(0, m, size=9 * m), rng.integers(0, m, size=9 * m)):
            pairs.add((int(u), int(v)))
            pairs.add((int(v), int(u)))
        pairs = sorted(pairs)
        s_l.append(np.array([u for u, _ in pairs], np.int32) + off)
        r_l.append(np.array([v for _, v in pairs], np.int32) + off)
        ne.append(len(pairs))
        off += m
    nn, ne = np.array(sizes, np.int32), np.array(ne, np.int32)
    s, r = np.concatenate(s_l), np.concatenate(r_l)
    n = int(nn.sum())
    assert 12 * n < len(s) < 24 * n     # sparse by the library's rule, yet > 256 edges per 64 rows
    x = rng.standard_normal((n, 6)).astype(np.float32)
    p = O.make_attn_grevnet_params(9, 3, 16, 2, 1, final_scale=0.3, **attn)
    ref = O.loss_and_grads(s, r, n, x, p, 1, activation="relu")
    tr = GRevNetTrainer(make_product_grevnet(hp, p))
    tr.stash_attention = False
    tr.stash_mlp_rows = False
    out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    for (name, a), (_, b) in zip(_flat_attn(tr.named_gradients(), False), _flat_attn(ref["grads"], False)):
        assert np.abs(a - b).max() <= 5e-4 * np.abs(b).max() + 1e-5, name


def test_attention_gradients_high_degree_rows():
    """Complete topology with a 70-node graph: rows with more than 64 in / out edges take the kernels' general
    (edge-tiled) path, the 9-node graph next to it the LDS-resident one."""
    from gnf_amd.datasets import senders_receivers
    from gnf_amd.train import GRevNetTrainer
    attn = dict(num_heads=2, kq_dim=3, v_dim=4, out_dim=6, concat=True, kq_dim_division=True, residual=False)
    hp = dict(D=6, latent=16, K=2, T=1, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=False, attn=attn)
    n_node = np.array([70, 9], np.int32)
    s, r, ne = senders_receivers(n_node)
    n = int(n_node.sum())
    x = np.random.default_rng(3).standard_normal((n, 6)).astype(np.float32)
    p = O.make_attn_grevnet_params(5, 3, 16, 2, 1, final_scale=0.3, **attn)
    ref = O.loss_and_grads(s, r, n, x, p, 1, activation="relu")
    tr = GRevNetTrainer(make_product_grevnet(hp, p))
    out = tr.loss_and_grads(graph_from_arrays(n_node, ne, s, r, x, DEV))
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    for (name, a), (_, b) in zip(_flat_attn(tr.named_gradients(), False), _flat_attn(ref["grads"], False)):
        assert np.abs(a - b).max() <= 5e-4 * np.abs(b).max() + 1e-5, name


def test_training_loop_with_the_default_gnn_and_batch_norm(community_medium):
    """The drivers' defaults together (attention GNN + use_batch_norm=True): a few iterations reduce the loss."""
    from gnf_amd.train import GRevNetTrainer
    attn = dict(num_heads=8, kq_dim=10, v_dim=10, out_dim=80, concat=True, kq_dim_division=False, residual=False)
    hp = dict(D=16, latent=64, K=3, T=2, agg="mean", combine="agg", epsilon=0.0, activation="relu",
              weight_sharing=False, attn=attn)
    nn, ne, s, r = _batch(community_medium, [3, 50, 77, 12, 100])
    n = int(nn.sum())
    x = (np.random.default_rng(0).standard_normal((n, 16)) * 2 + 1).astype(np.float32)
    p = O.make_attn_grevnet_params(17, 8, 64, 3, 2, final_scale=0.3, **attn)
    p["bn"] = O.make_bn_params(18, 8, 2)
    net = make_product_grevnet(hp, p)
    tr = GRevNetTrainer(net, lr=2e-3, use_lr_decay=False)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    ref = O.loss_and_grads(s, r, n, x, p, 2, activation="relu")
    first = tr.loss_and_grads(graph)
    torch.cuda.synchronize()
    assert abs(float(first["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    got = tr.named_gradients()
    for key in ("wq", "wo"):
        a, b = got["t"][1][0]["attn"][key], ref["grads"]["t"][1][0]["attn"][key]
        assert np.abs(a - b).max() <= 5e-4 * np.abs(b).max() + 1e-5
    np.testing.assert_allclose(got["bn"][0][1]["gamma"], ref["grads"]["bn"][0][1]["gamma"],
                               atol=5e-4 * np.abs(ref["grads"]["bn"][0][1]["gamma"]).max() + 1e-4)
    losses = [float(tr.step(graph)["loss_per_node"]) for _ in range(25)]
    assert losses[-1] < losses[0] - 0.05, losses


@pytest.mark.parametrize("ws", [False, True], ids=["", "weight_sharing"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "gemm"])
def test_gradients_with_batch_norm_vs_oracle(grid_small, fused, ws):
    """use_batch_norm=True (the drivers' default, run_grevnet.py:90): gradients of the MLP weights AND of every
    bijector's gamma / beta; the reversible walk also undoes the normalisation (reconstruction = x)."""
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=8, latent=32, K=3, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=ws)
    nn, ne, s, r = _batch(grid_small, list(range(12)))
    n = int(nn.sum())
    rng = np.random.default_rng(31)
    x = (rng.standard_normal((n, 8)) * 1.5 + 0.5).astype(np.float32)
    p = O.make_grevnet_params(8, 4, 32, 3, 2, weight_sharing=ws, final_scale=0.3)
    p["bn"] = O.make_bn_params(9, 4, 2)
    ref = O.loss_and_grads(s, r, n, x, p, 2, ws)
    net = make_product_grevnet(hp, p)
    net.fused = fused
    tr = GRevNetTrainer(net)
    out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
    got = tr.named_gradients()
    _check_grads(got, ref["grads"], ws)
    for half in range(2):
        for i in range(2):
            for key in ("gamma", "beta"):
                a, b = got["bn"][half][i][key], ref["grads"]["bn"][half][i][key]
                assert np.abs(a - b).max() <= 3e-4 * np.abs(b).max() + 1e-4, (half, i, key, a, b)


def test_training_with_batch_norm_updates_moving_statistics(grid_small):
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=8, latent=32, K=3, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    nn, ne, s, r = _batch(grid_small, list(range(12)))
    n = int(nn.sum())
    x = (np.random.default_rng(0).standard_normal((n, 8)) * 2 + 1).astype(np.float32)
    p = O.make_grevnet_params(4, 4, 32, 3, 2, final_scale=0.3)
    p["bn"] = O.make_bn_params(5, 4, 2)
    net = make_product_grevnet(hp, p)
    tr = GRevNetTrainer(net, lr=3e-3, use_lr_decay=False)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    mm0 = p["bn"][0][0]["moving_mean"].copy()
    losses = [float(tr.step(graph)["loss_per_node"]) for _ in range(25)]
    assert losses[-1] < losses[0] - 0.05, losses
    bn = net.bns[0][0]
    assert float(bn.gamma.min()) > 0.0                                       # the constraint projection
    # first bijector sees the raw data every step: its batch mean is the data mean, the moving mean crept towards it
    want = mm0 * 0.99 ** 25 + x[:, :4].mean(axis=0) * (1 - 0.99 ** 25)
    np.testing.assert_allclose(bn.moving_mean.cpu().numpy(), want, atol=1e-4)


def test_randomised_parity_sweep():
    """tools/fuzz_parity.py: random flow hyper-parameters (message passing / attention, batch norm, weight sharing,
    K = 1..4, odd widths) x random ragged batches (isolated nodes, duplicated / directed edges, complete graphs, a
    high-degree star) through forward, inverse and gradients on both kernel paths vs the float64 oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    results = [fz.run_case(i, 20260928) for i in range(24)]
    assert sum(r.startswith("ok") for r in results) >= 20, results


def test_gradients_on_a_batch_with_more_tiles_than_cus(community_medium):
    """> 256 sixteen-node tiles: the fused backward kernel switches to 32-node workgroups (and the forward kernel
    to its (2, 2) shape); gradients still match the oracle, with and without batch norm."""
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=16, latent=48, K=3, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    rng = np.random.default_rng(11)
    nn, ne, s, r = _batch(community_medium, rng.choice(168, size=110, replace=True))
    n = int(nn.sum())
    assert (n + 15) // 16 > 256
    x = rng.standard_normal((n, 16)).astype(np.float32)
    for use_bn in (False, True):
        p = O.make_grevnet_params(13, 8, 48, 3, 2, final_scale=0.3)
        if use_bn:
            p["bn"] = O.make_bn_params(14, 8, 2)
        ref = O.loss_and_grads(s, r, n, x, p, 2)
        tr = GRevNetTrainer(make_product_grevnet(hp, p))
        out = tr.loss_and_grads(graph_from_arrays(nn, ne, s, r, x, DEV))
        torch.cuda.synchronize()
        assert abs(float(out["loss_per_node"]) - ref["total_loss"] / n) <= 1e-4
        np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
        _check_grads(tr.named_gradients(), ref["grads"], False, scale=1e-3)


def test_thin_layers_behind_wide_ones_take_the_split_k_path(grid_small):
    """640-wide hidden layers on a ~100-node batch: the 640 -> H output layer (forward and recompute) and the first
    layer's dX GEMM have a handful of tiles and a long reduction, which the generic path splits over the reduction
    (launch_gemm split-K + k_splitk_epilogue; the layered forward borrows its free ping-pong buffer for the slabs).
    Forward, inverse and every gradient against the oracle, with and without batch norm."""
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=24, latent=640, K=3, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    nn, ne, s, r = _batch(grid_small, list(range(10)))
    n = int(nn.sum())
    rng = np.random.default_rng(17)
    x = (rng.standard_normal((n, 24)) * 0.8).astype(np.float32)
    for use_bn in (False, True):
        p = O.make_grevnet_params(21, 12, 640, 3, 2, final_scale=0.3)
        if use_bn:
            p["bn"] = O.make_bn_params(22, 12, 2)
        ref = O.loss_and_grads(s, r, n, x, p, 2)
        net = make_product_grevnet(hp, p)
        net.fused = False                         # layered forward + generic (GEMM) backward
        graph = graph_from_arrays(nn, ne, s, r, x, DEV)
        tr = GRevNetTrainer(net)
        out = tr.loss_and_grads(graph)
        torch.cuda.synchronize()
        assert abs(float(out["loss_per_node"]) - ref["total_loss"] / n) <= 1e-4
        np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
        _check_grads(tr.named_gradients(), ref["grads"], False, scale=1e-3)
        if not use_bn:                            # g(f(x)) = x through the layered inverse (moving stats aside)
            z, _ = net(graph, inverse=True)
            back = net(z, inverse=False)
            np.testing.assert_allclose(back.nodes.cpu().numpy(), x, atol=3e-4, rtol=3e-4)


DW_MODES = [   # (gnf_set_option values, dw_modes_check.py flags)
    ({}, ""),                                                     # what the library picks by itself (this batch: the merged launch)
    ({}, "ws"),                                                   # ... with weight sharing: the in-launch reduce accumulates
    ({"dw_wide_units": 8}, ""),                                   # merged launch, few dW workgroups: cheap units ride behind, strided
    ({"dw_wide_units": 13}, "ws"),                                # ... stream-K with an odd workgroup count
    ({}, "ws,nostash"),                                           # merged launch recomputing the MLP rows (no stash)
    ({}, "big"),                                                  # more than 192 tiles: dW GEMMs on the auxiliary stream
    ({"dw_grouped": 1}, ""),                                      # the 128 x 64 grouped kernel
    ({"dw_wide_units": 8}, "big"),
    ({"dw_wide_units": 200}, "ws,big"),                           # many node chunks per job + accumulating reduce
    ({"dw_wide_units": 40}, "ws,serial,big"),                     # no auxiliary stream
    ({"dw_wide_units": 13}, "ws,big"),
    ({"bwd_generic": 1}, ""),                                     # generic (GEMM) backward
    ({"bwd_generic": 1}, "ws"),
]


def _mode_id(m):
    return ("-".join(f"{k}{v}" for k, v in m[0].items()) or "auto") + ("_" + m[1].replace(",", "_") if m[1] else "")


@pytest.mark.parametrize("env,arg", DW_MODES, ids=[_mode_id(m) for m in DW_MODES])
def test_weight_gradient_kernel_launch_shapes(env, arg):
    """The dW GEMM has several launch shapes chosen per batch; each forced shape (developer
    options of the library, gnf_set_option, handed to the child through the binding's GNF_OPTIONS variable) runs
    tools/dw_modes_check.py (gradients of a ~900-node batch with ragged layer widths vs the oracle, 1e-3 of each
    tensor's max) in its own interpreter so that no option leaks into the other tests."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e["GNF_OPTIONS"] = ",".join(f"{k}={v}" for k, v in env.items())
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dw_modes_check.py")] + ([arg] if arg else []),
                       env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "dw-modes-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_cross_rank_batch_norm_moments_two_ranks_one_gpu():
    """GRevNet(sync_batch_norm=True) / GnfFlow.bn_allreduce (ABI v5): two ranks on this GPU (gloo), half of the graphs
    each, reproduce the single-process whole-batch z (bitwise in practice), loss, batch moments and - after the
    gradient all-reduce - gradients; per-shard moments do not (tools/sync_bn_check.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sync_bn_check.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "sync-bn-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_batch_norm_allreduce_hook_contract(grid_small):
    """The C-ABI side of the hook on one rank: a missing bn_sync_buf and a failing hook are reported, an identity
    hook (one rank: the sums are already the whole batch's) gives exactly the result of the hook-less path."""
    import ctypes as C
    from gnf_amd import _abi
    hp = dict(D=8, latent=32, K=2, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    nn, ne, s, r = _batch(grid_small, list(range(8)))
    n = int(nn.sum())
    x = (np.random.default_rng(2).standard_normal((n, 8)) * 1.2).astype(np.float32)
    p = O.make_grevnet_params(8, 4, 32, 2, 2, final_scale=0.3)
    p["bn"] = O.make_bn_params(9, 4, 2)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    net = make_product_grevnet(hp, p)
    z_ref, ld_ref = net(graph, inverse=True)
    calls = []
    rc = [0]

    def hook(ctx, buf, count, stream):
        calls.append(int(count))
        return rc[0]
    cb = _abi.BN_ALLREDUCE_FN(hook)
    flow = net._flow(4, torch.device(DEV))          # the cached descriptor the next calls use
    flow.bn_allreduce = cb
    with pytest.raises(_abi.GnfError, match="bn_sync_buf"):
        net(graph, inverse=True)
    buf = torch.zeros(2 * 4 + 1, dtype=torch.float64, device=DEV)
    flow.bn_sync_buf = buf.data_ptr()
    z, ld = net(graph, inverse=True)
    torch.cuda.synchronize()
    assert calls == [9] * 4                          # one exchange per bijector call: 2 T half-steps
    assert torch.equal(z.nodes, z_ref.nodes) and float(ld) == float(ld_ref)
    assert float(buf[8]) == n                        # [sum, sum of squares] x 4 features, then the node count
    rc[0] = -7
    with pytest.raises(_abi.GnfError, match="hook failed"):
        net(graph, inverse=True)
    flow.bn_allreduce = _abi.BN_ALLREDUCE_FN()       # back to NULL
    flow.bn_sync_buf = None


def test_set_params_after_training_started_reseats_the_arena(grid_small):
    """parameters replaced with set_params() after the first step live outside the trainer's flat arena;
    the trainer must notice and re-seat the arena on them (Adam would otherwise keep updating a dead copy)."""
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=8, latent=16, K=2, T=2, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    nn, ne, s, r = _batch(grid_small, list(range(6)))
    n = int(nn.sum())
    x = np.random.default_rng(0).standard_normal((n, 8)).astype(np.float32)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    p1 = O.make_grevnet_params(1, 4, 16, 2, 2, final_scale=0.3)
    p2 = O.make_grevnet_params(2, 4, 16, 2, 2, final_scale=0.3)
    net = make_product_grevnet(hp, p1)
    tr = GRevNetTrainer(net, lr=1e-3, use_lr_decay=False)
    tr.step(graph)
    net.set_params(p2)                                  # new tensors, outside tr.theta
    out = tr.loss_and_grads(graph)
    ref = O.loss_and_grads(s, r, n, x, p2, 2)
    assert abs(float(out["total_loss"]) - ref["total_loss"]) / n <= 1e-4      # the forward reads p2 ...
    w_first = net.mlps("s")[0].params[0][0]
    lo = tr.theta.data_ptr()
    assert lo <= w_first.data_ptr() < lo + 4 * tr.theta.numel()               # ... and p2 now lives inside the arena
    before = w_first.clone()
    tr.apply_gradients()
    torch.cuda.synchronize()
    assert float((w_first - before).abs().max()) > 0                         # Adam moves the tensors the kernels read


def test_pred_adj_rejects_a_launch_bound_below_the_largest_graph():
    from gnf_amd.flow import pred_adj
    z = np.zeros((12, 4), np.float32)
    g = graph_from_arrays([5, 7], [0, 0], np.zeros(0, np.int32), np.zeros(0, np.int32), z, DEV)
    with pytest.raises(ValueError, match="below the largest graph"):
        pred_adj(g, max_nodes_per_graph=6)


def test_wide_nets_packed_copies_follow_the_weights(community_medium):
    """Nets too wide for the fused kernels (1280-wide hidden layers) on a batch large enough for k_linear_big: that kernel
    reads the PACKED copy of the middle layer, the backward pass and Adam the raw W.  After optimiser steps the forward
    through the packed copy must equal the forward through the raw weights (net.fused = False: generic tile), and a step
    taken from the updated weights must see them (the loss of step 3 is the loss the raw weights give)."""
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=24, latent=1280, K=3, T=1, agg="mean", combine="agg", epsilon=1.0, activation="relu", weight_sharing=False)
    nn, ne, s, r = _batch(community_medium, list(range(64)))
    n = int(nn.sum())
    assert n >= 1650                               # (launch_linear_big wants two row-tile units per CU slot)
    x = (np.random.default_rng(5).standard_normal((n, 24)) * 0.7).astype(np.float32)
    p = O.make_grevnet_params(31, 12, 1280, 3, 1, final_scale=0.3)
    net = make_product_grevnet(hp, p)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    before = float(log_prob_terms(net, graph)["log_prob_xs_per_node"])
    tr = GRevNetTrainer(net, lr=2e-3, use_lr_decay=False)
    for _ in range(2):
        tr.step(graph)
    packed = log_prob_terms(net, graph)
    z_packed = packed["z_graph"].nodes.cpu().numpy()
    lp_packed = float(packed["log_prob_xs_per_node"])
    assert abs(lp_packed - before) > 1e-2, "the two steps did not move the model: the test would prove nothing"
    loss3 = float(tr.step(graph)["loss_per_node"])            # forward of step 3 = the forward after two updates
    assert abs(loss3 + lp_packed) <= 1e-5 * max(1.0, abs(lp_packed))
    # ... and the same weights through the raw-W kernels: restore step 2's state first
    net2 = make_product_grevnet(hp, p)
    tr2 = GRevNetTrainer(net2, lr=2e-3, use_lr_decay=False)
    for _ in range(2):
        tr2.step(graph)
    net2.fused = False
    raw = log_prob_terms(net2, graph)
    assert abs(float(raw["log_prob_xs_per_node"]) - lp_packed) <= 2e-5 * max(1.0, abs(lp_packed))
    np.testing.assert_allclose(raw["z_graph"].nodes.cpu().numpy(), z_packed, atol=2e-4, rtol=2e-4)


def test_packed_bias_rows_of_every_layer_follow_an_optimiser_step(community_medium):
    """gnf_pack_flow (include/gnf.h: "every bias row"): a wide net's last layer 1280 -> 150 (H > 128: not the wide kernel's
    fused last layer, no short first layer, no wide product in either direction) has NO reader of its fragment copies, so
    round 5 emitted no descriptor for it - and its padded bias row in `packed` kept the initial values for ever.
    After two optimiser steps every layer's bias row in `packed` must equal the layer's current bias."""
    from gnf_amd.train import GRevNetTrainer
    d, latent = 300, 1280
    hp = dict(D=d, latent=latent, K=3, T=1, agg="mean", combine="agg", epsilon=1.0, activation="relu", weight_sharing=False)
    nn, ne, s, r = _batch(community_medium, list(range(64)))
    n = int(nn.sum())
    x = (np.random.default_rng(6).standard_normal((n, d)) * 0.7).astype(np.float32)
    net = make_product_grevnet(hp, O.make_grevnet_params(33, d // 2, latent, 3, 1, final_scale=0.3))
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    tr = GRevNetTrainer(net, lr=2e-3, use_lr_decay=False)
    b_before = [b.detach().cpu().clone() for (_w, b) in net.mlps("s")[0].params]
    for _ in range(2):
        tr.step(graph)
    torch.cuda.synchronize()
    packed = net._cache[2][3]                                   # all nets' packed copies, s-nets first
    pad = lambda v: (v + 15) // 16 * 16                         # noqa: E731
    dims = [d // 2, latent, latent, d // 2]
    woff = sum(pad(dims[j]) * pad(dims[j + 1]) for j in range(3))
    off = woff                                                  # net 0 (s, half 0, step 0): bias rows behind its Wp
    for j, (_w, b) in enumerate(net.mlps("s")[0].params):
        row = packed[off:off + pad(dims[j + 1])]
        assert float((b.cpu() - b_before[j]).abs().max()) > 0, "the steps did not move this bias: the test would prove nothing"
        assert torch.equal(row[:dims[j + 1]], b), f"layer {j}: packed bias row is stale"
        assert float(row[dims[j + 1]:].abs().max()) == 0.0 if pad(dims[j + 1]) > dims[j + 1] else True
        off += pad(dims[j + 1])


def test_mlp_row_stash_beyond_its_budget_trains_through_the_recomputing_walk(community_medium):
    """gnf_mlp_stash_bytes offers the layered stash at any size up to 48 GB and the trainer allocated it
    unconditionally - a batch that trained through the recomputing walk could now die of a device OOM.  The trainer takes the
    stash only within its budget (a cap, or half of the free device memory): with a cap below the stash's size the step runs
    without it (mlp_stash = NULL), says so, and produces the gradients of the recomputing walk bit for bit."""
    from gnf_amd.train import GRevNetTrainer
    hp = dict(D=24, latent=1280, K=3, T=1, agg="mean", combine="agg", epsilon=1.0, activation="relu", weight_sharing=False)
    nn, ne, s, r = _batch(community_medium, list(range(64)))
    n = int(nn.sum())
    x = (np.random.default_rng(5).standard_normal((n, 24)) * 0.7).astype(np.float32)
    p = O.make_grevnet_params(31, 12, 1280, 3, 1, final_scale=0.3)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    grads = {}
    for mode in ("capped", "recompute", "stash"):
        net = make_product_grevnet(hp, p)
        tr = GRevNetTrainer(net, lr=1e-3, use_lr_decay=False)
        if mode == "capped":
            tr.mlp_stash_max_bytes = 1 << 20                    # the stash of this batch is ~100 MB
        if mode == "recompute":
            tr.stash_mlp_rows = False
        tr.loss_and_grads(graph)
        torch.cuda.synchronize()
        grads[mode] = tr.grad.clone()
        if mode == "capped":
            assert tr._mlp_stash is None and tr.mlp_stash_declined is not None
            assert tr.mlp_stash_declined[0] > (1 << 20) and "mlp_stash_max_bytes" in tr.mlp_stash_declined[1]
        if mode == "stash":
            assert tr._mlp_stash is not None and tr.mlp_stash_declined is None
    assert torch.equal(grads["capped"], grads["recompute"])
    assert float((grads["stash"] - grads["recompute"]).abs().max()) <= 2e-3 * float(grads["recompute"].abs().max())


@pytest.mark.parametrize("gnn_kind", ["avg_then_mlp", "dm_attn_bn"])
def test_layered_mlp_row_stash_matches_the_recomputing_walk(community_medium, gnn_kind):
    """Nets too wide for the fused kernels (layered forward, generic backward): with GnfFlow.mlp_stash the forward's hidden
    activations and s, t go straight into the stash and the backward pass skips its recompute of both MLPs
    (layered_stash_mode).  The library offers the stash for such a flow, the gradients agree with the fully reversible walk to
    rounding (the recompute starts from the RECONSTRUCTED inputs, the stash holds the forward's own rows) and with the
    oracle, and a plain forward afterwards does not touch the stash."""
    import ctypes as C
    from gnf_amd import _abi
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.train import GRevNetTrainer
    d, latent, k, t = 24, 1280, 3, 2
    attn = dict(num_heads=1, kq_dim=64, v_dim=64, out_dim=64, concat=True, kq_dim_division=True, residual=False) if gnn_kind == "dm_attn_bn" else None
    hp = dict(D=d, latent=latent, K=k, T=t, agg="mean", combine="agg", epsilon=1.0, activation="relu", weight_sharing=False)
    if attn:
        hp["attn"] = attn
    nn, ne, s, r = _batch(community_medium, list(range(64)))
    n = int(nn.sum())
    x = (np.random.default_rng(8).standard_normal((n, d)) * 0.7).astype(np.float32)
    if attn:
        p = O.make_attn_grevnet_params(41, d // 2, latent, k, t, final_scale=0.3, **attn)
        p["bn"] = O.make_bn_params(42, d // 2, t)
    else:
        p = O.make_grevnet_params(41, d // 2, latent, k, t, final_scale=0.3)
    ref = O.loss_and_grads(s, r, n, x, p, t, activation="relu")
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    got = {}
    for stash in (True, False):
        net = make_product_grevnet(hp, p)
        tr = GRevNetTrainer(net)
        tr.stash_mlp_rows = stash
        out = tr.loss_and_grads(graph)
        torch.cuda.synchronize()
        assert (tr._mlp_stash is not None) == stash
        if stash:
            flow = net._flow(d // 2, torch.device(DEV))
            in0 = d // 2 + (64 if attn else 0)
            want = 2 * t * 4 * sum((v + 63) // 64 * 64 for v in (n * in0, n * latent, n * latent, n * latent, n * latent, n * (d // 2), n * (d // 2)))
            assert _abi.lib().gnf_mlp_stash_bytes(n, d, C.byref(flow)) >= want   # (+ the fused mode's ballot words)
        assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
        np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
        got[stash] = tr.named_gradients()
        before = tr._mlp_stash.clone() if stash else None
        log_prob_terms(net, graph)                  # a plain forward: flow.mlp_stash is NULL again
        torch.cuda.synchronize()
        if stash:
            assert torch.equal(before, tr._mlp_stash)

    def flat(gr):
        for kind in ("s", "t"):
            for net_ in gr[kind][0] + gr[kind][1]:
                mlp = net_["mlp"] if isinstance(net_, dict) else net_
                if isinstance(net_, dict):
                    for key in ("wq", "wk", "wv", "wo"):
                        yield net_["attn"][key]
                for (w, b) in mlp:
                    yield w
                    yield b
    gmax = max(float(np.abs(b).max()) for b in flat(ref["grads"]))
    for a, b, c in zip(flat(got[True]), flat(got[False]), flat(ref["grads"])):
        scale = max(float(np.abs(c).max()), 1e-3 * gmax)
        assert float(np.abs(a - b).max()) <= 2e-3 * scale          # stash vs recompute: relu kinks of the reconstruction aside
        assert float(np.linalg.norm(a - c)) <= 2e-3 * max(float(np.linalg.norm(c)), 1e-3 * gmax * np.sqrt(c.size))


@pytest.mark.parametrize("d,latent,act,combine", [(14, 1280, "leaky_relu", "agg"), (200, 1040, "relu", "agg"), (256, 1280, "relu", "agg"),
                                                  (200, 1280, "leaky_relu", "concat")],
                         ids=["H7_leaky", "H100_ragged_hidden", "H128_widest", "H100_concat_in200"])
def test_wide_layer_takes_the_thin_last_layer_along(community_medium, d, latent, act, combine):
    """Layered forward of nets too wide for the fused kernels: the wide layer in front of the thin last one multiplies it out
    of its accumulators (k_linear_big's second epilogue, launch_linear_big_fused) and the coupling kernel adds the column
    blocks' partial products and the bias.  Output widths that are no multiple of 4 or 16, a hidden width that is no
    multiple of 256 (a column block with dead waves) and the widest last layer it takes (128): forward against the oracle,
    the inverse through the same kernels, every gradient (stash mode: the coupling kernel also writes s, t into the slot).
    The first layer of these nets is the short-reduction kernel's (k_linear_short: 100 -> 1040 with a column block of dead waves,
    128 -> 1280 = 8 k-groups exactly, 200 -> 1280 = the widest reduction it holds in registers; 7 -> 1280 is not whole float4s
    and stays on the generic tile)."""
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.train import GRevNetTrainer
    t = 2
    hp = dict(D=d, latent=latent, K=3, T=t, agg="mean", combine=combine, epsilon=0.0 if combine == "concat" else 1.0, activation=act,
              weight_sharing=False)
    nn, ne, s, r = _batch(community_medium, list(range(64)))
    n = int(nn.sum())
    x = (np.random.default_rng(9).standard_normal((n, d)) * 0.7).astype(np.float32)
    p = O.make_grevnet_params(51, d // 2, latent, 3, t, combine=combine, final_scale=0.3)
    ref = O.loss_and_grads(s, r, n, x, p, t, activation=act, combine=combine, epsilon=hp["epsilon"])
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    net = make_product_grevnet(hp, p)
    terms = log_prob_terms(net, graph)
    assert abs(float(terms["log_prob_xs_per_node"]) + ref["total_loss"] / n) <= 1e-4 * max(1.0, abs(ref["total_loss"] / n))
    z = terms["z_graph"]
    back = net(z, inverse=False)
    np.testing.assert_allclose(back.nodes.cpu().numpy(), x, atol=3e-4, rtol=3e-4)
    for stash in (True, False):
        tr = GRevNetTrainer(make_product_grevnet(hp, p))
        tr.stash_mlp_rows = stash
        out = tr.loss_and_grads(graph)
        torch.cuda.synchronize()
        assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
        gmax = max(float(np.abs(b).max()) for _, b in _flat(ref["grads"], False))
        for (name, a), (_, c) in zip(_flat(tr.named_gradients(), False), _flat(ref["grads"], False)):
            # (2-norm per tensor: single elements next to a relu kink of the reconstruction may land on the other side)
            bound = 2e-3 * max(float(np.linalg.norm(c)), 1e-3 * gmax * np.sqrt(c.size))
            assert float(np.linalg.norm(a - c)) <= bound, name


def test_checkpoint_restore_of_a_wide_net_repacks_what_the_wide_kernels_read(community_medium, tmp_path):
    """examples/driver_utils.py save_checkpoint / load_checkpoint (the drivers' tf.train.Saver, run_grevnet.py:379,449-453) on a
    net too wide for the fused kernels: after the restore the forward through the packed middle layer (k_linear_big) equals
    the forward of the trainer that wrote the checkpoint, and the forward through the raw weights (gnf_pack_flow
    used to skip such nets, a restored model ran on its INITIAL middle layer)."""
    import os
    import sys
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.train import GRevNetTrainer
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import driver_utils as DU
    hp = dict(D=24, latent=1280, K=3, T=1, agg="mean", combine="agg", epsilon=1.0, activation="relu", weight_sharing=False)
    nn, ne, s, r = _batch(community_medium, list(range(64)))
    n = int(nn.sum())
    x = (np.random.default_rng(6).standard_normal((n, 24)) * 0.7).astype(np.float32)
    p = O.make_grevnet_params(51, 12, 1280, 3, 1, final_scale=0.3)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    net = make_product_grevnet(hp, p)
    tr = GRevNetTrainer(net, lr=2e-3, use_lr_decay=False)
    for _ in range(3):
        tr.step(graph)
    want = float(log_prob_terms(net, graph)["log_prob_xs_per_node"])
    path = str(tmp_path / "ckpt.pt")
    DU.save_checkpoint(tr, path)
    net2 = make_product_grevnet(hp, p)                  # a fresh model with the INITIAL weights
    tr2 = GRevNetTrainer(net2, lr=2e-3, use_lr_decay=False)
    tr2.loss_and_grads(graph)                           # connects the variables (the restore needs them to exist)
    first = float(log_prob_terms(net2, graph)["log_prob_xs_per_node"])
    assert abs(first - want) > 1e-2
    DU.load_checkpoint(tr2, path)
    assert tr2.global_step == 3
    got = float(log_prob_terms(net2, graph)["log_prob_xs_per_node"])
    assert abs(got - want) <= 1e-6 * max(1.0, abs(want))
    net2.fused = False
    raw = float(log_prob_terms(net2, graph)["log_prob_xs_per_node"])
    assert abs(raw - want) <= 2e-5 * max(1.0, abs(want))
