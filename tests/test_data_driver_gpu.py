"""The DATA driver's literal defaults on the GPU (/root/reference/train_grevnet_with_data.py:40-46, 100-117, 272-310,
336-355): dm_self_attn_gnn(kq_dim=64, v_dim=64, num_heads=1, concat_heads_output_dim=64, kq_dim_division=True) around a
relu MLP, D = 200, use_batch_norm=True, complete graphs from transform_example - forward, inverse and every gradient tensor
against the fp64 oracle: once at a hidden width the oracle does in seconds, once at the literal 2048 x 3 (T reduced; the
widths are not), through the fused-where-possible and the layered path; a golden fixture of the head geometry; and one
kq / v value on either side of every boundary the attention dispatch has (10 | 16 | 32 | 33 | 64 | 65, heads 1 / 2 / 9).

Tolerances: the per-node log-prob bar (1e-4) is absolute; z and g(z) 3e-4; a gradient tensor is compared with
atol = 5e-4 * max|g| of that tensor (+ 1e-5 + 1e-6 of the flow's gradient scale); the literal-width case derives its bound
from the same autograd run in float32 on the CPU (relu kinks: see the test)."""
import numpy as np
import pytest
import torch

from helpers import DATA_DRIVER_GOLDEN_CASES, graph_from_arrays, load_golden, make_product_grevnet
from oracle import gnf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

DATA_ATTN = dict(num_heads=1, kq_dim=64, v_dim=64, out_dim=64, concat=True, kq_dim_division=True, residual=False)


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from gnf_amd import _abi
    _abi.lib()


def _complete_batch(n_node):
    """transform_example's topology (train_grevnet_with_data.py:237-244; utils.py:164-183) through the product's own
    host function, as numpy edge lists for the oracle."""
    from gnf_amd.datasets import senders_receivers
    s, r, ne = senders_receivers(np.asarray(n_node, np.int32))
    return np.asarray(n_node, np.int32), ne.astype(np.int32), s.astype(np.int32), r.astype(np.int32)


def _flat_all(grads):
    for kind in ("s", "t"):
        for q, net in enumerate(grads[kind][0] + grads[kind][1]):
            for key in ("wq", "wk", "wv", "wo"):
                yield f"{kind}[{q}].{key}", net["attn"][key]
            for j, (w, b) in enumerate(net["mlp"]):
                yield f"{kind}[{q}].W{j}", w
                yield f"{kind}[{q}].b{j}", b
    if "bn" in grads:
        for half in range(2):
            for i, b in enumerate(grads["bn"][half]):
                yield f"bn[{half}][{i}].gamma", b["gamma"]
                yield f"bn[{half}][{i}].beta", b["beta"]


def _check_all_grads(got, ref, scale):
    gmax = max(float(np.abs(b).max()) for _, b in _flat_all(ref))
    worst = 0.0
    for (name, a), (_, b) in zip(_flat_all(got), _flat_all(ref)):
        tol = scale * float(np.abs(b).max()) + 1e-5 + 1e-6 * gmax
        err = float(np.abs(np.asarray(a) - np.asarray(b)).max())
        worst = max(worst, err / max(float(np.abs(b).max()), 1e-30))
        assert err <= tol, f"{name}: max err {err:.3e} > {tol:.3e} (max|g| {np.abs(b).max():.3e})"
    return worst


def _grad_errors(got, ref):
    """Per tensor: max |a - b| / max |b| and ||a - b||_2 / ||b||_2 (tensors that are zero by cancellation judged against the
    flow's overall gradient scale)."""
    gmax = max(float(np.abs(b).max()) for _, b in _flat_all(ref))
    out = []
    for (name, a), (_, b) in zip(_flat_all(got), _flat_all(ref)):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        scale = max(float(np.abs(b).max()), 1e-3 * gmax)
        l2 = max(float(np.linalg.norm(b)), 1e-3 * gmax * np.sqrt(b.size))
        out.append((name, float(np.abs(a - b).max()) / scale, float(np.linalg.norm(a - b)) / l2))
    return out


def _hp(d, latent, k, t, attn, ws=False):
    return dict(D=d, latent=latent, K=k, T=t, agg="mean", combine="agg", epsilon=0.0, activation="relu",
                weight_sharing=ws, attn=attn)


def _forward_inverse_vs_oracle(net, nn, ne, s, r, x, p, t, rng, lp_tol=1e-4):
    from gnf_amd.flow import log_prob_terms
    n, d = x.shape
    o = O.Fp64Dense(s, r, n, activation="relu")
    ref = o.log_prob(x, p, t)
    graph = graph_from_arrays(nn, ne, s, r, x, DEV)
    out = log_prob_terms(net, graph)
    torch.cuda.synchronize()
    assert abs(float(out["log_prob_xs_per_node"]) - ref["log_prob_xs_per_node"]) <= lp_tol
    np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), ref["z"], atol=3e-4, rtol=3e-4)
    zs = rng.standard_normal((n, d)).astype(np.float32)
    xg = net(graph.replace(nodes=torch.as_tensor(zs).to(DEV)), inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(xg, o.g(zs, p, t), atol=3e-4, rtol=3e-4)      # (moving statistics in g: gnn.py:356-358)
    return graph, ref


@pytest.mark.parametrize("stash", [True, False], ids=["stash", "recompute"])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
def test_data_driver_defaults_at_a_width_the_oracle_does_in_seconds(grid_small, fused, stash):
    """D = 200 (H = 100), dm_attn 1 x 64 / 64 / C 64 with kq_dim_division, relu MLP 96 x 3, batch norm, T = 2, complete
    graphs: forward + inverse + every gradient tensor (attention, MLP, bijectors) vs fp64."""
    from gnf_amd.train import GRevNetTrainer
    d, latent, k, t = 200, 96, 3, 2
    nn, ne, s, r = _complete_batch(grid_small[0][[6, 0, 7, 2, 9, 4, 11]])
    n = int(nn.sum())
    rng = np.random.default_rng(64)
    x = (rng.standard_normal((n, d)) * 0.9 + 0.1).astype(np.float32)
    p = O.make_attn_grevnet_params(65, d // 2, latent, k, t, final_scale=0.3, **DATA_ATTN)
    p["bn"] = O.make_bn_params(66, d // 2, t)
    net = make_product_grevnet(_hp(d, latent, k, t, DATA_ATTN), p)
    net.fused = fused
    graph, _ = _forward_inverse_vs_oracle(net, nn, ne, s, r, x, p, t, rng)
    ref = O.loss_and_grads(s, r, n, x, p, t, activation="relu")
    tr = GRevNetTrainer(net)
    tr.stash_attention = tr.stash_mlp_rows = stash
    out = tr.loss_and_grads(graph)
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
    _check_all_grads(tr.named_gradients(), ref["grads"], 5e-4)


@pytest.mark.parametrize("fused", [True, False], ids=["packed", "raw_weights"])
def test_data_driver_defaults_at_the_literal_widths(community_medium, fused):
    """relu MLP 2048 x 3 (train_grevnet_with_data.py:113-114), D = 200, dm_attn 1 x 64 / 64 / 64, batch norm, complete
    graphs; T = 1 (the widths are the literal ones, the depth is not).  40 community_medium graphs (~1 650 nodes: enough
    row tiles for the wide layers to take k_linear_big) for forward / inverse, and every gradient tensor vs fp64
    autograd on the same batch."""
    from gnf_amd.train import GRevNetTrainer
    d, latent, k, t = 200, 2048, 3, 1
    ids = np.random.default_rng(12345).choice(168, size=40, replace=True)
    nn, ne, s, r = _complete_batch(community_medium[0][ids])
    n = int(nn.sum())
    assert n >= 1100
    rng = np.random.default_rng(2048)
    x = (rng.standard_normal((n, d)) * 0.8).astype(np.float32)
    p = O.make_attn_grevnet_params(2049, d // 2, latent, k, t, final_scale=0.25, **DATA_ATTN)
    p["bn"] = O.make_bn_params(2050, d // 2, t)
    net = make_product_grevnet(_hp(d, latent, k, t, DATA_ATTN), p)
    net.fused = fused
    graph, _ = _forward_inverse_vs_oracle(net, nn, ne, s, r, x, p, t, rng)
    ref = O.loss_and_grads(s, r, n, x, p, t, activation="relu")
    tr = GRevNetTrainer(net)
    out = tr.loss_and_grads(graph)
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    np.testing.assert_allclose(out["reconstruction"].cpu().numpy(), x, atol=3e-4, rtol=3e-4)
    # 3.5 M relu pre-activations per net: some lie within fp32 rounding of the kink and take the other side than in float64,
    # which moves a whole term of a gradient column (tests/test_fullsize_gpu.py states the same for the run_grevnet.py
    # defaults).  The bound is derived: the SAME autograd in float32 on the CPU shows what single precision costs on these
    # inputs; the device has to stay within SLACK of that run's worst tensor in the maximum norm and in the 2-norm, and
    # inside 2e-3 in the 2-norm whatever the CPU run does.
    r32 = O.loss_and_grads(s, r, n, x, p, t, activation="relu", dtype=torch.float32)
    e32 = _grad_errors(r32["grads"], ref["grads"])
    errs = _grad_errors(tr.named_gradients(), ref["grads"])
    worst_max, worst_l2 = max(errs, key=lambda e: e[1]), max(errs, key=lambda e: e[2])
    print(f"literal widths: worst tensor {worst_max[0]} {worst_max[1]:.2e} (max norm), {worst_l2[0]} {worst_l2[2]:.2e} (2-norm); "
          f"float32 CPU autograd {max(e[1] for e in e32):.2e} / {max(e[2] for e in e32):.2e}")
    assert worst_max[1] <= 6.0 * max(e[1] for e in e32) + 1e-4, (worst_max, max(e[1] for e in e32))
    assert worst_l2[2] <= min(6.0 * max(e[2] for e in e32) + 1e-5, 2e-3), (worst_l2, max(e[2] for e in e32))


# one kq / v value on either side of every boundary of the attention dispatch: the per-(row, head) thread kernels' register
# instances (<= 10, <= 32), the matrix-core / generic kernels beyond (33 .. 64, 65 ..), heads <= 8 / > 8, the ABI's limit
# heads * kq <= 256
BOUNDARY_SHAPES = [
    # heads, kq, v, C, D
    (1, 10, 10, 12, 24), (1, 16, 16, 16, 24), (1, 32, 32, 24, 24), (1, 33, 32, 24, 24), (1, 32, 33, 24, 24),
    (1, 64, 64, 64, 24), (1, 65, 64, 32, 24), (1, 64, 65, 32, 24), (2, 64, 16, 40, 24), (2, 33, 64, 40, 24),
    (9, 8, 8, 30, 24), (4, 64, 64, 48, 24), (1, 256, 256, 64, 24), (3, 40, 70, 50, 16),
    # the limit's last clause, pad16(2 heads kq + v) + pad16(H) + H <= 1272 (sixteen rows of the backward pass's dL/dx_cond
    # product in one CU's LDS): a wide feature half at the data driver's head (found by tools/fuzz_parity.py --wide: the
    # backward pass used to reject H = 200 at kq = v = 64 while the forward ran it), and the clause's edge for two geometries
    (1, 64, 64, 16, 400), (4, 64, 64, 48, 688), (1, 256, 256, 64, 496),
]


@pytest.mark.parametrize("dense", [True, False], ids=["complete_graphs", "dataset_topology"])
@pytest.mark.parametrize("shape", BOUNDARY_SHAPES, ids=[f"h{s[0]}_kq{s[1]}_v{s[2]}_C{s[3]}" + (f"_D{s[4]}" if s[4] > 24 else "") for s in BOUNDARY_SHAPES])
def test_attention_head_geometry_boundaries(community_medium, shape, dense):
    """Forward, inverse and gradients for head geometries around every dispatch boundary, on complete graphs (mean
    degree ~40: the rows / matrix-core kernels) and on the dataset's sparse topology (the edge-tiled / front-end kernels)."""
    from gnf_amd.train import GRevNetTrainer
    nh, kq, vd, c, d = shape
    attn = dict(num_heads=nh, kq_dim=kq, v_dim=vd, out_dim=c, concat=True, kq_dim_division=True, residual=False)
    latent, k, t = 48, 2, 1
    ids = [3, 50, 77, 12, 100, 9]
    if dense:
        nn, ne, s, r = _complete_batch(community_medium[0][ids])
    else:
        nn, ne, s, r = O.batch_graphs(*community_medium, ids)
    n = int(nn.sum())
    rng = np.random.default_rng(1000 * nh + 10 * kq + vd)
    x = rng.standard_normal((n, d)).astype(np.float32)
    p = O.make_attn_grevnet_params(kq + vd, d // 2, latent, k, t, final_scale=0.3, **attn)
    net = make_product_grevnet(_hp(d, latent, k, t, attn), p)
    graph, _ = _forward_inverse_vs_oracle(net, nn, ne, s, r, x, p, t, rng)
    ref = O.loss_and_grads(s, r, n, x, p, t, activation="relu")
    tr = GRevNetTrainer(net)
    out = tr.loss_and_grads(graph)
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    _check_all_grads(tr.named_gradients(), ref["grads"], 5e-4)


def test_head_geometry_beyond_the_abi_limit_is_rejected(grid_small):
    """include/gnf.h states ONE limit (heads <= 64, heads * kq <= 256, heads * v <= 256, pad16(2 heads kq + v) + pad16(H) + H
    <= 1272); forward and backward reject anything beyond it with GNF_ESHAPE instead of running an untested shape - the
    feature-width clause one element past the two edges test_attention_head_geometry_boundaries runs."""
    from gnf_amd import _abi
    from gnf_amd.flow import log_prob_terms
    nn, ne, s, r = _complete_batch(grid_small[0][[6]])
    n = int(nn.sum())
    for nh, kq, vd, d in ((1, 257, 8, 8), (1, 8, 257, 8), (5, 52, 8, 8), (65, 2, 2, 8), (1, 256, 256, 498), (4, 64, 64, 690)):
        x = np.zeros((n, d), np.float32)
        attn = dict(num_heads=nh, kq_dim=kq, v_dim=vd, out_dim=8, concat=True, kq_dim_division=True, residual=False)
        p = O.make_attn_grevnet_params(1, d // 2, 16, 2, 1, final_scale=0.3, **attn)
        net = make_product_grevnet(_hp(d, 16, 2, 1, attn), p)
        with pytest.raises(_abi.GnfError, match="heads"):
            log_prob_terms(net, graph_from_arrays(nn, ne, s, r, x, DEV))


def test_more_than_32_timesteps_pack_their_attention_weights_in_batches(grid_small):
    """T = 33 without weight sharing: 33 s-nets and 33 t-nets.  The backward pass packs Wo^T and [Wq | Wk | Wv]^T of every net
    once per call, 32 nets of either kind per launch (the kernel's argument block holds 64 pointers per array); flows deeper
    than 32 steps used to skip the pack and take the kernels that read the raw weights.  Every gradient against the oracle."""
    from gnf_amd.train import GRevNetTrainer
    d, latent, k, t = 8, 16, 2, 33
    attn = dict(num_heads=2, kq_dim=4, v_dim=4, out_dim=6, concat=True, kq_dim_division=True, residual=False)
    nn, ne, s, r = _complete_batch(grid_small[0][[6, 0, 7]])
    n = int(nn.sum())
    rng = np.random.default_rng(33)
    x = (rng.standard_normal((n, d)) * 0.8).astype(np.float32)
    p = O.make_attn_grevnet_params(34, d // 2, latent, k, t, final_scale=0.1, **attn)
    net = make_product_grevnet(_hp(d, latent, k, t, attn), p)
    graph, _ = _forward_inverse_vs_oracle(net, nn, ne, s, r, x, p, t, rng)
    ref = O.loss_and_grads(s, r, n, x, p, t, activation="relu")
    tr = GRevNetTrainer(net)
    out = tr.loss_and_grads(graph)
    torch.cuda.synchronize()
    assert abs(float(out["total_loss"]) - ref["total_loss"]) <= 1e-4 * n
    _check_all_grads(tr.named_gradients(), ref["grads"], 5e-4)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layered"])
@pytest.mark.parametrize("name", DATA_DRIVER_GOLDEN_CASES)
def test_data_driver_golden_fixture(name, fused):
    g = load_golden(name)
    hp = {k: g[k] for k in ("D", "latent", "K", "T", "agg", "combine", "epsilon", "activation", "weight_sharing")}
    hp["attn"] = g["attn"]
    assert (g["attn"]["num_heads"], g["attn"]["kq_dim"], g["attn"]["v_dim"], g["attn"]["out_dim"]) == (1, 64, 64, 64)
    net = make_product_grevnet(hp, g["params"])
    assert net.use_batch_norm
    net.fused = fused
    from gnf_amd.flow import log_prob_terms
    graph = graph_from_arrays(g["n_node"], g["n_edge"], g["senders"], g["receivers"], g["x"], DEV)
    out = log_prob_terms(net, graph)
    torch.cuda.synchronize()
    n = g["x"].shape[0]
    np.testing.assert_allclose(out["z_graph"].nodes.cpu().numpy(), g["z"], atol=3e-4, rtol=3e-4)
    assert abs(float(out["log_det_jacobian"]) - float(g["logdet"])) <= 1e-4 * n
    assert abs(float(out["log_prob_xs_per_node"]) - float(g["log_prob_xs_per_node"])) <= 1e-4
    for half in range(2):
        for i in range(g["T"]):
            bn = net.bns[half][i]
            np.testing.assert_allclose(bn.batch_mean.cpu().numpy(), g[f"bn_{half}_{i}_batch_mean"], atol=3e-5)
            np.testing.assert_allclose(bn.batch_variance.cpu().numpy(), g[f"bn_{half}_{i}_batch_variance"], rtol=3e-5, atol=3e-5)
    x_back = net(out["z_graph"], inverse=False).nodes.cpu().numpy()
    np.testing.assert_allclose(x_back, g["x_roundtrip"], atol=1e-3, rtol=1e-3)
