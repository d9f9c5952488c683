#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random flow hyper-parameters x random batches (ragged graph sizes,
isolated nodes, duplicated and directed edges) through forward / inverse / gradients, fused and layered / GEMM
paths, against the float64 oracle.  `python tools/fuzz_parity.py --cases 200 --seed 0`; exits non-zero on the
first mismatch and prints the failing configuration (re-run it with --only <index>)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import gnf_oracle as O                      # noqa: E402
from helpers import graph_from_arrays, make_product_grevnet   # noqa: E402


def random_batch(rng):
    b = int(rng.integers(1, 6))
    n_node = rng.integers(1, 40, size=b)
    s_all, r_all, n_edge, off = [], [], [], 0
    for n in n_node:
        style = rng.integers(0, 4)
        if style == 0:      # sparse random directed, maybe duplicates, isolated nodes likely
            e = int(rng.integers(0, 3 * n + 1))
            s, r = rng.integers(0, n, size=e), rng.integers(0, n, size=e)
        elif style == 1:    # symmetric ring + self loops
            i = np.arange(n)
            s = np.concatenate([i, i, (i + 1) % n])
            r = np.concatenate([i, (i + 1) % n, i])
        elif style == 2:    # complete with self loops
            s, r = np.repeat(np.arange(n), n), np.tile(np.arange(n), n)
        else:               # star into node 0 (one high-degree receiver) + a few random edges
            s = np.concatenate([np.arange(n), rng.integers(0, n, size=n // 2)])
            r = np.concatenate([np.zeros(n, int), rng.integers(0, n, size=n // 2)])
        s_all.append(s + off)
        r_all.append(r + off)
        n_edge.append(len(s))
        off += int(n)
    return (n_node.astype(np.int32), np.asarray(n_edge, np.int32), np.concatenate(s_all).astype(np.int32),
            np.concatenate(r_all).astype(np.int32))


def random_case(rng):
    d = int(2 * rng.integers(1, 20))
    hp = dict(D=d, latent=int(rng.integers(4, 80)), K=int(rng.integers(1, 5)), T=int(rng.integers(1, 4)),
              agg=str(rng.choice(["sum", "mean"])), combine=str(rng.choice(["agg", "concat"])),
              epsilon=float(rng.choice([0.0, 0.5, 1.0])), activation=str(rng.choice(["relu", "leaky_relu"])),
              weight_sharing=bool(rng.integers(0, 2)))
    attn = None
    if rng.random() < 0.3:
        attn = dict(num_heads=int(rng.integers(1, 5)), kq_dim=int(rng.integers(1, 8)), v_dim=int(rng.integers(1, 8)),
                    out_dim=int(rng.integers(1, 12)), concat=bool(rng.integers(0, 2)),
                    kq_dim_division=bool(rng.integers(0, 2)), residual=False)
        if rng.random() < 0.3:          # --attn_layer_norm (gnn.py:550-552), with or without the residual
            attn.update(layer_norm=True, residual=bool(rng.integers(0, 2)))
        hp.update(attn=attn, activation="relu", agg="mean", combine="agg", epsilon=0.0)
    return hp, attn, rng.random() < 0.4


def _run_case(idx, seed, verbose=False, perturb=0.0):
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.train import GRevNetTrainer
    rng = np.random.default_rng([seed, idx])
    hp, attn, use_bn = random_case(rng)
    nn, ne, s, r = random_batch(rng)
    n, d, t, ws = int(nn.sum()), hp["D"], hp["T"], hp["weight_sharing"]
    x = rng.standard_normal((n, d)).astype(np.float32)
    if perturb:
        x = (x + perturb * np.random.default_rng(99).standard_normal(x.shape)).astype(np.float32)
    fs = 0.3 if hp["agg"] == "mean" else 0.05
    if attn:
        p = O.make_attn_grevnet_params(idx, d // 2, hp["latent"], hp["K"], t, weight_sharing=ws, final_scale=0.3, **attn)
    else:
        p = O.make_grevnet_params(idx, d // 2, hp["latent"], hp["K"], t, combine=hp["combine"], weight_sharing=ws,
                                  final_scale=fs)
    if use_bn and n >= 4:
        p["bn"] = O.make_bn_params(idx + 7, d // 2, t)
    kw = dict(agg=hp["agg"], combine=hp["combine"], epsilon=hp["epsilon"], activation=hp["activation"])
    desc = f"case {idx}: {hp} bn={'bn' in p} n_node={nn.tolist()} E={len(s)}"
    if verbose:
        print(desc, flush=True)
    ref = O.loss_and_grads(s, r, n, x, p, t, ws, **kw)
    if not np.isfinite(ref["total_loss"]) or np.abs(ref["z"]).max() > 1e3:
        return "skipped (oracle overflow)"
    o64 = O.Fp64Dense(s, r, n, **kw)
    zs = rng.standard_normal((n, d)).astype(np.float32)
    xg_ref = o64.g(zs, p, t, ws)
    graph = graph_from_arrays(nn, ne, s, r, x, "cuda:0")
    for fused in (True, False):
        net = make_product_grevnet(hp, p)
        net.fused = fused
        out = log_prob_terms(net, graph)
        z = out["z_graph"].nodes.cpu().numpy()
        tol = 5e-4 * max(1.0, float(np.abs(ref["z"]).max()))
        assert np.abs(z - ref["z"]).max() <= tol, f"{desc}\n fused={fused}: z max err {np.abs(z - ref['z']).max():.3e}"
        lp_ref = -ref["total_loss"] / n
        assert abs(float(out["log_prob_xs_per_node"]) - lp_ref) <= 2e-4 * max(1.0, abs(lp_ref)), \
            f"{desc}\n fused={fused}: log-prob {float(out['log_prob_xs_per_node'])} vs {lp_ref}"
        if np.abs(xg_ref).max() < 1e3:
            xg = net(graph.replace(nodes=torch.as_tensor(zs).cuda()), inverse=False).nodes.cpu().numpy()
            tolg = 5e-4 * max(1.0, float(np.abs(xg_ref).max()))
            assert np.abs(xg - xg_ref).max() <= tolg, f"{desc}\n fused={fused}: inverse max err {np.abs(xg - xg_ref).max():.3e}"
        tr = GRevNetTrainer(net)
        bw = tr.loss_and_grads(graph)
        torch.cuda.synchronize()
        # reversible back-propagation rebuilds every half-step's input with the inverse update; in an
        # ill-conditioned flow (large |s|) that reconstruction loses fp32 digits and so do the gradients - inherent
        # to the algorithm (a float32 autograd run that STORES the activations stays at 1e-6).  Judge gradients only
        # where the round trip is healthy.
        recon = float((bw["reconstruction"] - graph.nodes).abs().max())
        if recon > 2e-5 * max(1.0, float(np.abs(x).max())):
            return "skipped (ill-conditioned flow: reversible reconstruction error %.1e)" % recon
        got = tr.named_gradients()
        gmax = [0.0]

        def scan(b):
            if isinstance(b, dict):
                for v in b.values():
                    scan(v)
            elif isinstance(b, (list, tuple)) and not isinstance(b, np.ndarray):
                for v in b:
                    scan(v)
            else:
                gmax[0] = max(gmax[0], float(np.abs(b).max()))
        scan(ref["grads"])

        def walk(a, b, path):
            if isinstance(b, dict):
                for k in b:
                    if k in a:
                        walk(a[k], b[k], path + "." + k)
            elif isinstance(b, (list, tuple)) and not isinstance(b, np.ndarray):
                for i, (aa, bb) in enumerate(zip(a, b)):
                    walk(aa, bb, f"{path}[{i}]")
            else:
                scale = float(np.abs(b).max())
                err = float(np.abs(np.asarray(a) - b).max())
                # relu / leaky_relu kinks and fp32 through exp(s): generous but meaningful.  A gradient that is
                # zero by cancellation (e.g. the bias of t in front of a batch-norm bijector) is judged against the
                # flow's overall gradient scale.
                assert err <= 3e-3 * scale + 2e-4 + 1e-4 * gmax[0], \
                    f"{desc}\n fused={fused}: grad{path} err {err:.3e} scale {scale:.3e} gmax {gmax[0]:.3e}"
        walk(got, ref["grads"], "")
    return "ok"


def run_case(idx, seed, verbose=False):
    """A gradient mismatch is re-tried once on a slightly perturbed input: relu / leaky_relu are not differentiable
    at 0, and an activation within fp32 rounding of 0 legitimately takes the other branch of act' than the float64
    oracle (the float32 autograd run of the oracle shows the same deviations, tools/fuzz_diag.py); such a
    coincidence disappears under perturbation, a bug does not."""
    try:
        return _run_case(idx, seed, verbose)
    except AssertionError as e:
        if ": grad" not in str(e):
            raise
        res = _run_case(idx, seed, verbose, perturb=1e-3)
        return "ok (activation-kink coincidence, passes perturbed)" if res == "ok" else res


def _run_wide_case(idx, seed, verbose=False, big=False):
    """--wide: nets too wide for the fused kernels on batches of a few thousand nodes - the layered path's kernels
    (k_linear_short, k_linear_big with and without the thin last layer in its epilogue, split-K tiles, the slab-adding coupling
    kernels, the MLP-row stash in its layered mode) with random widths: hidden widths that are no multiple of 16 or 256,
    input widths on both sides of every dispatch rule, 2 - 4 layers, batch norm, the data driver's attention block."""
    from gnf_amd.flow import log_prob_terms
    from gnf_amd.train import GRevNetTrainer
    rng = np.random.default_rng([seed, idx, 77])
    d = int(rng.choice([8, 14, 24, 52, 100, 128, 200, 256, 328, 400]))
    latent = int(rng.choice([512, 528, 640, 1000, 1040, 1100, 1280, 1536]))
    if big:   # --big: nets the fused kernels hold, on batches large enough for their large-batch forms (k_half_big, split row tiles, kernel A)
        d = int(rng.choice([8, 14, 24, 64, 100, 128, 256]))
        latent = int(rng.choice([32, 64, 100, 128, 200, 256]))
    hp = dict(D=d, latent=latent, K=int(rng.integers(2, 5)), T=int(rng.integers(1, 3)), agg=str(rng.choice(["mean", "mean", "sum"])),
              combine=str(rng.choice(["agg", "concat"])), epsilon=float(rng.choice([0.0, 1.0])),
              activation=str(rng.choice(["relu", "leaky_relu"])), weight_sharing=bool(rng.random() < 0.2))
    attn = None
    if rng.random() < 0.35:
        w = int(rng.choice([16, 32, 64]))
        attn = dict(num_heads=1, kq_dim=w, v_dim=w, out_dim=int(rng.choice([16, 40, 64])), concat=True, kq_dim_division=True,
                    residual=False)
        if big:   # the drivers' head shapes and their neighbours
            w = int(rng.choice([4, 10, 16, 32]))
            attn.update(num_heads=int(rng.choice([1, 2, 8])), kq_dim=w, v_dim=int(rng.choice([w, 10])), out_dim=int(rng.choice([8, 20, 80])))
        if rng.random() < 0.3:          # --attn_layer_norm (gnn.py:550-552), with or without the residual
            attn.update(layer_norm=True, residual=bool(rng.integers(0, 2)))
        hp.update(attn=attn, activation="relu", agg="mean", combine="agg", epsilon=0.0)
    use_bn = rng.random() < 0.5
    if os.environ.get("FUZZ_BN"):        # (diagnosis: the same case with one thing changed)
        use_bn = os.environ["FUZZ_BN"] == "1"
    if os.environ.get("FUZZ_ACT"):
        hp["activation"] = os.environ["FUZZ_ACT"]
    if os.environ.get("FUZZ_K"):
        hp["K"] = int(os.environ["FUZZ_K"])
    if os.environ.get("FUZZ_T"):
        hp["T"] = int(os.environ["FUZZ_T"])
    if os.environ.get("FUZZ_D"):
        hp["D"] = d = int(os.environ["FUZZ_D"])
    if os.environ.get("FUZZ_ATTN"):      # "heads,kq,v,C" or "0"
        if os.environ["FUZZ_ATTN"] == "0":
            attn = None
            hp.pop("attn", None)
        else:
            nh_, kq_, v_, c_ = (int(v) for v in os.environ["FUZZ_ATTN"].split(","))
            attn = dict(num_heads=nh_, kq_dim=kq_, v_dim=v_, out_dim=c_, concat=True, kq_dim_division=True, residual=False)
            hp.update(attn=attn, activation="relu", agg="mean", combine="agg", epsilon=0.0)
    if os.environ.get("FUZZ_LATENT"):
        hp["latent"] = latent = int(os.environ["FUZZ_LATENT"])
    sizes, tot = [], 0
    target = int(rng.integers(4000, 12000)) if big else int(rng.integers(1700, 3400))
    if os.environ.get("FUZZ_NODES"):
        target = int(os.environ["FUZZ_NODES"])
    while tot < target:
        m = int(rng.integers(6, 60)) if not big or rng.random() < 0.8 else int(rng.integers(100, 400))
        sizes.append(m)
        tot += m
    s_l, r_l, ne, off = [], [], [], 0
    for m in sizes:
        if m < 100 and (attn or rng.random() < 0.5):   # complete graphs with self loops (the data driver's topology)
            a, b = np.repeat(np.arange(m), m), np.tile(np.arange(m), m)
        else:                             # symmetric ring + a few random edges
            i = np.arange(m)
            a = np.concatenate([i, (i + 1) % m, rng.integers(0, m, size=m)])
            b = np.concatenate([(i + 1) % m, i, rng.integers(0, m, size=m)])
        s_l.append(a + off), r_l.append(b + off), ne.append(len(a))
        off += m
    nn, ne = np.asarray(sizes, np.int32), np.asarray(ne, np.int32)
    s, r = np.concatenate(s_l).astype(np.int32), np.concatenate(r_l).astype(np.int32)
    n, t = int(nn.sum()), hp["T"]
    x = (rng.standard_normal((n, d)) * 0.7).astype(np.float32)
    if os.environ.get("FUZZ_PERTURB"):
        x = (x + 1e-3 * np.random.default_rng(int(os.environ["FUZZ_PERTURB"])).standard_normal(x.shape)).astype(np.float32)
    ws = hp["weight_sharing"]
    fs = 0.3 if hp["agg"] == "mean" else 0.02   # (sum over tens of neighbours: keep |s| moderate)
    if attn:
        p = O.make_attn_grevnet_params(idx, d // 2, latent, hp["K"], t, weight_sharing=ws, final_scale=0.3, **attn)
    else:
        p = O.make_grevnet_params(idx, d // 2, latent, hp["K"], t, combine=hp["combine"], weight_sharing=ws, final_scale=fs)
    if use_bn:
        p["bn"] = O.make_bn_params(idx + 7, d // 2, t)
    kw = dict(agg=hp["agg"], combine=hp["combine"], epsilon=hp["epsilon"], activation=hp["activation"])
    desc = f"wide case {idx}: {hp} bn={use_bn} n={n} graphs={len(sizes)} E={len(s)}"
    if verbose:
        print(desc, flush=True)
    ref = O.loss_and_grads(s, r, n, x, p, t, ws, **kw)
    if not np.isfinite(ref["total_loss"]) or np.abs(ref["z"]).max() > 1e3:
        return "skipped (oracle overflow)"
    graph = graph_from_arrays(nn, ne, s, r, x, "cuda:0")
    net = make_product_grevnet(hp, p)
    out = log_prob_terms(net, graph)
    z = out["z_graph"].nodes.cpu().numpy()
    tol = 5e-4 * max(1.0, float(np.abs(ref["z"]).max()))
    assert np.abs(z - ref["z"]).max() <= tol, f"{desc}\n z max err {np.abs(z - ref['z']).max():.3e}"
    lp_ref = -ref["total_loss"] / n
    assert abs(float(out["log_prob_xs_per_node"]) - lp_ref) <= 2e-4 * max(1.0, abs(lp_ref)), \
        f"{desc}\n log-prob {float(out['log_prob_xs_per_node'])} vs {lp_ref}"
    if not use_bn:   # g(f(x)) = x through the same kernels (the bijectors' moving statistics aside)
        back = net(out["z_graph"], inverse=False).nodes.cpu().numpy()
        assert np.abs(back - x).max() <= 5e-4 * max(1.0, float(np.abs(x).max())), f"{desc}\n round trip {np.abs(back - x).max():.3e}"
    flat = []

    def scan(b, path):
        if isinstance(b, dict):
            for k in b:
                scan(b[k], path + "." + k)
        elif isinstance(b, (list, tuple)) and not isinstance(b, np.ndarray):
            for i, v in enumerate(b):
                scan(v, f"{path}[{i}]")
        else:
            flat.append((path, b))
    scan(ref["grads"], "")
    gmax = max(float(np.abs(b).max()) for _, b in flat)
    # Bound per tensor, in the 2-norm: 6 x what the SAME autograd costs in float32 on the CPU on these inputs (relu kinks: a
    # pre-activation within single-precision rounding of 0 takes the other branch of act' than the float64 run), at least 2e-3
    r32 = O.loss_and_grads(s, r, n, x, p, t, ws, dtype=torch.float32, **kw)
    f32 = []
    flat64, flat[:] = list(flat), []
    scan(r32["grads"], "")
    f32, flat = list(flat), flat64
    floor = lambda c: max(float(np.linalg.norm(c)), 1e-3 * gmax * np.sqrt(c.size))   # noqa: E731
    rel32 = max(float(np.linalg.norm(np.asarray(a) - c)) / floor(c) for (_, a), (_, c) in zip(f32, flat))
    rel_bound = max(2e-3, 6.0 * rel32)
    failed = []
    for stash in (True, False):
        tr = GRevNetTrainer(make_product_grevnet(hp, p))
        tr.stash_mlp_rows = stash
        bw = tr.loss_and_grads(graph)
        torch.cuda.synchronize()
        assert abs(float(bw["total_loss"]) - ref["total_loss"]) <= 2e-4 * max(n, abs(ref["total_loss"])), f"{desc}\n stash={stash}: loss"
        got = tr.named_gradients()

        def pick(a, path):
            for tok in path.replace("]", "").replace("[", ".").split(".")[1:]:
                a = a[int(tok)] if tok.isdigit() else a[tok]
            return np.asarray(a)
        recon = float((bw["reconstruction"] - graph.nodes).abs().max())
        errs = []
        for path, c in flat:
            try:
                a = pick(got, path)
            except (KeyError, IndexError, TypeError):
                continue
            errs.append((float(np.linalg.norm(a - c)) / floor(c), path))
        errs.sort(reverse=True)
        if verbose:
            print(f"  stash={stash}: reconstruction max err {recon:.2e}; worst tensors (2-norm, relative): "
                  + ", ".join(f"{p_} {e:.1e}" for e, p_ in errs[:6]) + f"; float32 CPU autograd worst {rel32:.1e}", flush=True)
        if errs[0][0] > rel_bound:
            failed.append(f"{desc}\n stash={stash}: grad{errs[0][1]} 2-norm err {errs[0][0]:.3e} relative > {rel_bound:.3e} (float32 CPU autograd: {rel32:.2e})")
    assert not failed, "\n".join(failed)
    return "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--wide", action="store_true", help="the layered path's kernels: nets too wide for the fused kernels on batches of thousands of nodes")
    ap.add_argument("--big", action="store_true", help="the fused kernels' large-batch forms: nets they hold on batches of 4 000 - 12 000 nodes")
    args = ap.parse_args()
    idxs = [args.only] if args.only is not None else range(args.cases)
    counts = {}
    def wide(i, seed, verbose):
        """One flipped relu costs ~1e-3 of a bias gradient's norm on these batches (a column sum of ~2 000 random-sign terms),
        and the float32 CPU autograd shows flips of its own (1e-3 .. 3e-3) on about half of the inputs: a case that misses the
        bound is re-run on up to three slightly perturbed inputs and passes when one of them has no flip on the device (every
        tensor at ~3e-5 then); a defect does not go away under a 1e-3 perturbation."""
        try:
            return _run_wide_case(i, seed, verbose, args.big)
        except AssertionError as e:
            if ": grad" not in str(e):
                raise
            first = str(e)
        for k in (1, 2, 3):
            os.environ["FUZZ_PERTURB"] = str(k)
            try:
                if _run_wide_case(i, seed, verbose, args.big) == "ok":
                    return "ok (activation-kink coincidence, passes perturbed)"
            except AssertionError as e:
                if ": grad" not in str(e):
                    raise
            finally:
                os.environ.pop("FUZZ_PERTURB", None)
        raise AssertionError(first)
    for i in idxs:
        res = (wide if args.wide or args.big else run_case)(i, args.seed, verbose=args.only is not None or args.wide or args.big)
        counts[res] = counts.get(res, 0) + 1
    print("fuzz parity:", counts)


if __name__ == "__main__":
    main()
