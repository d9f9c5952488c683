"""Developer diagnostic for one fuzz_parity case: HIP gradient (both paths) vs float64 oracle vs a float32 autograd
run of the same oracle (the intrinsic fp32 / activation-kink noise floor)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import fuzz_parity as fz
from oracle import gnf_oracle as O
from helpers import graph_from_arrays, make_product_grevnet
from gnf_amd.train import GRevNetTrainer

seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed, idx])
hp, attn, use_bn = fz.random_case(rng)
nn, ne, s, r = fz.random_batch(rng)
n, d, t, ws = int(nn.sum()), hp["D"], hp["T"], hp["weight_sharing"]
x = rng.standard_normal((n, d)).astype(np.float32)
if len(sys.argv) > 3:   # perturb the input: an activation-kink coincidence disappears, a bug does not
    x = (x + float(sys.argv[3]) * np.random.default_rng(99).standard_normal(x.shape)).astype(np.float32)
fs = 0.3 if hp["agg"] == "mean" else 0.05
p = (O.make_attn_grevnet_params(idx, d // 2, hp["latent"], hp["K"], t, weight_sharing=ws, final_scale=0.3, **attn) if attn
     else O.make_grevnet_params(idx, d // 2, hp["latent"], hp["K"], t, combine=hp["combine"], weight_sharing=ws, final_scale=fs))
if use_bn and n >= 4:
    p["bn"] = O.make_bn_params(idx + 7, d // 2, t)
kw = dict(agg=hp["agg"], combine=hp["combine"], epsilon=hp["epsilon"], activation=hp["activation"])
ref = O.loss_and_grads(s, r, n, x, p, t, ws, **kw)

def flat(g, out, path=""):
    if isinstance(g, dict):
        for k, v in g.items(): flat(v, out, path + "." + k)
    elif isinstance(g, (list, tuple)) and not isinstance(g, np.ndarray):
        for i, v in enumerate(g): flat(v, out, f"{path}[{i}]")
    else: out[path] = np.asarray(g)
    return out
fr = flat(ref["grads"], {})
# float32 autograd of the oracle: perturb inputs by casting through fp32 arithmetic
import oracle.gnf_oracle as OO
_orig = OO.Fp32Gather.__init__
def _init32(self, *a, **k):
    k["dtype"] = torch.float32
    _orig(self, *a, **k)
OO.Fp32Gather.__init__ = _init32
r32 = O.loss_and_grads(s, r, n, x, p, t, ws, **kw)
OO.Fp32Gather.__init__ = _orig
f32 = flat(r32["grads"], {})
graph = graph_from_arrays(nn, ne, s, r, x, "cuda:0")
res = {}
for fused in (True, False):
    net = make_product_grevnet(hp, p); net.fused = fused
    tr = GRevNetTrainer(net); tr.loss_and_grads(graph); torch.cuda.synchronize()
    res[fused] = flat(tr.named_gradients(), {})
print(hp, "bn", "bn" in p, "n", n)
worst = sorted(fr, key=lambda k: -np.abs(res[True][k] - fr[k]).max() / (np.abs(fr[k]).max() + 1e-30))[:6]
for k in worst:
    sc = np.abs(fr[k]).max()
    print(f"{k:28s} scale {sc:9.3e} | hip fused {np.abs(res[True][k]-fr[k]).max()/sc:8.2e} | hip gemm {np.abs(res[False][k]-fr[k]).max()/sc:8.2e} "
          f"| fp32 oracle {np.abs(f32[k]-fr[k]).max()/sc:8.2e} | fused-vs-gemm {np.abs(res[True][k]-res[False][k]).max()/sc:8.2e}")
