"""Developer probe: phase timing (s_memtime ticks, 100 MHz) of k_attn_fwd_rows on the moons_100 default batch.
Build the trace library first:  GNF_EXTRA_FLAGS=-DGNF_ATTN_TRACE python __graft_entry__.py  (or see tools/build_variants.sh)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from functools import partial
from gnf_amd import gnn, _abi
from gnf_amd.grevnet_synthetic_data import DATASETS_MAP
g = DATASETS_MAP["moons_100"].get_next_batch(32, "cuda:0")
mk = partial(gnn.dm_self_attn_gnn, kq_dim=10, v_dim=10, make_mlp_fn=partial(gnn.make_mlp_model, 256, 1, 5, gnn.relu, 0.1, 0.1),
             num_heads=8, concat_heads_output_dim=80)
net = gnn.GRevNet(mk, 1, 2)
for _ in range(3):
    net(g, inverse=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
_abi.lib().gnf_debug_read_attn_trace(buf)
t = list(buf)
names = ["rowptr+Wo staged", "window staged", "cols staged", "edge loop", "barrier", "projection + write"]
for i, nm in enumerate(names):
    print(f"{nm:22s} {(t[i + 1] - t[i]) / 100.0:8.2f} us")
print(f"{'total':22s} {(t[6] - t[0]) / 100.0:8.2f} us")
