#!/usr/bin/env python3
"""Developer probe: per-wave s_memtime stamps of TWO workgroups of the large-batch fused kernel (k_half_big) on a
bench workload (needs a -DGNF_BIG_TRACE build: tools/build_big_variant.sh bigtrace "-DGNF_BIG_TRACE").
  python tools/probe_big_trace.py [variant] [workload] [force_shape] [block0] [block1]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else "bigtrace"
workload = sys.argv[2] if len(sys.argv) > 2 else "config4"
force = int(sys.argv[3]) if len(sys.argv) > 3 else 40
b0 = int(sys.argv[4]) if len(sys.argv) > 4 else 0
b1 = int(sys.argv[5]) if len(sys.argv) > 5 else 256
os.environ["GNF_LIB_PATH"] = os.path.join(ROOT, "graph-normalizing-flows_amd", "variants", f"libgnf_{variant}.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import make_product_grevnet  # noqa: E402
from gnf_amd import _abi  # noqa: E402
from gnf_amd.graphs import csr_of, data_dicts_to_graphs_tuple  # noqa: E402

dev = torch.device("cuda:0")
bench.WORKLOAD = bench.WORKLOADS[workload]
bench.GRAPHS_PER_GPU = int(sys.argv[6]) if len(sys.argv) > 6 else bench.WORKLOAD["graphs"]
bench.HP.update(bench.WORKLOAD["hp"])
HP = bench.HP
dicts, n, e = bench.make_batch(1, 0)
graph = data_dicts_to_graphs_tuple(dicts, dev)
net = make_product_grevnet(HP, bench.make_params(bench.WEIGHT_SEED, HP, bench.FINAL_SCALE))
lib = _abi.lib()
raw = C.CDLL(os.environ["GNF_LIB_PATH"])
_abi.set_option("force_shape", force)
h = HP["D"] // 2
K = HP["K"]
flow = net._flow(h, dev)
csr = csr_of(graph)
ws_bytes = lib.gnf_workspace_bytes(n, HP["D"], C.byref(flow))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
buf = graph.nodes.clone()
st = _abi.stream_ptr(dev)
assert raw.gnf_debug_big_trace(None, None, b0, b1) == 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    ev0.record()
    for q in (0, 1):
        lib.gnf_coupling_half_f32(C.byref(csr.desc), C.byref(flow.s_nets[q]), C.byref(flow.t_nets[q]),
                                  C.byref(flow.gnn), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 4 * h),
                                  buf.stride(0), h, 0, None, _abi.ptr(ws), ws_bytes, st)
    ev1.record()
torch.cuda.synchronize()
print(f"# {workload}: N = {n}, force_shape {force}; two half-steps {ev0.elapsed_time(ev1) * 1e3:.1f} us; traced blocks {b0}, {b1}")
out = (C.c_ulonglong * (2 * 8 * 64))()
hw = (C.c_uint * 16)()
assert raw.gnf_debug_big_trace(out, hw, -1, -1) == 0
t = np.array(list(out), dtype=np.int64).reshape(2, 8, 64)
hwid = np.array(list(hw), dtype=np.int64).reshape(2, 8)
for s in range(2):
    print(f"# block slot {s}: " + " ".join(f"w{w}: xcc {hwid[s, w] >> 16} se {(hwid[s, w] >> 13) & 7} cu {(hwid[s, w] >> 8) & 15} simd {(hwid[s, w] >> 4) & 3}" for w in (0, 1)))
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
names = {0: "start", 1: "rowptr+bar", 2: "h0 staged", 3: "bar"}
for net_ in range(2):
    for j in range(K):
        base = 4 + 6 * (net_ * K + j)
        tag = f"{'st'[net_]}{j}"
        names[base] = tag + " enter"
        names[base + 1] = tag + " mfma done"
        names[base + 2] = tag + " bar A"
        names[base + 3] = tag + " wr issued"
        names[base + 4] = tag + " drained"
        names[base + 5] = tag + " bar B"
names[60] = "coupling top"
names[61] = "coupling done"
names[63] = "end"
print("s_memtime ticks (100 MHz: 10 ns per tick) relative to the first traced wave's start; columns = waves 0..7 of block slot 0, then slot 1")
for sl in sorted(names):
    if t[:, :, sl].max() == 0 or (sl == 63 and False):
        continue
    nw = int((t[0, :, 0] > 0).sum())
    print(f"{names[sl]:14s} " + " ".join(f"{int(t[0, w, sl] - t0):7d}" for w in range(nw)) + "  |  " + " ".join(f"{int(t[1, w, sl] - t0):7d}" for w in range(nw)))

# ---- every workgroup's span.  s_memtime is a PER-CU clock (offsets between CUs are arbitrary): durations and per-CU
# timelines come from it; the launch's ramp and drain across CUs from s_memrealtime (100 MHz, one clock for the chip) ----
sp = (C.c_ulonglong * (8192 * 5))()
if hasattr(raw, "gnf_debug_big_spans") and raw.gnf_debug_big_spans(sp) == 0:
    a = np.array(list(sp), dtype=np.int64).reshape(8192, 5)
    nblk = int((a[:, 1] > 0).sum())          # (the array is zero before the first launch of this process)
    a = a[:nblk]
    if os.environ.get("GNF_SPANS_OUT"):
        np.save(os.environ["GNF_SPANS_OUT"], a)
    hw = a[:, 2] & 0xffff
    cu = (hw >> 8) & 15 | (((hw >> 13) & 7) << 4) | ((a[:, 2] >> 16) << 8)
    dur = a[:, 1] - a[:, 0]
    r0 = a[:, 3].min()
    rs, re = (a[:, 3] - r0) * 10, (a[:, 4] - r0) * 10   # ns since the first workgroup's start
    print(f"# {nblk} workgroups on {len(np.unique(cu))} CUs; launch span {re.max() / 1e3:.1f} us (first start to last end, s_memrealtime)")
    first = np.argsort(rs)[:min(nblk, 2 * len(np.unique(cu)))]
    print(f"#   ramp: the first {len(first)} workgroups start within {rs[first].max() / 1e3:.2f} us (median {np.median(rs[first]) / 1e3:.2f}); "
          f"last end per CU: min {min(re[cu == c].max() for c in np.unique(cu)) / 1e3:.1f} us, median {np.median([re[cu == c].max() for c in np.unique(cu)]) / 1e3:.1f}, max {re.max() / 1e3:.1f}")
    for lo in range(0, nblk, 256):
        part = slice(lo, min(nblk, lo + 256))
        print(f"#   blocks {lo:5d}..{min(nblk, lo + 256) - 1:5d}: start {rs[part].mean() / 1e3:7.1f} us (min {rs[part].min() / 1e3:7.1f} max {rs[part].max() / 1e3:7.1f})  "
              f"duration mean {dur[part].mean():9.0f} cycles (min {dur[part].min()} max {dur[part].max()}) = {(re[part] - rs[part]).mean() / 1e3:.1f} us")
    one = two = 0
    for c in np.unique(cu):
        m = cu == c
        ev = sorted([(t_, 1) for t_ in a[m, 0]] + [(t_, -1) for t_ in a[m, 1]])
        lvl, last = 0, 0
        for t_, d in ev:
            if lvl == 1:
                one += t_ - last
            elif lvl >= 2:
                two += t_ - last
            lvl += d
            last = t_
    ncu = len(np.unique(cu))
    cnt = np.bincount(np.unique(cu, return_inverse=True)[1])
    print(f"# per CU: cycles with one resident workgroup {one / ncu:.0f}, with two {two / ncu:.0f}; workgroups per CU min {cnt.min()} max {cnt.max()}")
    print(f"# clock: {dur.sum() / ((re - rs).sum() / 1e3):.0f} cycles per us")
