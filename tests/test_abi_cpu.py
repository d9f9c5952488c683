"""C-ABI checks that need no GPU: the library loads, exports every symbol include/gnf.h declares,
computes host-side sizes, and rejects bad arguments with the documented codes BEFORE any launch."""
import ctypes as C
import os
import re

import pytest

from gnf_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _abi.lib()
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gnf.h but not exported by libgnf_hip.so"
    # and the binding covers the header exactly
    assert sorted(_abi.EXPORTED_SYMBOLS) == syms
    assert lib.gnf_abi_version() == _abi.GNF_ABI_VERSION


def _mlp(dims, fake_ptr=0x1000):
    m = _abi.GnfMlp()
    m.num_layers = len(dims) - 1
    for j, d in enumerate(dims):
        m.dims[j] = d
    for j in range(len(dims) - 1):
        m.W[j] = fake_ptr
        m.b[j] = fake_ptr
    return m


def test_packed_floats_host_computation():
    lib = _abi.lib()
    m = _mlp([32, 256, 256, 256, 256, 32])
    # per layer: W fragments + W^T fragments (backward) + bias
    want = 2 * 32 * 256 + 256 + 3 * (2 * 256 * 256 + 256) + 2 * 256 * 32 + 32
    assert lib.gnf_packed_floats(C.byref(m)) == want
    m = _mlp([1, 16, 1])                     # D=2 reference default: widths pad to 16
    assert lib.gnf_packed_floats(C.byref(m)) == (2 * 16 * 16 + 16) * 2
    m = _mlp([50, 100, 50])                  # D=100: 50 -> 64, 100 -> 112
    assert lib.gnf_packed_floats(C.byref(m)) == 2 * 64 * 112 + 112 + 2 * 112 * 64 + 64


def test_workspace_sizes_are_monotone():
    lib = _abi.lib()
    s = (_abi.GnfMlp * 2)(_mlp([8, 32, 8]), _mlp([8, 32, 8]))
    flow = _abi.GnfFlow(2, 1, C.cast(s, C.POINTER(_abi.GnfMlp)), C.cast(s, C.POINTER(_abi.GnfMlp)),
                        _abi.GnfGnnSpec(1, 0, 1.0, 1, 0.2))
    a = lib.gnf_workspace_bytes(100, 16, C.byref(flow))
    b = lib.gnf_workspace_bytes(1000, 16, C.byref(flow))
    assert 0 < a < b
    assert lib.gnf_csr_workspace_bytes(4, 100) == (2 * 5 + 101) * 4


def _err():
    return _abi.lib().gnf_last_error().decode()


def test_argument_validation_without_a_gpu():
    lib = _abi.lib()
    csr = _abi.GnfCsr(0x1000, 0x1000, 10, 20)
    nets = (_abi.GnfMlp * 2)(_mlp([4, 8, 4]), _mlp([4, 8, 4]))
    spec = _abi.GnfGnnSpec(1, 0, 1.0, 1, 0.2)
    flow = _abi.GnfFlow(1, 1, C.cast(nets, C.POINTER(_abi.GnfMlp)), C.cast(nets, C.POINTER(_abi.GnfMlp)), spec)
    # odd D: tf.split needs an even width
    rc = lib.gnf_grevnet_f32(C.byref(csr), C.byref(flow), 0x1000, 7, 7, 0, 0x1000, 0x1000, 1 << 20, None)
    assert rc == -2 and "even" in _err()
    # ld < D
    assert lib.gnf_grevnet_f32(C.byref(csr), C.byref(flow), 0x1000, 4, 8, 0, 0x1000, 0x1000, 1 << 20, None) == -2
    # MLP width does not match H
    assert lib.gnf_grevnet_f32(C.byref(csr), C.byref(flow), 0x1000, 16, 16, 0, 0x1000, 0x1000, 1 << 20, None) == -2
    assert "needs" in _err()
    # bad direction
    assert lib.gnf_grevnet_f32(C.byref(csr), C.byref(flow), 0x1000, 8, 8, 5, 0x1000, 0x1000, 1 << 20, None) == -1
    # forward without a sums buffer
    assert lib.gnf_grevnet_f32(C.byref(csr), C.byref(flow), 0x1000, 8, 8, 0, None, 0x1000, 1 << 20, None) == -1
    # workspace too small
    assert lib.gnf_grevnet_f32(C.byref(csr), C.byref(flow), 0x1000, 8, 8, 0, 0x1000, 0x1000, 16, None) == -3
    assert "workspace" in _err()
    # bad enum in the spec
    bad = _abi.GnfGnnSpec(7, 0, 1.0, 1, 0.2)
    assert lib.gnf_coupling_half_f32(C.byref(csr), C.byref(nets[0]), C.byref(nets[1]), C.byref(bad), 0x1000,
                                     0x1000, 8, 4, 0, None, 0x1000, 1 << 20, None) == -1
    # too many layers
    m = _mlp([4, 8, 4])
    m.num_layers = 9
    assert lib.gnf_pack_mlp(C.byref(m), 0x1000, None) == -2
    # null pointers
    assert lib.gnf_aggregate_f32(C.byref(csr), None, 4, 4, 0, 0x1000, 4, None) == -1
    assert lib.gnf_gauss_sumsq_f32(0x1000, 4, 4, 2, 0x1000, 0x1000, 1 << 20, None) == -2
    assert lib.gnf_build_csr(None, None, None, None, 1, 4, 4, 0x1000, 0x1000, 0x1000, 1 << 20, None) == -1


def test_empty_batch_is_a_no_op_success():
    """N = 0 returns GNF_OK without touching the device (validation only)."""
    lib = _abi.lib()
    csr = _abi.GnfCsr(0, 0, 0, 0)
    nets = (_abi.GnfMlp * 2)(_mlp([4, 8, 4]), _mlp([4, 8, 4]))
    spec = _abi.GnfGnnSpec(1, 0, 1.0, 1, 0.2)
    assert lib.gnf_coupling_half_f32(C.byref(csr), C.byref(nets[0]), C.byref(nets[1]), C.byref(spec), None, None,
                                     8, 4, 0, None, None, 0, None) == 0
    assert lib.gnf_aggregate_f32(C.byref(csr), None, 4, 4, 0, None, 4, None) == 0


def test_product_fails_loudly_without_a_hip_device():
    """No CPU fallback: CPU tensors must raise, never silently compute somewhere else."""
    import numpy as np
    import torch
    from helpers import graph_from_arrays, make_product_grevnet
    from gnf_amd.flow import gauss_sumsq
    hp = dict(D=4, latent=8, K=2, T=1, agg="mean", combine="agg", epsilon=1.0, activation="leaky_relu",
              weight_sharing=False)
    net = make_product_grevnet(hp, None)
    g = graph_from_arrays([2], [2], [0, 1], [0, 1], np.zeros((2, 4), np.float32))
    with pytest.raises(_abi.GnfError):
        net(g, inverse=True)
    with pytest.raises(_abi.GnfError):
        net.s[0][0](g.replace(nodes=g.nodes[:, :2]))
    with pytest.raises(_abi.GnfError):
        gauss_sumsq(torch.zeros(2, 4))


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_abi.GnfError):
        _abi.lib()


def test_abi_v5_size_helpers_are_host_computations():
    """gnf_clip_workspace_bytes / gnf_attn_stash_bytes touch no device: 64 fp64 partials per tensor; the stash is
    zero without an attention front-end and 2T slots of 2 n (P + in0 + heads v + 3 heads) floats with one (q | k | v,
    layer-0 inputs, attended values, softmax statistics of both nets)."""
    lib = _abi.lib()
    assert lib.gnf_clip_workspace_bytes(0) == 0
    assert lib.gnf_clip_workspace_bytes(7) == 7 * 64 * 8
    s = (_abi.GnfMlp * 2)(_mlp([8, 32, 8]), _mlp([8, 32, 8]))
    flow = _abi.GnfFlow(3, 1, C.cast(s, C.POINTER(_abi.GnfMlp)), C.cast(s, C.POINTER(_abi.GnfMlp)),
                        _abi.GnfGnnSpec(1, 0, 1.0, 1, 0.2))
    assert lib.gnf_attn_stash_bytes(100, 16, C.byref(flow)) == 0          # message-passing nets: nothing to stash
    at = _abi.GnfAttn()
    at.num_heads, at.kq_dim, at.v_dim, at.out_dim = 4, 5, 6, 12
    sa = (_abi.GnfMlp * 2)(_mlp([20, 32, 8]), _mlp([20, 32, 8]))
    for m in sa:
        m.attn = C.pointer(at)
    fa = _abi.GnfFlow(3, 1, C.cast(sa, C.POINTER(_abi.GnfMlp)), C.cast(sa, C.POINTER(_abi.GnfMlp)),
                      _abi.GnfGnnSpec(1, 0, 1.0, 1, 0.2))
    p = 2 * 4 * 5 + 6
    assert lib.gnf_attn_stash_bytes(100, 16, C.byref(fa)) == 2 * 3 * (2 * 100 * (p + 20 + 4 * 6 + 3 * 4)) * 4
    assert lib.gnf_attn_stash_bytes(0, 16, C.byref(fa)) == 0
