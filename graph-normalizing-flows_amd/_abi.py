"""ctypes binding of libgnf_hip.so (the C ABI declared in include/gnf.h).

This is the whole FFI surface: plain pointers and sizes, PyTorch tensors only as the owners of
device memory (`tensor.data_ptr()`) and of the stream (`torch.cuda.current_stream().cuda_stream`).
There is NO fallback: if the HIP library is missing or a call fails, we raise.
"""
import ctypes as C
import os

GNF_MAX_LAYERS = 8
GNF_ABI_VERSION = 9

GNF_AGG_SUM, GNF_AGG_MEAN = 0, 1
GNF_COMBINE_EPS, GNF_COMBINE_CONCAT = 0, 1
GNF_ACT_RELU, GNF_ACT_LEAKY_RELU = 0, 1
GNF_FORWARD, GNF_INVERSE = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNF_LIB_PATH: developer override (a library built from another checkout, for A/B runs); still a HIP build
LIB_PATH = os.environ.get("GNF_LIB_PATH") or os.path.join(_HERE, "libgnf_hip.so")


class GnfError(RuntimeError):
    pass


class GnfCsr(C.Structure):
    _fields_ = [("rowptr", C.c_void_p), ("col", C.c_void_p), ("n_nodes", C.c_int64),
                ("n_edges", C.c_int64)]


class GnfAttn(C.Structure):
    _fields_ = [("num_heads", C.c_int32), ("kq_dim", C.c_int32), ("v_dim", C.c_int32), ("out_dim", C.c_int32),
                ("concat", C.c_int32), ("kq_dim_division", C.c_int32), ("residual", C.c_int32),
                ("layer_norm", C.c_int32), ("Wq", C.c_void_p), ("Wk", C.c_void_p), ("Wv", C.c_void_p),
                ("Wo", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p)]


class GnfMlp(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("dims", C.c_int32 * (GNF_MAX_LAYERS + 1)),
                ("W", C.c_void_p * GNF_MAX_LAYERS), ("b", C.c_void_p * GNF_MAX_LAYERS),
                ("packed", C.c_void_p), ("attn", C.POINTER(GnfAttn))]


class GnfGnnSpec(C.Structure):
    _fields_ = [("agg", C.c_int32), ("combine", C.c_int32), ("epsilon", C.c_float),
                ("activation", C.c_int32), ("alpha", C.c_float)]


class GnfBatchNorm(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("moving_mean", C.c_void_p),
                ("moving_variance", C.c_void_p), ("batch_mean", C.c_void_p), ("batch_variance", C.c_void_p),
                ("epsilon", C.c_float), ("reserved", C.c_int32)]


# ABI v5: int hook(void* ctx, double* device_buf, int64_t count, gnf_stream_t stream) - in-place SUM all-reduce
BN_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class GnfFlow(C.Structure):
    _fields_ = [("num_timesteps", C.c_int32), ("weight_sharing", C.c_int32),
                ("s_nets", C.POINTER(GnfMlp)), ("t_nets", C.POINTER(GnfMlp)), ("gnn", GnfGnnSpec),
                ("bns", C.POINTER(GnfBatchNorm)),
                ("bn_allreduce", BN_ALLREDUCE_FN), ("bn_allreduce_ctx", C.c_void_p), ("bn_sync_buf", C.c_void_p),
                ("attn_stash", C.c_void_p), ("attn_stash_bytes", C.c_size_t),
                ("mlp_stash", C.c_void_p), ("mlp_stash_bytes", C.c_size_t)]


_SIGNATURES = {
    "gnf_abi_version": (C.c_int, []),
    "gnf_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "gnf_get_option": (C.c_int64, [C.c_char_p]),
    "gnf_attn_stash_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.POINTER(GnfFlow)]),
    "gnf_mlp_stash_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.POINTER(GnfFlow)]),
    "gnf_last_error": (C.c_char_p, []),
    "gnf_packed_floats": (C.c_int64, [C.POINTER(GnfMlp)]),
    "gnf_pack_mlp": (C.c_int, [C.POINTER(GnfMlp), C.c_void_p, C.c_void_p]),
    "gnf_csr_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gnf_build_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnf_aggregate_f32": (C.c_int, [C.POINTER(GnfCsr), C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_int64, C.c_void_p]),
    "gnf_gnn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.POINTER(GnfMlp), C.c_int32]),
    "gnf_gnn_apply_f32": (C.c_int, [C.POINTER(GnfCsr), C.POINTER(GnfMlp), C.POINTER(GnfGnnSpec),
                                    C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnf_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.POINTER(GnfFlow)]),
    "gnf_coupling_half_f32": (C.c_int, [C.POINTER(GnfCsr), C.POINTER(GnfMlp), C.POINTER(GnfMlp),
                                        C.POINTER(GnfGnnSpec), C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "gnf_grevnet_f32": (C.c_int, [C.POINTER(GnfCsr), C.POINTER(GnfFlow), C.c_void_p, C.c_int64,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnf_grevnet_from_f32": (C.c_int, [C.POINTER(GnfCsr), C.POINTER(GnfFlow), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnf_pred_adj_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnf_pred_adj_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnf_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.POINTER(GnfFlow)]),
    "gnf_grevnet_backward_f32": (C.c_int, [C.POINTER(GnfCsr), C.POINTER(GnfCsr), C.POINTER(GnfFlow),
                                           C.POINTER(GnfFlow), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                           C.c_size_t, C.c_void_p, C.c_void_p]),
    "gnf_pack_flow": (C.c_int, [C.POINTER(GnfFlow), C.c_void_p]),
    "gnf_bn_post_step_f32": (C.c_int, [C.POINTER(GnfFlow), C.c_int32, C.c_float, C.c_void_p]),
    "gnf_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "gnf_rccl_comm_create": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "gnf_rccl_comm_destroy": (C.c_int, [C.c_void_p]),
    "gnf_rccl_allreduce_sum_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gnf_adam_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_void_p]),
    "gnf_clip_by_value_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p]),
    "gnf_clip_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gnf_clip_by_norm_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gnf_gauss_sumsq_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load libgnf_hip.so once.  Raises GnfError (never falls back) if it is absent or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GnfError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.gnf_abi_version() != GNF_ABI_VERSION:
            raise GnfError(f"libgnf_hip.so ABI {handle.gnf_abi_version()} != binding {GNF_ABI_VERSION}")
        _lib = handle
        # developer convenience of THIS binding (the library itself never reads the environment):
        # GNF_OPTIONS="dw_grouped=1,force_shape=21" -> gnf_set_option calls, for tools/ and the launch-shape tests
        for item in filter(None, os.environ.get("GNF_OPTIONS", "").split(",")):
            name, _, val = item.partition("=")
            set_option(name.strip(), int(val or 1))
    return _lib


def set_option(name, value):
    """gnf_set_option: process-wide developer option of the library (0 = automatic)."""
    check(lib().gnf_set_option(name.encode(), int(value)), f"gnf_set_option({name})")


def get_option(name):
    return int(lib().gnf_get_option(name.encode()))


def check(rc, what):
    if rc != 0:
        msg = lib().gnf_last_error()
        raise GnfError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
