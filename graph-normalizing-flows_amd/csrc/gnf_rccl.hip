// The path's collectives without a host language in the launch path (ABI v9): a communicator created through RCCL
// directly and a ready-made GnfFlow.bn_allreduce hook on it.
//
// What the path exchanges: ONE all-reduce of [log_det_jacobian, sum z^2, num_nodes] (3 x fp64) per
// forward (the caller's, e.g. torch.distributed in bench.py), and - only with the batch-norm bijector and
// sync_batch_norm - an all-reduce of 2 H + 1 doubles per bijector call from INSIDE the flow's launch sequence
// (gnf_bn.hip, bn_sync_exchange).  That second one used to be reachable only through a Python callback
// (gnn.py::_bn_allreduce_hook -> torch.distributed).  gnf_rccl_allreduce_sum_f64 has the hook's signature and takes an
// ncclComm_t as its context: ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream), enqueued on the
// caller's stream like every kernel of the library.
//
// librccl.so is resolved with dlopen at first use (the library has no link-time dependency on it: single-GPU users
// never load it).  A process that already holds RCCL (PyTorch bundles its own copy) gets that copy: same soname.
// Reference counterpart: none (the reference is single-device).
#include <dlfcn.h>

#include <atomic>
#include <cstring>
#include <mutex>

#include "gnf_common.h"

namespace gnf {

// the few RCCL declarations used, restated (rccl.h: ncclResult_t ncclSuccess = 0, ncclDataType_t ncclFloat64 = 8,
// ncclRedOp_t ncclSum = 0, NCCL_UNIQUE_ID_BYTES = 128; the struct is passed BY VALUE to ncclCommInitRank)
struct NcclUniqueId {
    char internal[GNF_RCCL_UNIQUE_ID_BYTES];
};
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, NcclUniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st);
typedef const char* (*fn_get_error_string)(int);

struct RcclApi {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_get_error_string get_error_string = nullptr;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;
static char g_rccl_load_error[256] = "";

static char g_rccl_loaded_from[128] = "";

static void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // a copy the process already holds (PyTorch ships one) first: two RCCLs in one process would each own a set of
    // communicators and device buffers; only then load one, privately (RTLD_LOCAL: its symbols stay out of the global scope)
    char errs[200] = "";
    for (int pass = 0; pass < 2 && !g_rccl.handle; ++pass)
        for (const char* nm : names) {
            g_rccl.handle = dlopen(nm, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_LOCAL));
            if (g_rccl.handle) {
                snprintf(g_rccl_loaded_from, sizeof(g_rccl_loaded_from), "%s (%s)", nm, pass == 0 ? "already in the process" : "loaded here");
                break;
            }
            if (pass == 1) {
                const char* e = dlerror();
                const size_t used = strlen(errs);
                if (used + 4 < sizeof(errs)) snprintf(errs + used, sizeof(errs) - used, "%s%s", used ? "; " : "", e ? e : nm);
            }
        }
    if (!g_rccl.handle) {
        snprintf(g_rccl_load_error, sizeof(g_rccl_load_error), "librccl.so not found (dlopen: %s)", errs);
        return;
    }
    g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.handle, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.handle, "ncclCommInitRank");
    g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.handle, "ncclCommDestroy");
    g_rccl.all_reduce = (fn_all_reduce)dlsym(g_rccl.handle, "ncclAllReduce");
    g_rccl.get_error_string = (fn_get_error_string)dlsym(g_rccl.handle, "ncclGetErrorString");
    typedef int (*fn_get_version)(int*);
    const fn_get_version get_version = (fn_get_version)dlsym(g_rccl.handle, "ncclGetVersion");
    int version = 0;
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce || !get_version ||
        get_version(&version) != 0 || version < 20000) {  // (the restated declarations above are NCCL 2.x's)
        snprintf(g_rccl_load_error, sizeof(g_rccl_load_error),
                 "%s lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetVersion >= 2.0 (version %d)",
                 g_rccl_loaded_from, version);
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
    }
}

static const RcclApi* rccl(const char* what) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) {
        set_error("%s: %s", what, g_rccl_load_error);
        return nullptr;
    }
    return &g_rccl;
}

static int rccl_fail(const RcclApi* api, const char* what, int rc) {
    set_error("%s failed: %s (ncclResult_t %d)", what, api->get_error_string ? api->get_error_string(rc) : "?", rc);
    return GNF_EHIP;
}

}  // namespace gnf

using namespace gnf;

extern "C" {

int gnf_rccl_unique_id(char* id_out) {
    if (!id_out) {
        set_error("gnf_rccl_unique_id: null buffer (needs GNF_RCCL_UNIQUE_ID_BYTES bytes)");
        return GNF_EINVAL;
    }
    const RcclApi* api = rccl("gnf_rccl_unique_id");
    if (!api) return GNF_EINVAL;
    NcclUniqueId id;
    const int rc = api->get_unique_id(&id);
    if (rc) return rccl_fail(api, "ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, GNF_RCCL_UNIQUE_ID_BYTES);
    return GNF_OK;
}

int gnf_rccl_comm_create(const char* id, int32_t n_ranks, int32_t rank, void** comm_out) {
    if (!id || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        set_error("gnf_rccl_comm_create: id / comm_out null or rank %d outside [0, %d)", rank, n_ranks);
        return GNF_EINVAL;
    }
    const RcclApi* api = rccl("gnf_rccl_comm_create");
    if (!api) return GNF_EINVAL;
    NcclUniqueId uid;
    memcpy(uid.internal, id, GNF_RCCL_UNIQUE_ID_BYTES);
    void* comm = nullptr;
    const int rc = api->comm_init_rank(&comm, n_ranks, uid, rank);  // (binds to the CURRENT HIP device, like every launch here)
    if (rc) return rccl_fail(api, "ncclCommInitRank", rc);
    *comm_out = comm;
    return GNF_OK;
}

int gnf_rccl_comm_destroy(void* comm) {
    if (!comm) return GNF_OK;
    const RcclApi* api = rccl("gnf_rccl_comm_destroy");
    if (!api) return GNF_EINVAL;
    const int rc = api->comm_destroy(comm);
    return rc ? rccl_fail(api, "ncclCommDestroy", rc) : GNF_OK;
}

int gnf_rccl_allreduce_sum_f64(void* comm, double* device_buf, int64_t count, gnf_stream_t stream) {
    if (!comm || !device_buf || count < 0) {
        set_error("gnf_rccl_allreduce_sum_f64: null communicator / buffer");
        return GNF_EINVAL;
    }
    const RcclApi* api = rccl("gnf_rccl_allreduce_sum_f64");
    if (!api) return GNF_EINVAL;
    const int rc = api->all_reduce(device_buf, device_buf, (size_t)count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, (hipStream_t)stream);
    return rc ? rccl_fail(api, "ncclAllReduce", rc) : GNF_OK;
}

}  // extern "C"
