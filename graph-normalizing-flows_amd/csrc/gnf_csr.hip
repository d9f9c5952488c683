// Device-side construction of the receiver-sorted CSR from a GraphsTuple edge list
// (senders/receivers with global node ids + per-graph n_node/n_edge; the container the reference
// builds at train_grevnet_with_data.py:265-271 and graph_data.py:122).
//
// Batched graphs are block-diagonal: every edge of graph g lies in one contiguous slice of the edge
// list and touches only g's nodes.  One workgroup per graph therefore does everything locally, in LDS, in ONE launch:
//   offsets : sums of n_node / n_edge over the graphs in front of it        -> n0, e0
//   count   : edge-parallel LDS-atomic histogram of the receivers
//   scan    : exclusive scan of the counts                                  -> rowptr[n0 + i] = e0 + prefix
//   place   : edge-parallel placement with an LDS-atomic cursor per receiver (order inside a row is
//             arbitrary at this point)
//   rank    : one wave per row counts, per entry, the row's smaller edge ids: its slot in EDGE ORDER.  Sorting by edge
//             index makes the result identical to a stable sort by receiver, i.e. neighbours appear in original edge
//             order, which fixes the fp32 summation order of the aggregation
//   gather  : col[] = senders[edge], every entry of the graph in one sweep.
// Graphs too large for the LDS budget (more than kCapE edges or kCapN nodes) take a slower path inside the same
// workgroup (global counts, one thread per receiver walks the edge slice in order).
#include "gnf_common.h"

namespace gnf {

static constexpr int kCsrBlock = 256;
static constexpr int kEdgeTile = 2048;
static constexpr int kCapE = 16384;  // edges per graph handled in LDS (64 KB of edge ids)
static constexpr int kCapN = 2048;   // nodes per graph handled in LDS

// ---- one workgroup per graph: offsets, count, scan, placement, rank, gather ---------------------------------------
// Everything a graph's rows need is local to the graph: rowptr[n0 + i] = e0 + (edges of the graph with a receiver
// below i), because the graphs before it own exactly the e0 edges in front of its slice - no scan over the batch's nodes.
// (Round 4: this kernel replaces five launches - graph offsets, zero, count, a one-workgroup scan over all nodes, fill -
// whose fill also paid a global round trip per ROW for the senders: 46 us of launches on the config-2 batch.)
__global__ __launch_bounds__(kCsrBlock) void k_csr_graph(const int32_t* __restrict__ senders,
                                                         const int32_t* __restrict__ receivers,
                                                         const int32_t* __restrict__ n_node,
                                                         const int32_t* __restrict__ n_edge, int64_t n_nodes,
                                                         int32_t* __restrict__ deg_ws,
                                                         int32_t* __restrict__ rowptr, int32_t* __restrict__ col) {
    extern __shared__ __attribute__((aligned(16))) int32_t sm[];
    __shared__ int32_t red[2][kCsrBlock / 64];
    __shared__ int32_t scan_sh[kCsrBlock + 1];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // this graph's first node / edge: sums over the graphs in front of it
    int sn = 0, se = 0;
    for (int j = tid; j < g; j += kCsrBlock) sn += n_node[j], se += n_edge[j];
    for (int off = 32; off > 0; off >>= 1) sn += __shfl_down(sn, off, 64), se += __shfl_down(se, off, 64);
    if (lane == 0) red[0][wave] = sn, red[1][wave] = se;
    __syncthreads();
    int n0 = 0, e0 = 0;
    for (int w = 0; w < kCsrBlock / 64; ++w) n0 += red[0][w], e0 += red[1][w];
    const int ng = n_node[g], eg = n_edge[g];
    if (g == 0 && tid == 0) rowptr[0] = 0;
    if (ng <= kCapN && eg <= kCapE) {
        int32_t* rp = sm;                  // [ng + 1] row starts relative to the graph's first edge
        int32_t* cur = sm + kCapN + 1;     // [ng] counts, then placement cursors
        int32_t* eid = cur + kCapN;        // [eg] edge ids (relative to e0), grouped by receiver, any order inside a row
        int32_t* srt = eid + kCapE;        // [eg] the same in edge order
        for (int i = tid; i < ng; i += kCsrBlock) cur[i] = 0;
        __syncthreads();
        for (int eb = 0; eb < eg; eb += 8 * kCsrBlock) {  // eight receiver loads in flight per thread, then the atomics
            int rr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = eb + u * kCsrBlock + tid;
                rr[u] = e < eg ? receivers[e0 + e] - n0 : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (rr[u] >= 0) atomicAdd(&cur[rr[u]], 1);
        }
        __syncthreads();
        {   // exclusive scan of the counts (each thread a contiguous run of rows), rowptr, cursors back to zero
            const int chunk = (ng + kCsrBlock - 1) / kCsrBlock;
            const int beg = tid * chunk, end = beg + chunk < ng ? beg + chunk : ng;
            int local = 0;
            for (int i = beg; i < end; ++i) local += cur[i];
            scan_sh[tid + 1] = local;
            if (tid == 0) scan_sh[0] = 0;
            __syncthreads();
            if (tid == 0)
                for (int i = 1; i <= kCsrBlock; ++i) scan_sh[i] += scan_sh[i - 1];
            __syncthreads();
            int run = scan_sh[tid];
            for (int i = beg; i < end; ++i) {
                rp[i] = run;
                rowptr[n0 + i] = e0 + run;
                run += cur[i];
                cur[i] = 0;
            }
            if (tid == 0) rp[ng] = eg, rowptr[n0 + ng] = e0 + eg;  // (= the next graph's first entry: the same value)
        }
        __syncthreads();
        for (int eb = 0; eb < eg; eb += 8 * kCsrBlock) {
            int rr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = eb + u * kCsrBlock + tid;
                rr[u] = e < eg ? receivers[e0 + e] - n0 : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (rr[u] >= 0) {
                    const int pos = atomicAdd(&cur[rr[u]], 1);
                    eid[rp[rr[u]] + pos] = eb + u * kCsrBlock + tid;
                }
        }
        __syncthreads();
        // the atomics above put a row's edges in any order: restore edge order by RANK - one wave per row, a lane per
        // entry counts the row's smaller edge ids (every lane reads the same LDS word: a broadcast); sorting by edge index
        // makes the result identical to a stable sort by receiver
        for (int i = wave; i < ng; i += kCsrBlock / 64) {
            const int beg = rp[i], d = rp[i + 1] - beg;
            for (int a = lane; a < d; a += 64) {
                const int key = eid[beg + a];
                int rank = 0;
                int b = 0;
                for (; b + 16 <= d; b += 16) {  // sixteen LDS reads in flight
                    int v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = eid[beg + b + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) rank += v[u] < key ? 1 : 0;
                }
                for (; b < d; ++b) rank += eid[beg + b] < key ? 1 : 0;
                srt[beg + rank] = key;
            }
        }
        __syncthreads();
        for (int j = tid; j < eg; j += kCsrBlock) col[e0 + j] = senders[e0 + srt[j]];  // every sender in one sweep
    } else {
        // a graph too large for LDS (rare): counts through global atomics into the workspace, a scan over its rows, then
        // one thread per receiver walks the graph's edge slice in order (tiles staged in LDS) - all by this workgroup
        int32_t* deg = deg_ws + n0;
        for (int i = tid; i < ng; i += kCsrBlock) deg[i] = 0;
        __threadfence_block();
        __syncthreads();
        for (int e = tid; e < eg; e += kCsrBlock) atomicAdd(&deg[receivers[e0 + e] - n0], 1);
        __threadfence_block();
        __syncthreads();
        {
            const int chunk = (ng + kCsrBlock - 1) / kCsrBlock;
            const int beg = tid * chunk, end = beg + chunk < ng ? beg + chunk : ng;
            int local = 0;
            for (int i = beg; i < end; ++i) local += __hip_atomic_load(&deg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            scan_sh[tid + 1] = local;
            if (tid == 0) scan_sh[0] = 0;
            __syncthreads();
            if (tid == 0)
                for (int i = 1; i <= kCsrBlock; ++i) scan_sh[i] += scan_sh[i - 1];
            __syncthreads();
            int run = scan_sh[tid];
            for (int i = beg; i < end; ++i) {
                rowptr[n0 + i] = e0 + run;
                run += __hip_atomic_load(&deg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) rowptr[n0 + ng] = e0 + eg;
        }
        __threadfence_block();
        __syncthreads();
        int32_t* s_recv = sm;
        int32_t* s_send = sm + kEdgeTile;
        for (int nb = 0; nb < ng; nb += kCsrBlock) {
            const int node = nb + tid;
            const bool live = node < ng;
            const int cursor = live ? rowptr[n0 + node] : 0;
            int count = 0;
            for (int eb = 0; eb < eg; eb += kEdgeTile) {
                const int m = (eg - eb) < kEdgeTile ? (eg - eb) : kEdgeTile;
                __syncthreads();
                for (int i = tid; i < m; i += kCsrBlock) {
                    s_recv[i] = receivers[e0 + eb + i] - n0;
                    s_send[i] = senders[e0 + eb + i];
                }
                __syncthreads();
                if (live)
                    for (int i = 0; i < m; ++i)
                        if (s_recv[i] == node) col[cursor + count++] = s_send[i];
            }
        }
    }
    (void)n_nodes;
}

static constexpr size_t kFillLds = (size_t)(2 * kCapN + 1 + 2 * kCapE) * sizeof(int32_t);  // 147.5 KB: one workgroup per CU

}  // namespace gnf

using namespace gnf;

extern "C" {

size_t gnf_csr_workspace_bytes(int64_t n_graphs, int64_t n_nodes) {
    if (n_graphs < 0 || n_nodes < 0) return 0;
    return (size_t)(2 * (n_graphs + 1) + n_nodes + 1) * sizeof(int32_t);
}

int gnf_build_csr(const int32_t* senders, const int32_t* receivers, const int32_t* n_node,
                  const int32_t* n_edge, int64_t n_graphs, int64_t n_nodes, int64_t n_edges,
                  int32_t* rowptr, int32_t* col, void* ws, size_t ws_bytes, gnf_stream_t stream) {
    if (n_graphs < 0 || n_nodes < 0 || n_edges < 0 || n_nodes > INT32_MAX || n_edges > INT32_MAX) {
        set_error("gnf_build_csr: n_graphs=%lld n_nodes=%lld n_edges=%lld", (long long)n_graphs,
                  (long long)n_nodes, (long long)n_edges);
        return GNF_ESHAPE;
    }
    if (!rowptr || !ws || (n_graphs > 0 && (!n_node || !n_edge)) ||
        (n_edges > 0 && (!senders || !receivers || !col))) {
        set_error("gnf_build_csr: null pointer argument");
        return GNF_EINVAL;
    }
    if (ws_bytes < gnf_csr_workspace_bytes(n_graphs, n_nodes)) {
        set_error("gnf_build_csr: workspace %zu < %zu bytes", ws_bytes,
                  gnf_csr_workspace_bytes(n_graphs, n_nodes));
        return GNF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int32_t* deg = (int32_t*)ws + 2 * (n_graphs + 1);  // (only graphs too large for LDS count through it)
    if (n_graphs == 0 || n_nodes == 0) {  // no rows: rowptr = [0 .. 0]
        GNF_HIP_TRY(hipMemsetAsync(rowptr, 0, (size_t)(n_nodes + 1) * sizeof(int32_t), st));
        return GNF_OK;
    }
    GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_csr_graph),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFillLds)));
    hipLaunchKernelGGL(k_csr_graph, dim3((unsigned)n_graphs), dim3(kCsrBlock), kFillLds, st, senders, receivers, n_node, n_edge,
                       n_nodes, deg, rowptr, col);
    GNF_LAUNCH_CHECK("k_csr_graph");
    return GNF_OK;
}

}  // extern "C"
