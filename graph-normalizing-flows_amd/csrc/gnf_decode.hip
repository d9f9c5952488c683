// Decoder that follows the sampling pass: pred_adj(graph, scaled_hacky_sigmoid_l2)
// (/root/reference/loss.py:45-53 distance, 131-151 block-diagonal mask, 154-159 pred_adj; called at
// train_grevnet_with_data.py:415-416, thresholded at 0.5 at :532-533):
//   P[i, j] = sigmoid(10 * (1 - ||z_i - z_j||^2 / sqrt(D)))   for i != j in the SAME graph, 0 otherwise.
// The reference materialises a dense [N, N] matrix and multiplies by a block-diagonal mask; here only
// the per-graph [n_g, n_g] blocks exist (concatenated, block g at blk_off[g]), which is also what the
// consumer slices out (train_grevnet_with_data.py:538-540).  One workgroup per (graph, 16-row tile):
// the tile's rows sit in LDS, every lane walks the columns j of its graph (rows z_j are coalesced reads).  Rows wider
// than the LDS tile holds (D > 1024) are read from global memory instead - same order of additions, no bound on D.
#include "gnf_common.h"

namespace gnf {

static constexpr int kDecTile = 16;

__global__ __launch_bounds__(256) void k_adj_offsets(const int32_t* __restrict__ n_node, int64_t n_graphs,
                                                     int64_t* __restrict__ node_off,
                                                     int64_t* __restrict__ blk_off) {
    __shared__ int64_t shn[257], shb[257];
    const int64_t chunk = (n_graphs + 255) / 256;
    const int64_t beg = (int64_t)threadIdx.x * chunk;
    int64_t end = beg + chunk;
    if (end > n_graphs) end = n_graphs;
    int64_t ln = 0, lb = 0;
    for (int64_t i = beg; i < end; ++i) {
        ln += n_node[i];
        lb += (int64_t)n_node[i] * n_node[i];
    }
    shn[threadIdx.x + 1] = ln;
    shb[threadIdx.x + 1] = lb;
    if (threadIdx.x == 0) shn[0] = shb[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i <= 256; ++i) {
            shn[i] += shn[i - 1];
            shb[i] += shb[i - 1];
        }
    __syncthreads();
    int64_t rn = shn[threadIdx.x], rb = shb[threadIdx.x];
    if (threadIdx.x == 0) node_off[0] = blk_off[0] = 0;
    for (int64_t i = beg; i < end; ++i) {
        rn += n_node[i];
        rb += (int64_t)n_node[i] * n_node[i];
        node_off[i + 1] = rn;
        blk_off[i + 1] = rb;
    }
}

template <bool LDS_ROWS>
__global__ __launch_bounds__(256) void k_pred_adj(const float* __restrict__ z, int64_t ld, int D,
                                                  const int64_t* __restrict__ node_off,
                                                  const int64_t* __restrict__ blk_off,
                                                  float* __restrict__ out, float inv_sqrt_d) {
    extern __shared__ float zi[];  // [kDecTile][D]
    const int g = blockIdx.x;
    const int64_t n0 = node_off[g];
    const int ng = (int)(node_off[g + 1] - n0);
    const int i0 = blockIdx.y * kDecTile;
    if (i0 >= ng) return;
    const int rows = ng - i0 < kDecTile ? ng - i0 : kDecTile;
    if constexpr (LDS_ROWS) {
        for (int i = threadIdx.x; i < rows * D; i += 256) {
            const int rl = i / D, f = i - rl * D;
            zi[i] = z[(n0 + i0 + rl) * ld + f];
        }
        __syncthreads();
    }
    float* blk = out + blk_off[g];
    for (int idx = threadIdx.x; idx < rows * ng; idx += 256) {
        const int rl = idx / ng, j = idx - rl * ng;
        const float* zj = z + (n0 + j) * ld;
        const float* zr = LDS_ROWS ? zi + rl * D : z + (n0 + i0 + rl) * ld;
        float d2 = 0.f;
        for (int f = 0; f < D; ++f) {
            const float df = zr[f] - zj[f];
            d2 = fmaf(df, df, d2);
        }
        const float a = 10.f * (1.f - d2 * inv_sqrt_d);
        const float p = 1.f / (1.f + expf(-a));
        blk[(int64_t)(i0 + rl) * ng + j] = (i0 + rl == j) ? 0.f : p;  // remove_diag (loss.py:157)
    }
}

}  // namespace gnf

using namespace gnf;

extern "C" {

size_t gnf_pred_adj_workspace_bytes(int64_t n_graphs) {
    if (n_graphs < 0) return 0;
    return (size_t)(2 * (n_graphs + 1)) * sizeof(int64_t);
}

int gnf_pred_adj_f32(const float* z, int64_t ld, int32_t D, const int32_t* n_node, int64_t n_graphs,
                     int32_t max_nodes_per_graph, float* out_blocks, int64_t* block_off, void* ws,
                     size_t ws_bytes, gnf_stream_t stream) {
    if (n_graphs < 0 || D < 1 || ld < D || max_nodes_per_graph < 0) {
        set_error("gnf_pred_adj_f32: n_graphs=%lld D=%d ld=%lld max_nodes=%d", (long long)n_graphs, D,
                  (long long)ld, max_nodes_per_graph);
        return GNF_ESHAPE;
    }
    if (!ws || !block_off || (n_graphs > 0 && (!n_node || !z || !out_blocks))) {
        set_error("gnf_pred_adj_f32: null pointer argument");
        return GNF_EINVAL;
    }
    if (ws_bytes < gnf_pred_adj_workspace_bytes(n_graphs)) {
        set_error("gnf_pred_adj_f32: workspace %zu < %zu bytes", ws_bytes, gnf_pred_adj_workspace_bytes(n_graphs));
        return GNF_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int64_t* node_off = (int64_t*)ws;
    hipLaunchKernelGGL(k_adj_offsets, dim3(1), dim3(256), 0, st, n_node, n_graphs, node_off, block_off);
    GNF_LAUNCH_CHECK("k_adj_offsets");
    if (n_graphs == 0 || max_nodes_per_graph == 0) return GNF_OK;
    const unsigned tiles = (unsigned)((max_nodes_per_graph + kDecTile - 1) / kDecTile);
    const size_t row_tile = (size_t)kDecTile * D * sizeof(float);
    if (row_tile <= 64 * 1024)
        hipLaunchKernelGGL(k_pred_adj<true>, dim3((unsigned)n_graphs, tiles), dim3(256), row_tile, st, z, ld, D, node_off,
                           block_off, out_blocks, 1.f / sqrtf((float)D));
    else
        hipLaunchKernelGGL(k_pred_adj<false>, dim3((unsigned)n_graphs, tiles), dim3(256), 0, st, z, ld, D, node_off,
                           block_off, out_blocks, 1.f / sqrtf((float)D));
    GNF_LAUNCH_CHECK("k_pred_adj");
    return GNF_OK;
}

}  // extern "C"
