// Fused backward of one coupling half-step for gfx950 (MI355X), the training-side twin of gnf_fused.hip.
// ONE launch does, for a tile of 16 nodes and BOTH nets (waves 0-3 the s-net, waves 4-7 the t-net):
//   A   the same CSR aggregate + combine prologue as the forward kernel (h0 also goes to global memory)
//   B   recompute of the K layers on the exact-fp32 matrix cores; every hidden activation h_j is kept as a
//       byte sign mask in LDS (act' for the way back) and written to global memory (the dW GEMM's operand)
//   C'  the coupling update undone and differentiated in LDS / registers:
//         x_b = (y_b - t) exp(-s);  g_s = g_b (y_b - t) - 1;  g_t = g_b;  g_b <- g_b exp(s)
//   B'  the K layers backwards: dP_{j-1} = (dP_j W_j^T) * act'(h_j), streaming the TRANSPOSED packed
//       fragments (WpT_j, written next to Wp_j by the pack kernels); every dP_j also goes to global memory
// so a training half-step is this launch + one grouped dW GEMM + one reduce + the message-passing
// backward (gnf_train.hip), instead of 2K + 2K GEMM launches through global memory.
// The 2K "layers" (K forward rows then K backward rows) share the forward kernel's chunk loop
// (gnf_fused_dev.h): a backward row is a layer whose packed weights are WpT_j, whose bias is zero and
// whose activation is the mask multiply.
#include "gnf_fused_bwd_dev.h"

#include <string.h>

namespace gnf {

template <int MT, bool STASHED = false>
__global__ __launch_bounds__(kBwdThreads) void k_half_bwd_fused(const BwdArgs a) {
    half_bwd_body<MT, STASHED>(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
static inline int pad16b(int v) { return (v + 15) & ~15; }

static int max_padded_width_b(const GnfMlp* m) {
    int w = 16;
    for (int j = 0; j <= m->num_layers; ++j) w = w > pad16b(m->dims[j]) ? w : pad16b(m->dims[j]);
    return w;
}

static size_t bwd_lds_bytes(const GnfMlp* m, int MT) {
    const int K = m->num_layers;
    const int LS = max_padded_width_b(m) + 4;
    int bias_tot = 0, mld = 1;
    for (int j = 0; j < K; ++j) bias_tot += pad16b(m->dims[j + 1]);
    for (int j = 1; j < K; ++j) mld = mld > pad16b(m->dims[j]) / 16 ? mld : pad16b(m->dims[j]) / 16;
    const int bias_tot2 = bias_tot + max_padded_width_b(m);
    size_t floats = (size_t)(4 * 16 * MT * LS + 2 * bias_tot2);
    floats = (floats + 1) & ~(size_t)1;  // the 64-bit mask words start 8-byte aligned
    const int HP = pad16b(m->dims[K]);  // output width = H for the nets of a coupling
    return floats * sizeof(float) + (size_t)(kRows * 16 + kBwdRowptrPad + kBwdColCap) * sizeof(int) +
           (size_t)2 * (K > 1 ? K - 1 : 0) * (MT * 4) * mld * sizeof(unsigned long long) +
           // the folded message-passing backward of the previous half-step: transposed CSR slice + two [TM][HP] terms
           (size_t)(kBwdRowptrPad + kBwdColCap) * sizeof(int) + (size_t)2 * 16 * MT * HP * sizeof(float);
}

bool fused_bwd_fits_lds(const GnfMlp* m) { return bwd_lds_bytes(m, 1) <= (size_t)kBwdLdsLimit; }

bool fused_bwd_supported(const GnfMlp* s, const GnfMlp* t) {
    if (!s->packed || !t->packed) return false;
    if (s->attn && s->attn->layer_norm) return false;  // snt.LayerNorm after the MLP: the generic path differentiates it
    if (s->num_layers != t->num_layers) return false;
    for (int j = 0; j <= s->num_layers; ++j)
        if (s->dims[j] != t->dims[j]) return false;
    return bwd_lds_bytes(s, 1) <= (size_t)kBwdLdsLimit;
}

// grid and LDS bytes launch_half_bwd_fused will use for n nodes (the dW launch is sized around them)
void fused_bwd_launch_shape(const GnfMlp* s, int64_t n, int64_t* tiles, size_t* lds) {
    const int MT = ((n + 15) / 16 > 256 && bwd_lds_bytes(s, 2) <= (size_t)kBwdLdsLimit) ? 2 : 1;
    *tiles = (n + 16 * MT - 1) / (16 * MT);
    *lds = bwd_lds_bytes(s, MT);
}

// hin / dP: [net * K + j] global buffers the dW GEMM will read: hin[.][j] = input of layer j (j >= 1; h0 is
// shared), dP[.][j] = dL/d(pre-activation of layer j) for j <= K-2; dh0: [net] = dL/dh0.
int build_bwd_args(const int32_t* rowptr, const int32_t* col, int64_t n, const GnfGnnSpec& gnn, const GnfMlp* s,
                   const GnfMlp* t, const float* x_cond, float* y_upd, int64_t ld, float* g_upd, int64_t ldg, int32_t H,
                   float* h0_out, const float* const* h0_in, float* const* hin, int64_t ldh, float* const* dP,
                   int64_t lddp, float* const* gst, float* const* dh0, BwdArgs* out, int* mt, int64_t* tiles_out,
                   size_t* lds, const BwdFold* fold) {
    const int K = s->num_layers;
    BwdArgs& a = *out;
    memset(&a, 0, sizeof(a));
    a.rowptr = rowptr;
    a.col = col;
    a.x_cond = x_cond;
    a.y_upd = y_upd;
    a.g_upd = g_upd;
    a.h0_out = h0_out;
    a.h0_in[0] = h0_in ? h0_in[0] : nullptr;
    a.h0_in[1] = h0_in ? h0_in[1] : nullptr;
    a.residual = (s->attn && s->attn->residual) ? 1 : 0;
    a.gst[0] = gst[0];
    a.gst[1] = gst[1];
    a.ld = ld;
    a.ldg = ldg;
    if (fold) {
        a.rowptr_t = fold->rowptr_t;
        a.col_t = fold->col_t;
        a.invdeg = fold->invdeg;
        a.dh_prev[0] = fold->dh_prev[0];
        a.dh_prev[1] = fold->dh_prev[1];
        a.fold_concat = gnn.combine == GNF_COMBINE_CONCAT ? 1 : 0;
        a.fold_aggcol = a.fold_concat ? H : 0;
    }
    int bias_tot = 0, mld = 1;
    int64_t wtot = 0;
    for (int j = 0; j < K; ++j) {
        bias_tot += pad16b(s->dims[j + 1]);
        wtot += (int64_t)pad16b(s->dims[j]) * pad16b(s->dims[j + 1]);
    }
    for (int j = 1; j < K; ++j) mld = mld > pad16b(s->dims[j]) / 16 ? mld : pad16b(s->dims[j]) / 16;
    a.bias[0] = s->packed + wtot;
    a.bias[1] = t->packed + wtot;
    a.bias_tot = bias_tot;
    a.bias_tot2 = bias_tot + max_padded_width_b(s);
    a.mld = mld;
    const GnfMlp* nets[2] = {s, t};
    auto put_ptr = [&](int r, int slot, const void* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        a.tab[r][slot] = (int)(unsigned)v;
        a.tab[r][slot + 1] = (int)(unsigned)(v >> 32);
    };
    int64_t off = 0;
    int boff = 0;
    for (int j = 0; j < K; ++j) {  // forward rows
        const int ip = pad16b(s->dims[j]), op = pad16b(s->dims[j + 1]);
        int* row = a.tab[j];
        row[0] = ip / 16;
        row[1] = op / 16;
        row[2] = boff;
        row[3] = s->dims[j + 1];
        row[4] = 0;
        row[5] = j < K - 1 ? j : -1;  // mask slot j = sign of h_{j+1}
        row[6] = (int)ldh;
        for (int q = 0; q < 2; ++q) {
            put_ptr(j, 8 + 2 * q, nets[q]->packed + off);
            put_ptr(j, 12 + 2 * q, j < K - 1 ? hin[q * K + j + 1] : nullptr);
        }
        // backward row of the same layer: r = K + (K-1-j)
        const int r = K + (K - 1 - j);
        int* brow = a.tab[r];
        brow[0] = op / 16;
        brow[1] = ip / 16;
        brow[2] = bias_tot;  // zero bias
        brow[3] = s->dims[j];
        brow[4] = 1;
        brow[5] = j >= 1 ? j - 1 : -1;  // act' of h_j
        brow[6] = j >= 1 ? (int)lddp : s->dims[0];
        for (int q = 0; q < 2; ++q) {
            put_ptr(r, 8 + 2 * q, nets[q]->packed + wtot + bias_tot + off);
            put_ptr(r, 12 + 2 * q, j >= 1 ? dP[q * K + j - 1] : dh0[q]);
        }
        boff += op;
        off += (int64_t)ip * op;
    }
    a.n_nodes = (int32_t)n;
    a.H = H;
    a.in0 = s->dims[0];
    a.K = K;
    a.n_rows = 2 * K;
    a.LS = max_padded_width_b(s) + 4;
    a.mean = gnn.agg == GNF_AGG_MEAN;
    a.concat = gnn.combine == GNF_COMBINE_CONCAT;
    a.act = gnn.activation;
    a.eps = gnn.epsilon;
    a.alpha = gnn.alpha;
    // 32 nodes per workgroup once there is more than one 16-node tile per CU (and the LDS budget allows)
    const int MT = ((n + 15) / 16 > 256 && bwd_lds_bytes(s, 2) <= (size_t)kBwdLdsLimit) ? 2 : 1;
    const int64_t tiles = (n + 16 * MT - 1) / (16 * MT);
    a.n_tiles = (int32_t)tiles;
    *mt = MT;
    *tiles_out = tiles;
    *lds = bwd_lds_bytes(s, MT);
    return GNF_OK;
}

int launch_half_bwd_fused(const int32_t* rowptr, const int32_t* col, int64_t n, const GnfGnnSpec& gnn,
                          const GnfMlp* s, const GnfMlp* t, const float* x_cond, float* y_upd, int64_t ld,
                          float* g_upd, int64_t ldg, int32_t H, float* h0_out, const float* const* h0_in, float* const* hin,
                          int64_t ldh, float* const* dP, int64_t lddp, float* const* gst, float* const* dh0,
                          hipStream_t st) {
    if (n == 0) return GNF_OK;
    BwdArgs a;
    int MT;
    int64_t tiles;
    size_t lds;
    const int rc = build_bwd_args(rowptr, col, n, gnn, s, t, x_cond, y_upd, ld, g_upd, ldg, H, h0_out, h0_in, hin, ldh, dP,
                                  lddp, gst, dh0, &a, &MT, &tiles, &lds);
    if (rc) return rc;
    GNF_ONCE_PER_DEVICE(
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_bwd_fused<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLdsLimit));
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_bwd_fused<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLdsLimit)));
    if (MT == 2)
        hipLaunchKernelGGL(k_half_bwd_fused<2>, dim3((unsigned)tiles), dim3(kBwdThreads), lds, st, a);
    else
        hipLaunchKernelGGL(k_half_bwd_fused<1>, dim3((unsigned)tiles), dim3(kBwdThreads), lds, st, a);
    GNF_LAUNCH_CHECK("k_half_bwd_fused");
    return GNF_OK;
}

bool bwd_args_add_dagg_row(BwdArgs* a, const float* const* wot, float* const* dagg, int C, int NV, int off) {
    const int r = 2 * a->K;
    const int Cp = pad16b(C), NVp = pad16b(NV);
    if (a->n_rows != r || r + 1 > kRows || (off & 3) || off + Cp > a->LS - 4 || NVp > a->LS - 4 || NVp > a->bias_tot2 - a->bias_tot) return false;
    int* row = a->tab[r];
    row[0] = Cp / 16;
    row[1] = NVp / 16;
    row[2] = a->bias_tot;  // zero bias
    row[3] = NV;
    row[4] = 1;
    row[5] = -1;
    row[6] = NV;
    row[7] = off;
    for (int q = 0; q < 2; ++q) {
        const unsigned long long w = reinterpret_cast<unsigned long long>(wot[q]), d = reinterpret_cast<unsigned long long>(dagg[q]);
        row[8 + 2 * q] = (int)(unsigned)w, row[9 + 2 * q] = (int)(unsigned)(w >> 32);
        row[12 + 2 * q] = (int)(unsigned)d, row[13 + 2 * q] = (int)(unsigned)(d >> 32);
    }
    a->n_rows = r + 1;
    return true;
}

// the same launch for arguments that carry a half-step's stash rows (st_in, mask_in set by the caller): 16-node tiles only
int launch_half_bwd_fused_stashed(const BwdArgs& a, int mt, int64_t tiles, size_t lds, hipStream_t st) {
    if (mt != 1) {
        set_error("internal: MLP-row stash with 32-node backward tiles (mlp_stash_supported is false there)");
        return GNF_EINVAL;
    }
    GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_bwd_fused<1, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLdsLimit)));
    hipLaunchKernelGGL((k_half_bwd_fused<1, true>), dim3((unsigned)tiles), dim3(kBwdThreads), lds, st, a);
    GNF_LAUNCH_CHECK("k_half_bwd_fused (stash)");
    return GNF_OK;
}

}  // namespace gnf
