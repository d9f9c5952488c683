// One-launch attention front-end for sparse batches (mean in-degree below ~24): everything DMSelfAttentionMLP._build
// (/root/reference/gnn.py:503-545) does before its MLP, for BOTH nets of a coupling half-step, per tile of 16
// receiver rows, entirely on the fp32 matrix cores (a block-sparse "flash attention" over the tile's incoming edges):
//   k[r]     = x[r] Wk                      own rows           [16 x H] x [H x nq]    (the module's receiver "queries")
//   q|v[s]   = x[s] [Wq | Wv]               every node s of the tile's SENDER WINDOW (the node range its incoming edges
//                                            point into: the graphs the tile's rows belong to), 128 nodes at a time
//   per (row r, head h): one thread walks the row's edges with the online-softmax recurrence in registers
//     (logit = <q[e,h,:], k[r,h,:]> * scale, running max / denominator / weighted sum of v[e,:]), reading the
//     projected q | v rows from LDS four edges at a time                                              gnn.py:458-475
//     [measured and dropped: the same as dense masked [64 x 16] tiles on the matrix cores (S = Q K^T, P V with the S
//      accumulator as A operand): 16 x redundant pairs + cross-lane reductions, 11.5k cycles per 64-edge chunk
//      against ~1.5k for this form - profiles/r2l_attn_front_trace_mfma_attention.txt]
//   new[r]   = (O[r,h,:] / den[r,h])_h Wo   [16 x heads*v] x [heads*v x C]                                    gnn.py:533-540
//   h0[r]    = [x[r] || new[r]]  or  new[r]                                                                   gnn.py:542-543
// The two-launch path (k_attn_proj + k_attn_agg, gnf_attn.hip) writes q|k|v of every node to global memory, reads the
// senders' rows back per tile and does logits / softmax / weighted sums as scalar LDS loops (43 us per half-step on the
// config-2 batch); here a tile projects its sender window itself (a node is projected once per tile whose window holds
// it: ~2-3 x the minimum on that batch; projecting per EDGE was 4 x and MFMA-bound, profiles/r2n_*) and nothing but h0
// goes back to global memory.  Dense batches (complete graphs) keep the per-node projection + the row-window kernels.
// Weights come pre-packed in MFMA fragment order (k_attn_pack, once per flow call into the caller's workspace): a B
// fragment is one coalesced 16-byte load per lane, exactly like the MLP weights of the fused half-step kernel.
// When a.qkv != NULL (training forward: GnfFlow.attn_stash) the tile also writes q|k|v of its OWN rows, in the
// layout and with the arithmetic (same MFMA k-order) the backward kernels expect.
#include "gnf_attn_front_dev.h"

namespace gnf {

size_t attn_pack_floats(const GnfAttn* at, int32_t H) {
    if (!at) return 0;
    return front_pack_floats(front_dims(H, at->num_heads, at->kq_dim, at->v_dim, at->out_dim));
}

struct PackArgs {
    const float* Wq;
    const float* Wk;
    const float* Wv;
    const float* Wo;
    float* out;
};
static constexpr int kPackMaxNets = 64;  // per launch (a flow has 4 T nets: several launches when T > 16)
struct PackBatch {
    PackArgs net[kPackMaxNets];
    FrontDims d;
};

__global__ __launch_bounds__(256) void k_attn_pack(const PackBatch b) {
    const PackArgs p = b.net[blockIdx.y];
    const FrontDims d = b.d;
    const int64_t n_qv = (int64_t)d.Hp * d.PW, n_k = (int64_t)d.Hp * d.nqp, n_o = (int64_t)d.NVp * d.Cp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_qv + n_k + n_o; i += (int64_t)gridDim.x * 256) {
        int64_t j = i;
        int nts, K, which;
        if (j < n_qv)
            which = 0, nts = d.PW >> 4, K = d.H;
        else if (j < n_qv + n_k)
            which = 1, j -= n_qv, nts = d.nqp >> 4, K = d.H;
        else
            which = 2, j -= n_qv + n_k, nts = d.Cp >> 4, K = d.NV;
        const int q = (int)(j & 3), ln = (int)((j >> 2) & 63);
        const int64_t blk = j >> 8;
        const int kg = (int)(blk / nts), nt = (int)(blk - (int64_t)kg * nts);
        const int k = 16 * kg + 4 * (ln >> 4) + q, c = 16 * nt + (ln & 15);
        float w = 0.f;
        if (k < K) {
            if (which == 0) {
                if (c < d.nq)
                    w = p.Wq[(int64_t)k * d.nq + c];
                else if (c >= d.nqp && c - d.nqp < d.vd)
                    w = p.Wv[(int64_t)k * d.vd + (c - d.nqp)];
            } else if (which == 1) {
                if (c < d.nq) w = p.Wk[(int64_t)k * d.nq + c];
            } else {
                if (c < d.C) w = p.Wo[(int64_t)k * d.C + c];
            }
        }
        p.out[i] = w;
    }
}

// packs `count` attention blocks (same hyper-parameters) into consecutive slots of `out` (attn_pack_floats each)
int launch_attn_pack(const GnfAttn* const* at, int count, int32_t H, float* out, hipStream_t st) {
    if (count <= 0) return GNF_OK;
    const FrontDims d = front_dims(H, at[0]->num_heads, at[0]->kq_dim, at[0]->v_dim, at[0]->out_dim);
    const size_t per = front_pack_floats(d);
    for (int base = 0; base < count; base += kPackMaxNets) {
        PackBatch b;
        b.d = d;
        const int nb = count - base < kPackMaxNets ? count - base : kPackMaxNets;
        for (int q = 0; q < nb; ++q)
            b.net[q] = PackArgs{at[base + q]->Wq, at[base + q]->Wk, at[base + q]->Wv, at[base + q]->Wo,
                                out + (size_t)(base + q) * per};
        unsigned bx = (unsigned)((per + 255) / 256);
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(k_attn_pack, dim3(bx, (unsigned)nb), dim3(256), 0, st, b);
        GNF_LAUNCH_CHECK("k_attn_pack");
    }
    return GNF_OK;
}

// Sender window (lo, hi) of every 16-row tile: min / max of the tile's slice of col; an empty tile gets (0x7fffffff, -1).
// One wave per tile.  Launched once per flow call (FrontArgs.tiles): the topology is the same for all 2 T half-steps.
__global__ __launch_bounds__(256) void k_attn_tiles(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int n_nodes,
                                                    int n_tiles, int32_t* __restrict__ out) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    const int r0 = tile * kFrRows, r1 = r0 + kFrRows < n_nodes ? r0 + kFrRows : n_nodes;
    const int e0 = rowptr[r0], e1 = rowptr[r1];
    int lo = 0x7fffffff, hi = -1;
    for (int base = e0; base < e1; base += 64 * 8) {
        int reg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + lane + 64 * u;
            reg[u] = col[e < e1 ? e : e1 - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            lo = reg[u] < lo ? reg[u] : lo;
            hi = reg[u] > hi ? reg[u] : hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) out[2 * tile] = lo, out[2 * tile + 1] = hi;
}

int launch_attn_tiles(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t* out, hipStream_t st) {
    if (n <= 0) return GNF_OK;
    const int n_tiles = (int)((n + kFrRows - 1) / kFrRows);
    hipLaunchKernelGGL(k_attn_tiles, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, st, rowptr, col, (int)n, n_tiles, out);
    GNF_LAUNCH_CHECK("k_attn_tiles");
    return GNF_OK;
}

// ------------------------------------------------------------------------------------------------------------------
template <bool HOIST, int KQM, int VDM, int EU, bool EXACT, bool FIXED = false>
__global__ __launch_bounds__(kFrThreads) void k_attn_front(const FrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    attn_front_tile<HOIST, KQM, VDM, EU, EXACT, false, FIXED>(a, lds, (int)blockIdx.x * kFrRows, nullptr, nullptr, 0, [] {});
}

bool attn_front_fixed_geometry(const FrontDims& d) { return d.H == 32 && d.nh == 8 && d.kq == 10 && d.vd == 10 && d.C == 80; }

// does the one-launch front-end handle this head geometry? (else: the two-launch kernels)
bool attn_front_fused_ok(const GnfAttn* at, int32_t H) {
    if (!at) return false;
    const FrontDims d = front_dims(H, at->num_heads, at->kq_dim, at->v_dim, at->out_dim);
    if (kFrRows * d.nh > 256 || d.vd > 16 || d.kq > 16 || d.Hp > 128 || d.NVp > 256) return false;
    return (size_t)front_lds(d).total * sizeof(float) <= (size_t)160 * 1024;
}

// packed[q]: the net's fragments (launch_attn_pack).  qkv_out may be NULL.
int launch_attn_front_fused(const int32_t* rowptr, const int32_t* col, int64_t n, const float* x, int64_t ldx, int32_t H,
                            const GnfAttn* const* at, int nets, int32_t in0, const float* const* packed,
                            float* const* qkv_out, float* const* h0_out, hipStream_t st, float* const* agg_out,
                            float* const* mz_out) {
    const GnfAttn* a0 = at[0];
    FrontArgs a;
    a.d = front_dims(H, a0->num_heads, a0->kq_dim, a0->v_dim, a0->out_dim);
    const FrontLds L = front_lds(a.d);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    for (int q = 0; q < 2; ++q) {
        const int s = q < nets ? q : 0;
        a.packed[q] = packed[s];
        a.qkv[q] = qkv_out ? qkv_out[s] : nullptr;
        a.h0[q] = h0_out[s];
        a.agg_out[q] = agg_out ? agg_out[s] : nullptr;
        a.mz_out[q] = mz_out ? mz_out[s] : nullptr;
    }
    a.rowptr = rowptr, a.col = col, a.x = x, a.ldx = ldx;
    a.tiles = nullptr;   // (the standalone launches scan their col slice themselves)
    a.bn_part = nullptr;
    a.n_nodes = (int32_t)n;
    a.concat = a0->concat ? 1 : 0;
    a.in0 = in0;
    a.scale = a0->kq_dim_division ? 1.f / sqrtf((float)a0->kq_dim) : 1.f;
    const dim3 grid((unsigned)((n + kFrRows - 1) / kFrRows));
    const bool hoist = (a.d.PW >> 4) <= 6 && (a.d.Hp >> 4) <= 2;
    if (attn_front_fixed_geometry(a.d)) {  // the drivers' defaults at D = 64: every width a compile-time constant
        GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_front<true, 10, 10, 4, true, true>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        hipLaunchKernelGGL((k_attn_front<true, 10, 10, 4, true, true>), grid, dim3(kFrThreads), lds_bytes, st, a);
    } else if (hoist && a.d.kq == 10 && a.d.vd == 10) {  // the reference's head geometry (run_grevnet.py:74-76) at H <= 32
        GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_front<true, 10, 10, 4, true>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        hipLaunchKernelGGL((k_attn_front<true, 10, 10, 4, true>), grid, dim3(kFrThreads), lds_bytes, st, a);
    } else {
        GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_front<false, 16, 16, 2, false>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        hipLaunchKernelGGL((k_attn_front<false, 16, 16, 2, false>), grid, dim3(kFrThreads), lds_bytes, st, a);
    }
    GNF_LAUNCH_CHECK("k_attn_front");
    return GNF_OK;
}

}  // namespace gnf
