// Edge-list multi-head self-attention front-end of a GNN: the part of
// DMSelfAttentionMLP._build (/root/reference/gnn.py:503-545) that precedes its MLP, i.e.
//   q = x Wq, k = x Wk  [N, heads, kq];  v = x Wv  [N, v] (repeated over heads, gnn.py:521-524)
//   logit[e, h] = < q[sender(e), h, :], k[receiver(e), h, :] >   ( / sqrt(kq) if kq_dim_division )
//        (the reference passes (values, q, k) into DMSelfAttention(values, KEYS, QUERIES), gnn.py:528:
//         the module's sender "keys" are the Wq projection, its receiver "queries" the Wk projection)
//   w[e, h]     = softmax over the incoming edges of receiver(e)    (gnn.py:458-464)
//   agg[r,h,:]  = sum_e w[e,h] * v[sender(e), :]                    (gnn.py:468-475)
//   new[r, :]   = reshape(agg[r], heads*v) @ Wo                     (gnn.py:533-540, no bias)
//   h0[r, :]    = [x[r] || new[r]]  (concat)  or  new[r]            (gnn.py:542-543)
// h0 then feeds the same MLP machinery as the message-passing GNNs (fused MFMA kernel or layered path).
// Two launches for BOTH nets of a coupling half-step (blockIdx.y = net): projection, then
// attention + output projection (+ concat).  The [E, heads, *] edge tensors of the TF graph are never
// materialised: each (receiver, head) thread walks its CSR row twice (max, then exp / weighted sum).
#include "gnf_attn_dev.h"

#include <stdlib.h>

namespace gnf {

struct AttnArgs {
    const float* Wq[2];
    const float* Wk[2];
    const float* Wv[2];
    const float* Wo[2];
    float* qkv[2];  // [N, 2*nh*kq + v] scratch per net
    float* h0[2];   // [N, in0] output per net
    float* agg_out[2];  // NULL, or [N, nh*v]: the attended values (training: the backward pass reads them back)
    float* mz_out[2];   // NULL, or [N, 3*nh]: softmax running max at [h], denominator at [nh + h] (third block: backward scratch)
    const int32_t* rowptr;
    const int32_t* col;
    const float* x;
    int64_t ldx;
    int32_t n_nodes, H, nh, kq, v, C, concat, in0;
    float scale;  // 1 or 1/sqrt(kq)
};

// RULE for these small kernels (measured: kernel time ~ number of SEQUENTIAL global-memory round trips,
// ~1-2 us each at this occupancy): every staging phase first issues ALL of its loads into registers
// and only then stores to LDS.  A "load; store; next iteration" loop costs one round trip per
// iteration.
#define GNF_STAGE_COPY(DST, SRC, COUNT, TID, NTHR)                                              \
    do {                                                                                         \
        for (int base_ = 0; base_ < (COUNT); base_ += (NTHR)*16) {                               \
            float reg_[16];                                                                      \
            _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                  \
                const int i_ = base_ + (TID) + q_ * (NTHR);                                      \
                reg_[q_] = (SRC)[i_ < (COUNT) ? i_ : 0];                                         \
            }                                                                                    \
            _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                  \
                const int i_ = base_ + (TID) + q_ * (NTHR);                                      \
                if (i_ < (COUNT)) (DST)[i_] = reg_[q_];                                          \
            }                                                                                    \
        }                                                                                        \
    } while (0)

// qkv[r, :] = x[r, :] @ [Wq | Wk | Wv].  Block = (64 column lanes) x (4 row quads) = 16 rows; Wq, Wk, Wv
// are staged in LDS as three plain copies (no index arithmetic), the 16 x rows likewise; a thread owns
// columns tx, tx+64, ... for its 4 rows (x values are LDS broadcasts).
static constexpr int kProjRows = 16;

template <int NH, int KQ, int VD>  // 0 = run-time value; the reference defaults (8, 10, 10) get a folded instance
__global__ __launch_bounds__(256) void k_attn_proj(const AttnArgs a) {
    extern __shared__ float sm[];  // Wq [H][nq] | Wk [H][nq] | Wv [H][v] | x rows [kProjRows][H]
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * kProjRows;
    const int H = a.H, nq = (NH ? NH : a.nh) * (KQ ? KQ : a.kq), vd = VD ? VD : a.v, P = 2 * nq + vd;
    float* wq = sm;
    float* wk = wq + H * nq;
    float* wv = wk + H * nq;
    float* xs = wv + H * vd;
    const int tid = threadIdx.x;
    const int tx = tid & 63, ty = tid >> 6;
    // one round trip: weights and the x rows (the rows are contiguous when ldx == H; else row by row)
    {
        const float* Wq = a.Wq[net];
        const float* Wk = a.Wk[net];
        const float* Wv = a.Wv[net];
        float xr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // up to 4*64 = 256 features per row quad lane set
            const int rl = ty * 4 + q;
            const int r = row0 + rl < a.n_nodes ? row0 + rl : a.n_nodes - 1;
            xr[q] = tx < H ? a.x[(int64_t)r * a.ldx + tx] : 0.f;
        }
        GNF_STAGE_COPY(wq, Wq, H * nq, tid, 256);
        GNF_STAGE_COPY(wk, Wk, H * nq, tid, 256);
        GNF_STAGE_COPY(wv, Wv, H * vd, tid, 256);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (tx < H) xs[(ty * 4 + q) * H + tx] = xr[q];
        for (int f = tx + 64; f < H; f += 64)  // H > 64: remaining features (rare)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rl = ty * 4 + q;
                const int r = row0 + rl < a.n_nodes ? row0 + rl : a.n_nodes - 1;
                xs[rl * H + f] = a.x[(int64_t)r * a.ldx + f];
            }
    }
    __syncthreads();
    float* out = a.qkv[net];
    const int rb = ty * 4;
    for (int c = tx; c < P; c += 64) {
        const float* w;
        int ldw;
        if (c < nq) {
            w = wq + c, ldw = nq;
        } else if (c < 2 * nq) {
            w = wk + (c - nq), ldw = nq;
        } else {
            w = wv + (c - 2 * nq), ldw = vd;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k = 0; k < H; ++k) {
            const float wv_ = w[k * ldw];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(xs[(rb + q) * H + k], wv_, acc[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = row0 + rb + q;
            if (r < a.n_nodes) out[(int64_t)r * P + c] = acc[q];
        }
    }
}

// Attention + output projection (+ concat) for RB consecutive receiver rows per workgroup.
//   round trip 1: rowptr slice, the receivers' k rows ("queries"), their x rows, Wo  -> LDS
//   round trip 2: the col segment of the tile
//   round trip 3: q ("keys") and v rows of every sender on the tile's edges, one sender row per wave
//                 iteration, coalesced along the feature axis, all issued before the first LDS store
//   then, from LDS only: a group of kEL consecutive lanes owns one (row, head) and takes the row's edges
//   kEL at a time (logit = <q[sender], k[receiver]>), max and sum combined with xor-shuffles; one lane
//   per (row, head, value component) forms attended = sum_e w[e,h] v[e,j]; output projection with Wo.
// All loops have run-time bounds.  Rows are processed in edge tiles of kEdgeCap (max pass over all
// tiles first when there is more than one), so any degree is supported.
static constexpr int kEL = 4;
static constexpr int kEdgeCap = 256;

template <int NH, int KQ, int VD, int OC>  // 0 = run-time value
__global__ __launch_bounds__(512) void k_attn_agg(const AttnArgs a, int RB) {
    extern __shared__ float lds[];
    const int net = blockIdx.y;
    const int nh = NH ? NH : a.nh, kq = KQ ? KQ : a.kq, vd = VD ? VD : a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const int OCr = OC ? OC : a.C;
    const int QV = nq + vd;  // floats staged per edge: q (all heads) then v
    float* agg_lds = lds;                          // [RB][NV]   attended values (accumulated over tiles)
    float* e_lds = agg_lds + RB * NV;              // [kEdgeCap][QV]
    float* k_lds = e_lds + kEdgeCap * QV;          // [RB][nq]   receivers' "queries"
    float* w_lds = k_lds + RB * nq;                // [kEdgeCap][nh] logits, then exp weights
    float* mx_lds = w_lds + kEdgeCap * nh;         // [RB][nh] running max
    float* den_lds = mx_lds + RB * nh;             // [RB][nh]
    float* wo_lds = den_lds + RB * nh;             // [NV][C]
    float* x_lds = wo_lds + NV * OCr;              // [RB][H]
    int* rp_lds = reinterpret_cast<int*>(x_lds + RB * a.H);  // [RB + 1]
    int* col_lds = rp_lds + RB + 1;                // [kEdgeCap]
    const int row0 = blockIdx.x * RB;
    const float* qkv = a.qkv[net];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    // ---- round trip 1 ----------------------------------------------------------------------------
    {
        int rp = 0;
        if (tid <= RB) {
            const int r = row0 + tid;
            rp = a.rowptr[r < a.n_nodes ? r : a.n_nodes];
        }
        // k rows and x rows of the RB receivers: (row, feature) flattened, rows clamped
        float kreg[4], xreg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * nthr;
            const int rl = i / nq < RB ? i / nq : RB - 1, c = i - (i / nq) * nq;
            const int r = row0 + rl < a.n_nodes ? row0 + rl : a.n_nodes - 1;
            kreg[q] = qkv[(int64_t)r * P + nq + c];
            const int rl2 = i / a.H < RB ? i / a.H : RB - 1, f = i - (i / a.H) * a.H;
            const int r2 = row0 + rl2 < a.n_nodes ? row0 + rl2 : a.n_nodes - 1;
            xreg[q] = a.x[(int64_t)r2 * a.ldx + f];
        }
        GNF_STAGE_COPY(wo_lds, a.Wo[net], NV * OCr, tid, nthr);
        if (tid <= RB) rp_lds[tid] = rp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * nthr;
            if (i < RB * nq) k_lds[i] = kreg[q];
            if (i < RB * a.H) x_lds[i] = xreg[q];
        }
        for (int i = tid + 4 * nthr; i < RB * nq; i += nthr) {  // very wide heads only
            const int rl = i / nq, c = i - rl * nq;
            const int r = row0 + rl < a.n_nodes ? row0 + rl : a.n_nodes - 1;
            k_lds[i] = qkv[(int64_t)r * P + nq + c];
        }
        for (int i = tid + 4 * nthr; i < RB * a.H; i += nthr) {
            const int rl = i / a.H, f = i - rl * a.H;
            const int r = row0 + rl < a.n_nodes ? row0 + rl : a.n_nodes - 1;
            x_lds[i] = a.x[(int64_t)r * a.ldx + f];
        }
        for (int i = tid; i < RB * NV; i += nthr) agg_lds[i] = 0.f;
        for (int i = tid; i < RB * nh; i += nthr) {
            mx_lds[i] = -INFINITY;
            den_lds[i] = 0.f;
        }
    }
    __syncthreads();
    const int seg_beg = rp_lds[0], seg_end = rp_lds[RB];
    const int el = tid & (kEL - 1);
    const int grp = tid / kEL;  // (row, head) pair index inside the workgroup
    const bool has_grp = grp < RB * nh;
    const int g_rl = has_grp ? grp / nh : 0, g_h = has_grp ? grp - g_rl * nh : 0;
    const int ntiles = (seg_end - seg_beg + kEdgeCap - 1) / kEdgeCap;

    for (int pass = (ntiles > 1 ? 0 : 1); pass < 2; ++pass) {  // pass 0: max over all tiles (multi-tile only)
        for (int t0 = seg_beg; t0 < seg_end; t0 += kEdgeCap) {
            const int t1 = t0 + kEdgeCap < seg_end ? t0 + kEdgeCap : seg_end;
            const int ne = t1 - t0;
            __syncthreads();
            // ---- round trip 2: col ------------------------------------------------------------------
            if (tid < ne) col_lds[tid] = a.col[t0 + tid];  // kEdgeCap <= blockDim is guaranteed by the launcher
            __syncthreads();
            // ---- round trip 3: sender rows (q, and v when needed); wave w takes edges w, w+nwave, ... -----
            {
                constexpr int kMaxIt = 16;  // kEdgeCap / min(nwave) = 128 / 8... covered by the outer loop
                const bool need_v = (pass == 1 || ntiles == 1);
                for (int eb = 0; eb < ne; eb += nwave * kMaxIt) {
                    float rq[kMaxIt][2], rv[kMaxIt];
#pragma unroll
                    for (int it = 0; it < kMaxIt; ++it) {
                        const int e = eb + wave + it * nwave;
                        const float* src = qkv + (int64_t)col_lds[e < ne ? e : 0] * P;
                        rq[it][0] = src[lane < nq ? lane : 0];
                        rq[it][1] = src[lane + 64 < nq ? lane + 64 : 0];
                        rv[it] = src[2 * nq + (lane < vd ? lane : 0)];
                    }
#pragma unroll
                    for (int it = 0; it < kMaxIt; ++it) {
                        const int e = eb + wave + it * nwave;
                        if (e < ne) {
                            if (lane < nq) e_lds[e * QV + lane] = rq[it][0];
                            if (lane + 64 < nq) e_lds[e * QV + lane + 64] = rq[it][1];
                            if (need_v && lane < vd) e_lds[e * QV + nq + lane] = rv[it];
                        }
                    }
                    // heads*kq > 128 or v > 64: the rest, plainly
                    for (int it = 0; it < kMaxIt; ++it) {
                        const int e = eb + wave + it * nwave;
                        if (e >= ne) break;
                        const float* src = qkv + (int64_t)col_lds[e] * P;
                        for (int c = lane + 128; c < nq; c += 64) e_lds[e * QV + c] = src[c];
                        if (need_v)
                            for (int c = lane + 64; c < vd; c += 64) e_lds[e * QV + nq + c] = src[2 * nq + c];
                    }
                }
            }
            __syncthreads();
            if (has_grp) {
                const int beg = rp_lds[g_rl] > t0 ? rp_lds[g_rl] : t0;
                const int end = rp_lds[g_rl + 1] < t1 ? rp_lds[g_rl + 1] : t1;
                const float* kr = k_lds + g_rl * nq + g_h * kq;
                float mx = -INFINITY;
                for (int e = beg + el; e < end; e += kEL) {
                    const float* qs = e_lds + (e - t0) * QV + g_h * kq;
                    float l = 0.f;
                    for (int d = 0; d < kq; ++d) l = fmaf(qs[d], kr[d], l);
                    l *= a.scale;
                    w_lds[(e - t0) * nh + g_h] = l;
                    mx = fmaxf(mx, l);
                }
#pragma unroll
                for (int o = 1; o < kEL; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                if (ntiles > 1 && pass == 0) {
                    if (el == 0) mx_lds[grp] = fmaxf(mx_lds[grp], mx);
                } else {
                    if (ntiles > 1) mx = mx_lds[grp];  // global max of the row from pass 0
                    else if (el == 0) mx_lds[grp] = mx;  // (kept for mz_out)
                    float den = 0.f;
                    for (int e = beg + el; e < end; e += kEL) {
                        const float w = expf(w_lds[(e - t0) * nh + g_h] - mx);
                        w_lds[(e - t0) * nh + g_h] = w;
                        den += w;
                    }
#pragma unroll
                    for (int o = 1; o < kEL; o <<= 1) den += __shfl_xor(den, o, 64);
                    if (el == 0) den_lds[grp] += den;
                }
            }
            if (pass == 1 || ntiles == 1) {
                __syncthreads();
                // attended[rl, h, j] += sum over the tile's edges of this row of w[e, h] * v[e, j]
                for (int i = tid; i < RB * NV; i += nthr) {
                    const int rl = i / NV, hj = i - rl * NV, h = hj / vd, j = hj - h * vd;
                    const int beg = rp_lds[rl] > t0 ? rp_lds[rl] : t0;
                    const int end = rp_lds[rl + 1] < t1 ? rp_lds[rl + 1] : t1;
                    float acc = 0.f;
                    for (int e = beg; e < end; ++e) acc = fmaf(w_lds[(e - t0) * nh + h], e_lds[(e - t0) * QV + nq + j], acc);
                    agg_lds[i] += acc;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < RB * NV; i += nthr) {
        const int rl = i / NV, h = (i - rl * NV) / vd;
        const float den = den_lds[rl * nh + h];
        agg_lds[i] = den > 0.f ? agg_lds[i] / den : 0.f;  // no incoming edge -> 0 (gnn.py:403)
    }
    __syncthreads();
    if (a.agg_out[net])
        for (int i = tid; i < RB * NV; i += nthr) {
            const int rl = i / NV;
            if (row0 + rl < a.n_nodes) a.agg_out[net][(int64_t)(row0 + rl) * NV + (i - rl * NV)] = agg_lds[i];
        }
    if (a.mz_out[net])
        for (int i = tid; i < RB * nh; i += nthr) {
            const int rl = i / nh, h = i - rl * nh;
            if (row0 + rl >= a.n_nodes) continue;
            float* mz = a.mz_out[net] + (int64_t)(row0 + rl) * 3 * nh;
            mz[h] = mx_lds[i];
            mz[nh + h] = den_lds[i] > 0.f ? den_lds[i] : 1.f;
        }
    float* h0 = a.h0[net];
    const int off = a.concat ? a.H : 0;
    for (int i = tid; i < RB * OCr; i += nthr) {
        const int rl = i / OCr, c = i - rl * OCr;
        const int r = row0 + rl;
        if (r >= a.n_nodes) continue;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < NV; ++k) acc = fmaf(agg_lds[rl * NV + k], wo_lds[k * OCr + c], acc);
        h0[(int64_t)r * a.in0 + off + c] = acc;
    }
    if (a.concat)
        for (int i = tid; i < RB * a.H; i += nthr) {
            const int rl = i / a.H, f = i - rl * a.H;
            const int r = row0 + rl;
            if (r < a.n_nodes) h0[(int64_t)r * a.in0 + f] = x_lds[i];
        }
}

int validate_attn(const GnfAttn* at, const GnfMlp* mlp, int32_t H, const char* what) {
    if (at->out_dim < 1 || !attn_geometry_ok(at->num_heads, at->kq_dim, at->v_dim, H)) {
        set_error("%s: attention dims heads=%d kq=%d v=%d out=%d on H=%d outside heads in 1..%d, heads*kq <= %d, heads*v <= %d, "
                  "pad16(2 heads kq + v) + pad16(H) + H <= %d, out >= 1",
                  what, at->num_heads, at->kq_dim, at->v_dim, at->out_dim, H, kAttnMaxHeads, kAttnMaxWidth, kAttnMaxWidth,
                  kAttnMaxRowFloats);
        return GNF_ESHAPE;
    }
    if (!at->Wq || !at->Wk || !at->Wv || !at->Wo) {
        set_error("%s: null attention weight pointer", what);
        return GNF_EINVAL;
    }
    const int in0 = (at->concat ? H : 0) + at->out_dim;
    if (mlp->dims[0] != in0) {
        set_error("%s: MLP input width %d but the attention front-end produces %d (H=%d, out_dim=%d, concat=%d)",
                  what, mlp->dims[0], in0, H, at->out_dim, at->concat);
        return GNF_ESHAPE;
    }
    if (at->residual && mlp->dims[mlp->num_layers] != H) {
        set_error("%s: residual needs MLP output width == H", what);
        return GNF_ESHAPE;
    }
    if (at->layer_norm && (!at->ln_gamma || !at->ln_beta)) {
        set_error("%s: layer_norm needs ln_gamma and ln_beta", what);
        return GNF_EINVAL;
    }
    return GNF_OK;
}

// ------------------------------------------------------------------------------------------------
// Attention + output projection + concat, thread = (receiver row, head).  A workgroup takes 64 consecutive receiver
// rows; wave w works on head w, lane = row.  The q | v columns of every node the tile's edges point at are staged
// once in the LDS row window (gnf_attn_dev.h); a thread then walks its own CSR row with the online-softmax
// recurrence, all arithmetic in registers: no per-edge staging, no cross-lane reductions.  On complete graphs (the
// drivers' default datasets) the lanes of a wave read the same sender row at the same time (LDS broadcast).
// The edge-tiled kernel above re-staged every sender row once per 16 receivers and took 290 us on 32 complete
// 100-node graphs; this one is the default whenever heads <= 8 and kq, v <= KQM, VDM.
// ------------------------------------------------------------------------------------------------
// PAR = 2: two lanes (lane, lane ^ 32) share a receiver row, each walks one half of its edges with its own running
// (max, Z, weighted values); the two states are merged by the online-softmax rule and lane parity 0 writes.
template <int KQM, int VDM, bool WIN, int PAR = 1>
__device__ __forceinline__ void attn_fwd_thread(const AttnArgs& a, const float* __restrict__ qkv, int r, int h,
                                                const float* win, int win_lo, int WS, const int* cols, int col_base,
                                                bool v2, float* agg_row, int par = 0, float* mz = nullptr) {
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd;
    float kreg[KQM], ag[VDM];
#pragma unroll
    for (int j = 0; j < KQM; ++j) kreg[j] = j < kq ? qkv[(int64_t)r * P + nq + h * kq + j] : 0.f;
#pragma unroll
    for (int j = 0; j < VDM; ++j) ag[j] = 0.f;
    float m = -INFINITY, z = 0.f;
    int beg = a.rowptr[r], end = a.rowptr[r + 1];
    const bool any_edge = end > beg;
    if (PAR == 2) {
        const int mid = beg + (end - beg + 1) / 2;
        if (par == 0) end = mid; else beg = mid;
    }
    // edges four at a time: the four column indices, then the four rows, are independent loads issued together;
    // only the softmax recurrence is sequential (a thread walking its row one edge at a time waits for
    // col -> row -> arithmetic on every edge)
    auto fetch = [&](int s_, float (&qq)[KQM], float (&vv)[VDM]) {
        if (WIN) {
            const float* row = win + (s_ - win_lo) * WS;
            load_row<KQM>(row + h * kq, kq, v2, qq);
            load_row<VDM>(row + nq, vd, v2, vv);
        } else {
            const float* row = qkv + (int64_t)s_ * P;
            load_row<KQM>(row + h * kq, kq, v2, qq);
            load_row<VDM>(row + 2 * nq, vd, v2, vv);
        }
    };
    auto update = [&](const float (&qq)[KQM], const float (&vv)[VDM]) {
        float lg = 0.f;
#pragma unroll
        for (int j = 0; j < KQM; ++j) lg += qq[j] * kreg[j];
        lg *= a.scale;
        const float mn = fmaxf(m, lg);
        const float sc = __expf(m - mn), pe = __expf(lg - mn);
        z = z * sc + pe;
#pragma unroll
        for (int j = 0; j < VDM; ++j) ag[j] = ag[j] * sc + pe * vv[j];
        m = mn;
    };
    int e = beg;
    for (; e + 4 <= end; e += 4) {
        int s4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] = cols[e + u - col_base];
        float qq[4][KQM], vv[4][VDM];
#pragma unroll
        for (int u = 0; u < 4; ++u) fetch(s4[u], qq[u], vv[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) update(qq[u], vv[u]);
    }
    for (; e < end; ++e) {
        float qq[KQM], vv[VDM];
        fetch(cols[e - col_base], qq, vv);
        update(qq, vv);
    }
    if (PAR == 2) {
        const float m2 = __shfl_xor(m, 32, 64), z2 = __shfl_xor(z, 32, 64);
        const float mn = fmaxf(m, m2);
        const float c1 = m == -INFINITY ? 0.f : __expf(m - mn), c2 = m2 == -INFINITY ? 0.f : __expf(m2 - mn);
        z = z * c1 + z2 * c2;
#pragma unroll
        for (int j = 0; j < VDM; ++j) ag[j] = ag[j] * c1 + __shfl_xor(ag[j], 32, 64) * c2;
        m = mn;
        if (par != 0) return;
    }
    const float inv = any_edge ? 1.f / z : 0.f;
#pragma unroll
    for (int j = 0; j < VDM; ++j)
        if (j < vd) agg_row[h * vd + j] = ag[j] * inv;
    if (mz) {
        mz[(int64_t)r * 3 * nh + h] = m;
        mz[(int64_t)r * 3 * nh + nh + h] = any_edge ? z : 1.f;
    }
}

// ROWS: receiver rows per workgroup (lanes ROWS .. 63 of every head's wave idle): 64, or 32 while 64-row tiles would be
// fewer workgroups than the chip has CUs (the drivers' default batch: 3200 nodes = 50 tiles per net)
template <int KQM, int VDM, int ROWS = 64>
__global__ __launch_bounds__(512) void k_attn_fwd_rows(const AttnArgs a, int win_cap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int net = blockIdx.y;
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd, C = a.C, H = a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * ROWS;
    int* s_rp = reinterpret_cast<int*>(sm);
    int* s_hdr = s_rp + ROWS + 1;
    float* wo = reinterpret_cast<float*>(s_hdr + 3);        // [NV][C]
    float* agg_s = wo + NV * C;                            // [64][NV + 1]
    int* s_col = reinterpret_cast<int*>(agg_s + ROWS * (NV + 1));  // [kRowsColCap]
    float* win = reinterpret_cast<float*>(s_col + kRowsColCap);          // [cap][WS]
    const int WS = (nq + vd + 2) & ~1;  // even: rows stay 8-byte aligned
    if (tid <= ROWS) {
        const int r = row0 + tid;
        s_rp[tid] = a.rowptr[r < a.n_nodes ? r : a.n_nodes];
    }
    {   // Wo: all loads issued before the stores
        const float* Wo = a.Wo[net];
        for (int base = 0; base < NV * C; base += 512 * 8) {
            float reg[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = base + tid + q * 512;
                reg[q] = Wo[i < NV * C ? i : 0];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = base + tid + q * 512;
                if (i < NV * C) wo[i] = reg[q];
            }
        }
    }
    __syncthreads();
    const float* qkv = a.qkv[net];
    bool cols_in_lds = false;
    const int lo = stage_window(a.col, s_rp, ROWS, s_hdr, win_cap, tid, 512, [&](int lo_, int cnt) {
        // q at [0, nq), v at [2 nq, 2 nq + vd) of a row; in 8-byte units when every segment starts on an even column
        if (((nq | vd) & 1) == 0 && (reinterpret_cast<uintptr_t>(qkv) & 7) == 0 && (reinterpret_cast<uintptr_t>(win) & 7) == 0)
            window_copy2(win, WS, cnt, (nq + vd) >> 1, tid, 512, [&](int rr, int c2) {
                const int c = 2 * c2;
                return *reinterpret_cast<const f32x2_win*>(qkv + (int64_t)(lo_ + rr) * P + (c < nq ? c : nq + c));
            });
        else
            window_copy(win, WS, cnt, nq + vd, tid, 512, [&](int rr, int c) {
                return qkv[(int64_t)(lo_ + rr) * P + (c < nq ? c : nq + c)];
            });
    }, s_col, kRowsColCap, &cols_in_lds);
    __syncthreads();
    const int* cols = cols_in_lds ? s_col : a.col;
    const int col_base = cols_in_lds ? s_rp[0] : 0;
    constexpr int PAR = ROWS == 32 ? 2 : 1;  // 32-row tiles: the wave's other 32 lanes take half of every row's edges
    const int row_l = PAR == 2 ? (lane & 31) : lane, par = PAR == 2 ? (lane >> 5) : 0;
    const int r = row0 + row_l;
    if (wave < nh && row_l < ROWS && r < a.n_nodes) {
        // 8-byte reads when every row segment is even-sized and 8-byte aligned (window base / qkv base and row stride)
        const bool even = ((kq | vd | nq) & 1) == 0;
        if (lo >= 0)
            attn_fwd_thread<KQM, VDM, true, PAR>(a, qkv, r, wave, win, lo, WS, cols, col_base,
                                                 even && (reinterpret_cast<uintptr_t>(win) & 7) == 0, agg_s + row_l * (NV + 1), par,
                                                 a.mz_out[net]);
        else
            attn_fwd_thread<KQM, VDM, false, PAR>(a, qkv, r, wave, win, 0, WS, cols, col_base,
                                                  even && (reinterpret_cast<uintptr_t>(qkv) & 7) == 0, agg_s + row_l * (NV + 1), par,
                                                  a.mz_out[net]);
    }
    __syncthreads();
    if (a.agg_out[net])  // the attended values stay for the backward pass (coalesced rows)
        for (int i = tid; i < ROWS * NV; i += 512) {
            const int rl = i / NV, c = i - rl * NV;
            if (row0 + rl < a.n_nodes) a.agg_out[net][(int64_t)(row0 + rl) * NV + c] = agg_s[rl * (NV + 1) + c];
        }
    // output projection new = agg Wo (Wo broadcast from LDS, agg row per lane) and h0 = [x || new] | new
    float* h0 = a.h0[net];
    const int off = a.concat ? H : 0;
    if (lane < ROWS && r < a.n_nodes) {  // (32-row tiles: lanes 0 .. 31, for which row_l = lane)
        // wave w takes the contiguous columns [w * cw, (w + 1) * cw): per agg element one LDS read of it and a run of
        // consecutive Wo values (broadcast), instead of two LDS reads per multiply-add
        const float* ar = agg_s + lane * (NV + 1);
        const int cw = (C + 7) / 8;
        for (int c0 = wave * cw; c0 < C && c0 < (wave + 1) * cw; c0 += 10) {
            float acc[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) acc[k] = 0.f;
            const int nc = C - c0 < (wave + 1) * cw - c0 ? C - c0 : (wave + 1) * cw - c0;
            for (int i = 0; i < NV; ++i) {
                const float av = ar[i];
                const float* wrow = wo + i * C + c0;
#pragma unroll
                for (int k = 0; k < 10; ++k) acc[k] += av * (k < nc ? wrow[k] : 0.f);
            }
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (k < nc) h0[(int64_t)r * a.in0 + off + c0 + k] = acc[k];
        }
        if (a.concat)
            for (int f = wave; f < H; f += 8) h0[(int64_t)r * a.in0 + f] = a.x[(int64_t)r * a.ldx + f];
    }
}

size_t attn_scratch_floats(const GnfAttn* at, int64_t n_nodes, int32_t in0) {
    if (!at) return 0;
    const size_t P = 2 * (size_t)at->num_heads * at->kq_dim + at->v_dim;
    // [2][n][P] q|k|v, [2][n][in0] h0, [2][n][heads*v] attended values, [2][n][3*heads] softmax statistics
    return 2 * (size_t)n_nodes * (P + (size_t)in0 + (size_t)at->num_heads * at->v_dim + 3 * (size_t)at->num_heads);
}

size_t attn_stash_slot_floats(const GnfFlow* flow, int64_t n_nodes) {
    const GnfMlp* net = flow && flow->s_nets ? &flow->s_nets[0] : nullptr;
    if (!net || !net->attn || n_nodes <= 0) return 0;
    return attn_scratch_floats(net->attn, n_nodes, net->dims[0]);
}

// nets: 1 or 2 attention blocks sharing x / topology; writes h0[q] ([N, in0]) for each.
int launch_attn_front(const int32_t* rowptr, const int32_t* col, int64_t n, const float* x, int64_t ldx,
                      int32_t H, const GnfAttn* const* at, int nets, int32_t in0, float* scratch,
                      float* const* h0_out, hipStream_t st, int64_t n_edges, bool need_qkv, const float* const* packed,
                      float* const* agg_out, float* const* mz_out) {
    if (n == 0) return GNF_OK;
    const GnfAttn* a0 = at[0];
    for (int q = 1; q < nets; ++q)
        if (at[q]->num_heads != a0->num_heads || at[q]->kq_dim != a0->kq_dim || at[q]->v_dim != a0->v_dim ||
            at[q]->out_dim != a0->out_dim || at[q]->concat != a0->concat ||
            at[q]->kq_dim_division != a0->kq_dim_division) {
            set_error("attention blocks of one coupling must have identical hyper-parameters");
            return GNF_ESHAPE;
        }
    AttnArgs a;
    const size_t P = 2 * (size_t)a0->num_heads * a0->kq_dim + a0->v_dim;
    // sparse batches (mean in-degree under ~24: the config-2 batch has 12): ONE launch, projections on the matrix
    // cores per tile (gnf_attn_front.hip); dense batches (complete graphs) keep the per-node projection below
    if (packed && packed[0] && n_edges > 0 && n_edges < 24 * n && attn_front_fused_ok(a0, H)) {
        float* qkv_ptr[2] = {scratch, scratch + (size_t)(nets > 1 ? 1 : 0) * n * P};
        return launch_attn_front_fused(rowptr, col, n, x, ldx, H, at, nets, in0, packed, need_qkv ? qkv_ptr : nullptr,
                                       h0_out, st, agg_out, mz_out);
    }
    for (int q = 0; q < 2; ++q) {
        const GnfAttn* t = at[q < nets ? q : 0];
        a.Wq[q] = t->Wq;
        a.Wk[q] = t->Wk;
        a.Wv[q] = t->Wv;
        a.Wo[q] = t->Wo;
        a.qkv[q] = scratch + (size_t)q * n * P;
        a.h0[q] = h0_out[q < nets ? q : 0];
        a.agg_out[q] = agg_out ? agg_out[q < nets ? q : 0] : nullptr;
        a.mz_out[q] = mz_out ? mz_out[q < nets ? q : 0] : nullptr;
    }
    a.rowptr = rowptr;
    a.col = col;
    a.x = x;
    a.ldx = ldx;
    a.n_nodes = (int32_t)n;
    a.H = H;
    a.nh = a0->num_heads;
    a.kq = a0->kq_dim;
    a.v = a0->v_dim;
    a.C = a0->out_dim;
    a.concat = a0->concat ? 1 : 0;
    a.in0 = in0;
    a.scale = a0->kq_dim_division ? 1.f / sqrtf((float)a0->kq_dim) : 1.f;
    const bool ref_default = a.nh == 8 && a.kq == 10 && a.v == 10;  // run_grevnet.py:74-76
    const bool old_attn = opt(OPT_ATTN_KERNEL) == 2;  // shape forcing for the parity tests (gnf_set_option): 1 rows kernel, 2 edge-tiled kernel,
    const bool rows_always = opt(OPT_ATTN_KERNEL) == 1;   // 3 the matrix-core attention core
    const bool core_always = opt(OPT_ATTN_KERNEL) == 3;
    // sparse batches (mean in-degree under ~24: the config-2 batch has 12) are 8 % faster through the edge-tiled kernel;
    // the rows kernel wins by 4.5 x on the complete graphs of the drivers' default dataset (degree 100)
    const bool sparse = !rows_always && n_edges > 0 && n_edges < 24 * n;
    // which kernel behind the projection: the thread-per-(row, head) rows kernel (dense batches, heads <= 8, kq, v <= 32, its
    // window in LDS), the matrix-core attention core (wider heads, or an edge tile that does not fit the LDS), else the
    // edge-tiled kernel
    int RB = 512 / (a.nh * kEL);  // (row, head) groups of kEL lanes of the edge-tiled kernel
    if (RB < 1) RB = 1;
    const size_t rows_fixed = (size_t)(kRowsTile + 1 + 3 + kRowsColCap) * sizeof(int) +
                              ((size_t)a.nh * a.v * a.C + (size_t)kRowsTile * (a.nh * a.v + 1)) * sizeof(float);
    const bool use_rows = !old_attn && !core_always && !sparse && a.nh <= kRowsMaxHeads && a.kq <= 32 && a.v <= 32 &&
                          rows_fixed + 64 * (size_t)(a.nh * a.kq + a.v + 2) * sizeof(float) <= (size_t)kRowsLdsBudget;
    const size_t tile_lds = ((size_t)RB * a.nh * a.v + (size_t)kEdgeCap * (a.nh * a.kq + a.v) + (size_t)RB * a.nh * a.kq +
                             (size_t)kEdgeCap * a.nh + 2 * (size_t)RB * a.nh + (size_t)a.nh * a.v * a.C + (size_t)RB * H) * sizeof(float) +
                            (size_t)(RB + 1 + kEdgeCap) * sizeof(int);
    const bool use_core = !use_rows && (core_always || (!old_attn && (a.kq > 32 || a.v > 32)) || tile_lds > 160 * 1024);
    const size_t proj_lds = ((size_t)H * P + (size_t)kProjRows * H) * sizeof(float);
    // The projection.  The drivers' default head geometry at small H keeps its folded vector-unit instance (the weights
    // fit the LDS many times over: run_grevnet.py:74-76); every other geometry - first of all the data driver's one head of
    // 64 / 64 at H = 100, where that kernel took 38.6 us per half-step - multiplies on the matrix cores.
    if (ref_default && proj_lds <= 64 * 1024) {
        GNF_ONCE_PER_DEVICE(
            GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_proj<8, 10, 10>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        const dim3 pgrid((unsigned)((n + kProjRows - 1) / kProjRows), nets);
        hipLaunchKernelGGL((k_attn_proj<8, 10, 10>), pgrid, dim3(256), proj_lds, st, a);
        GNF_LAUNCH_CHECK("k_attn_proj");
    } else {
        // (the core kernel's concat half of h0 rides along here, from the registers that hold the rows)
        const int rc = launch_attn_proj_mfma(at, nets, n, x, ldx, H, a.qkv, st, use_core && a.concat ? a.h0 : nullptr, in0);
        if (rc) return rc;
    }
    if (use_rows) {
        const int NV = a.nh * a.v, nq = a.nh * a.kq;
        const size_t fixed = rows_fixed;
        {
            const int cap = (int)((kRowsLdsBudget - fixed) / ((size_t)(nq + a.v + 2) * sizeof(float)));
            GNF_ONCE_PER_DEVICE(
                const void* ks[4] = {reinterpret_cast<const void*>(k_attn_fwd_rows<10, 10, 64>),
                                     reinterpret_cast<const void*>(k_attn_fwd_rows<32, 32, 64>),
                                     reinterpret_cast<const void*>(k_attn_fwd_rows<10, 10, 32>),
                                     reinterpret_cast<const void*>(k_attn_fwd_rows<32, 32, 32>)};
                for (const void* k : ks) GNF_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
            // (the LDS plan above is the 64-row one: a 32-row tile needs less)
            bool half_tiles = (int64_t)nets * ((n + 63) / 64) < 256;
            if (const int64_t force = opt(OPT_ATTN_BWD_ROWS); force == 64 || force == 32) half_tiles = force == 32;  // developer option
            const int rows = half_tiles ? 32 : 64;
            const dim3 rgrid((unsigned)((n + rows - 1) / rows), nets);
            const bool small = a.kq <= 10 && a.v <= 10;
            if (half_tiles) {
                if (small)
                    hipLaunchKernelGGL((k_attn_fwd_rows<10, 10, 32>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, cap);
                else
                    hipLaunchKernelGGL((k_attn_fwd_rows<32, 32, 32>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, cap);
            } else {
                if (small)
                    hipLaunchKernelGGL((k_attn_fwd_rows<10, 10, 64>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, cap);
                else
                    hipLaunchKernelGGL((k_attn_fwd_rows<32, 32, 64>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, cap);
            }
            GNF_LAUNCH_CHECK("k_attn_fwd_rows");
            return GNF_OK;
        }
    }
    {
        // Heads wider than the thread-per-(row, head) kernels' registers (kq or v above 32: the DATA driver's default is one
        // head of 64 / 64, train_grevnet_with_data.py:40-46) - or a geometry whose edge tile does not fit the LDS - take the
        // matrix-core attention core (gnf_attn_core.hip), then new = agg Wo on the generic GEMM tile into h0's columns.
        if (use_core) {
            if (a.concat && ref_default && proj_lds <= 64 * 1024) {   // (the folded projection instance does not write h0's concat half)
                for (int q = 0; q < nets; ++q) {
                    const int rc0 = launch_copy_rows(x, ldx, a.h0[q], in0, n, H, st);
                    if (rc0) return rc0;
                }
            }
            const int NV = a.nh * a.v;
            float* agg_def = scratch + 2 * (size_t)n * (P + (size_t)in0);   // attn_scratch_floats' layout
            float* aggs[2] = {a.agg_out[0] ? a.agg_out[0] : agg_def, a.agg_out[1] ? a.agg_out[1] : agg_def + (size_t)(nets > 1 ? 1 : 0) * n * NV};
            const float* qk[2] = {a.qkv[0], a.qkv[1]};
            float* mzs[2] = {a.mz_out[0], a.mz_out[1]};
            int rc = launch_attn_core(a0, nets, rowptr, col, n, x, ldx, H, in0, qk, aggs, mzs, a.h0, st);
            if (rc) return rc;
            const float* as_[2] = {aggs[0], aggs[1]};
            const float* nob[2] = {nullptr, nullptr};
            float* ys[2] = {a.h0[0] + (a.concat ? H : 0), a.h0[1] + (a.concat ? H : 0)};
            return launch_linear_splitk(as_, (int64_t)NV, a.Wo, nob, ys, (int64_t)in0, nets, n, NV, a.C, GNF_ACT_RELU, 0.f, 0, nullptr, 0, st);
        }
    }
    int threads = RB * a.nh * kEL;  // nh <= 64 -> at most 512 when RB == 1
    threads = (threads + 63) / 64 * 64;
    if (threads < kEdgeCap) threads = kEdgeCap;  // the col tile is staged one edge per thread
    const size_t agg_lds = ((size_t)RB * a.nh * a.v + (size_t)kEdgeCap * (a.nh * a.kq + a.v) + (size_t)RB * a.nh * a.kq +
                            (size_t)kEdgeCap * a.nh + 2 * (size_t)RB * a.nh + (size_t)a.nh * a.v * a.C + (size_t)RB * H) *
                               sizeof(float) +
                           (size_t)(RB + 1 + kEdgeCap) * sizeof(int);
    if (agg_lds > 160 * 1024) {
        set_error("attention tile needs %zu bytes of LDS: unsupported head configuration", agg_lds);
        return GNF_EUNSUPPORTED;
    }
    GNF_ONCE_PER_DEVICE(
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_agg<0, 0, 0, 0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_agg<8, 10, 10, 80>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    const dim3 agrid((unsigned)((n + RB - 1) / RB), nets);
    if (ref_default && a.C == 80)  // + attn_concat_heads_output_dim = 80 (run_grevnet.py:77)
        hipLaunchKernelGGL((k_attn_agg<8, 10, 10, 80>), agrid, dim3(threads), agg_lds, st, a, RB);
    else
        hipLaunchKernelGGL((k_attn_agg<0, 0, 0, 0>), agrid, dim3(threads), agg_lds, st, a, RB);
    GNF_LAUNCH_CHECK("k_attn_agg");
    return GNF_OK;
}

}  // namespace gnf
