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
#include "gnf_common.h"

namespace gnf {

static constexpr int kAttnMaxKq = 32;
static constexpr int kAttnMaxV = 32;
static constexpr int kProjRows = 16;

struct AttnArgs {
    const float* Wq[2];
    const float* Wk[2];
    const float* Wv[2];
    const float* Wo[2];
    float* qkv[2];  // [N, 2*nh*kq + v] scratch per net
    float* h0[2];   // [N, in0] output per net
    const int32_t* rowptr;
    const int32_t* col;
    const float* x;
    int64_t ldx;
    int32_t n_nodes, H, nh, kq, v, C, concat, in0;
    float scale;  // 1 or 1/sqrt(kq)
};

// qkv[r, :] = x[r, :] @ [Wq | Wk | Wv]
__global__ __launch_bounds__(256) void k_attn_proj(const AttnArgs a) {
    extern __shared__ float xs[];  // [kProjRows][H]
    const int net = blockIdx.y;
    const int row0 = blockIdx.x * kProjRows;
    const int H = a.H, nq = a.nh * a.kq, P = 2 * nq + a.v;
    for (int i = threadIdx.x; i < kProjRows * H; i += 256) {
        const int rl = i / H, f = i - rl * H;
        const int r = row0 + rl;
        xs[i] = r < a.n_nodes ? a.x[(int64_t)r * a.ldx + f] : 0.f;
    }
    __syncthreads();
    const float* Wq = a.Wq[net];
    const float* Wk = a.Wk[net];
    const float* Wv = a.Wv[net];
    float* out = a.qkv[net];
    for (int i = threadIdx.x; i < kProjRows * P; i += 256) {
        const int rl = i / P, c = i - rl * P;
        const int r = row0 + rl;
        if (r >= a.n_nodes) continue;
        const float* W;
        int ldw, cc;
        if (c < nq) {
            W = Wq, ldw = nq, cc = c;
        } else if (c < 2 * nq) {
            W = Wk, ldw = nq, cc = c - nq;
        } else {
            W = Wv, ldw = a.v, cc = c - 2 * nq;
        }
        float acc = 0.f;
        for (int k = 0; k < H; ++k) acc = fmaf(xs[rl * H + k], W[(int64_t)k * ldw + cc], acc);
        out[(int64_t)r * P + c] = acc;
    }
}

// one thread per (receiver row, head); RB rows per workgroup; then the output projection
__global__ __launch_bounds__(256) void k_attn_agg(const AttnArgs a, int RB) {
    extern __shared__ float agg_lds[];  // [RB][nh*v]
    const int net = blockIdx.y;
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const int row0 = blockIdx.x * RB;
    const float* qkv = a.qkv[net];
    const int tid = threadIdx.x;
    if (tid < RB * nh) {
        const int rl = tid / nh, h = tid - rl * nh;
        const int r = row0 + rl;
        float acc[kAttnMaxV];
#pragma unroll
        for (int j = 0; j < kAttnMaxV; ++j) acc[j] = 0.f;
        float den = 0.f;
        if (r < a.n_nodes) {
            float kr[kAttnMaxKq];  // the receiver's "query" (Wk projection), this head
#pragma unroll
            for (int d = 0; d < kAttnMaxKq; ++d) kr[d] = d < kq ? qkv[(int64_t)r * P + nq + h * kq + d] : 0.f;
            const int beg = a.rowptr[r], end = a.rowptr[r + 1];
            float mx = -INFINITY;
            for (int e = beg; e < end; ++e) {
                const float* qs = qkv + (int64_t)a.col[e] * P + h * kq;
                float l = 0.f;
#pragma unroll
                for (int d = 0; d < kAttnMaxKq; ++d)
                    if (d < kq) l = fmaf(qs[d], kr[d], l);
                mx = fmaxf(mx, l * a.scale);
            }
            for (int e = beg; e < end; ++e) {
                const int sidx = a.col[e];
                const float* qs = qkv + (int64_t)sidx * P + h * kq;
                float l = 0.f;
#pragma unroll
                for (int d = 0; d < kAttnMaxKq; ++d)
                    if (d < kq) l = fmaf(qs[d], kr[d], l);
                const float w = expf(l * a.scale - mx);
                den += w;
                const float* vs = qkv + (int64_t)sidx * P + 2 * nq;
#pragma unroll
                for (int j = 0; j < kAttnMaxV; ++j)
                    if (j < vd) acc[j] = fmaf(w, vs[j], acc[j]);
            }
        }
        const float inv = den > 0.f ? 1.f / den : 0.f;  // no incoming edge -> 0 (gnn.py:403)
#pragma unroll
        for (int j = 0; j < kAttnMaxV; ++j)
            if (j < vd) agg_lds[rl * NV + h * vd + j] = acc[j] * inv;
    }
    __syncthreads();
    const float* Wo = a.Wo[net];
    float* h0 = a.h0[net];
    const int off = a.concat ? a.H : 0;
    for (int i = tid; i < RB * a.C; i += blockDim.x) {
        const int rl = i / a.C, c = i - rl * a.C;
        const int r = row0 + rl;
        if (r >= a.n_nodes) continue;
        float acc = 0.f;
        for (int k = 0; k < NV; ++k) acc = fmaf(agg_lds[rl * NV + k], Wo[(int64_t)k * a.C + c], acc);
        h0[(int64_t)r * a.in0 + off + c] = acc;
    }
    if (a.concat)
        for (int i = tid; i < RB * a.H; i += blockDim.x) {
            const int rl = i / a.H, f = i - rl * a.H;
            const int r = row0 + rl;
            if (r < a.n_nodes) h0[(int64_t)r * a.in0 + f] = a.x[(int64_t)r * a.ldx + f];
        }
}

int validate_attn(const GnfAttn* at, const GnfMlp* mlp, int32_t H, const char* what) {
    if (at->num_heads < 1 || at->num_heads > 64 || at->kq_dim < 1 || at->kq_dim > kAttnMaxKq ||
        at->v_dim < 1 || at->v_dim > kAttnMaxV || at->out_dim < 1) {
        set_error("%s: attention dims heads=%d kq=%d v=%d out=%d outside (1..64, 1..%d, 1..%d, >=1)", what,
                  at->num_heads, at->kq_dim, at->v_dim, at->out_dim, kAttnMaxKq, kAttnMaxV);
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
    return GNF_OK;
}

size_t attn_scratch_floats(const GnfAttn* at, int64_t n_nodes, int32_t in0) {
    if (!at) return 0;
    const size_t P = 2 * (size_t)at->num_heads * at->kq_dim + at->v_dim;
    return 2 * (size_t)n_nodes * (P + (size_t)in0);
}

// nets: 1 or 2 attention blocks sharing x / topology; writes h0[q] ([N, in0]) for each.
int launch_attn_front(const int32_t* rowptr, const int32_t* col, int64_t n, const float* x, int64_t ldx,
                      int32_t H, const GnfAttn* const* at, int nets, int32_t in0, float* scratch,
                      float* const* h0_out, hipStream_t st) {
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
    for (int q = 0; q < 2; ++q) {
        const GnfAttn* t = at[q < nets ? q : 0];
        a.Wq[q] = t->Wq;
        a.Wk[q] = t->Wk;
        a.Wv[q] = t->Wv;
        a.Wo[q] = t->Wo;
        a.qkv[q] = scratch + (size_t)q * n * P;
        a.h0[q] = h0_out[q < nets ? q : 0];
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
    hipLaunchKernelGGL(k_attn_proj, dim3((unsigned)((n + kProjRows - 1) / kProjRows), nets), dim3(256),
                       kProjRows * H * sizeof(float), st, a);
    GNF_LAUNCH_CHECK("k_attn_proj");
    int RB = 256 / a.nh;
    if (RB < 1) RB = 1;
    if (RB > 32) RB = 32;
    while ((size_t)RB * a.nh * a.v * sizeof(float) > 60 * 1024 && RB > 1) RB >>= 1;
    int threads = RB * a.nh;
    threads = (threads + 63) / 64 * 64;
    if (threads < 64) threads = 64;
    hipLaunchKernelGGL(k_attn_agg, dim3((unsigned)((n + RB - 1) / RB), nets), dim3(threads),
                       (size_t)RB * a.nh * a.v * sizeof(float), st, a, RB);
    GNF_LAUNCH_CHECK("k_attn_agg");
    return GNF_OK;
}

}  // namespace gnf
