// Backward of the edge-list attention front-end (forward: gnf_attn.hip; reference gnn.py:385-553), so that the
// drivers' DEFAULT GNN (run_grevnet.py:56,199-211) trains.  Per net, with the forward quantities
//   q = x Wq, k = x Wk, v = x Wv;  logit[e,h] = scale <q[s_e,h,:], k[r_e,h,:]>;  w = softmax over the edges into r_e
//   agg[r,h,:] = sum_e w[e,h] v[s_e,:];  new = agg Wo;  h0 = [x || new] | new
// and dh0 = dL/dh0 coming back from the MLP:
//   dnew = dh0[:, off:]           dagg = dnew Wo^T                      dWo = agg^T dnew
//   dw[e,h] = <dagg[r_e,h,:], v[s_e,:]>      dlogit[e,h] = w[e,h] (dw[e,h] - sum_{e' into r_e} w[e',h] dw[e',h])
//   dk[r,h,:] = sum_{e into r}  scale dlogit[e,h] q[s_e,h,:]            (receiver side:  k_attn_bwd_recv)
//   dq[u,h,:] = sum_{e out of u} scale dlogit[e,h] k[r_e,h,:]           (sender side:    k_attn_bwd_send,
//   dv[u,:]   = sum_{e out of u} sum_h w[e,h] dagg[r_e,h,:]              by-sender CSR, no atomics)
//   dx = dq Wq^T + dk Wk^T + dv Wv^T (+ dh0[:, :H] when the node is concatenated)   (k_attn_bwd_dx)
//   dWq = x^T dq, dWk = x^T dk, dWv = x^T dv                             (grouped dW GEMM, gnf_train.hip)
// The sender-side pass needs no edge ids: it recomputes w[e,h] from the per-(receiver, head) softmax max and
// normaliser the receiver-side pass leaves in `stats`, and reads dagg per receiver.
// One wave per row; lanes run over the row's edges (tiles of 64), components are wave-reduced.  Correctness
// first: these kernels are latency-bound like their forward twins.
#include "gnf_common.h"

namespace gnf {

struct AttnBwdArgs {
    const float* qkv[2];   // [N, P]  q | k | v  (P = 2 nh kq + v)
    const float* Wo[2];    // [nh v, C]
    const float* dh0[2];   // [N, in0]
    float* dqkv[2];        // [N, P]  dq | dk | dv
    float* agg[2];         // [N, nh v]
    float* dagg[2];        // [N, nh v]
    float* stats[2];       // [N, 3 nh]  softmax max | normaliser | sum_e w dw
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* rowptr_t;
    const int32_t* col_t;
    int32_t n, H, nh, kq, v, C, concat, in0;
    float scale;
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

static constexpr int kRowsPerBlock = 4;  // one wave per row

__global__ __launch_bounds__(256) void k_attn_bwd_recv(const AttnBwdArgs a) {
    extern __shared__ float sm[];  // Wo [NV][C] | per wave: dnew [C] | dagg [NV]
    const int net = blockIdx.y;
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd, C = a.C;
    float* wo = sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* dnew = wo + NV * C + wave * (C + NV);
    float* dagg = dnew + C;
    for (int i = tid; i < NV * C; i += 256) wo[i] = a.Wo[net][i];
    __syncthreads();
    const int r = blockIdx.x * kRowsPerBlock + wave;
    if (r >= a.n) return;
    const float* qkv = a.qkv[net];
    const int off = a.concat ? a.H : 0;
    for (int c = lane; c < C; c += 64) dnew[c] = a.dh0[net][(int64_t)r * a.in0 + off + c];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < NV; i += 64) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += dnew[c] * wo[i * C + c];
        dagg[i] = s;
        a.dagg[net][(int64_t)r * NV + i] = s;
    }
    __builtin_amdgcn_wave_barrier();
    const int beg = a.rowptr[r], end = a.rowptr[r + 1];
    const float* krow = qkv + (int64_t)r * P + nq;
    for (int h = 0; h < nh; ++h) {
        const float* kh = krow + h * kq;
        auto logit_of = [&](int s) -> float {
            const float* qh = qkv + (int64_t)s * P + h * kq;
            float d = 0.f;
            for (int j = 0; j < kq; ++j) d += qh[j] * kh[j];
            return d * a.scale;
        };
        auto dw_of = [&](int s) -> float {
            const float* vs = qkv + (int64_t)s * P + 2 * nq;
            float d = 0.f;
            for (int j = 0; j < vd; ++j) d += dagg[h * vd + j] * vs[j];
            return d;
        };
        // pass A: max;  pass B: normaliser and sum exp * dw
        float m = -INFINITY;
        for (int e0 = beg; e0 < end; e0 += 64) {
            const int e = e0 + lane;
            m = fmaxf(m, e < end ? logit_of(a.col[e]) : -INFINITY);
        }
        m = wave_max(m);
        float z = 0.f, s1 = 0.f;
        for (int e0 = beg; e0 < end; e0 += 64) {
            const int e = e0 + lane;
            if (e < end) {
                const int s = a.col[e];
                const float ex = expf(logit_of(s) - m);
                z += ex;
                s1 += ex * dw_of(s);
            }
        }
        z = wave_sum(z);
        s1 = wave_sum(s1);
        const float sumw = end > beg ? s1 / z : 0.f;
        if (lane == 0) {
            float* st = a.stats[net] + (int64_t)r * 3 * nh;
            st[h] = m;
            st[nh + h] = z;
            st[2 * nh + h] = sumw;
        }
        // pass C: dk[r, h, :] and agg[r, h, :]
        for (int j0 = 0; j0 < (kq > vd ? kq : vd); ++j0) {
            float dk = 0.f, ag = 0.f;
            for (int e0 = beg; e0 < end; e0 += 64) {
                const int e = e0 + lane;
                if (e < end) {
                    const int s = a.col[e];
                    const float w = expf(logit_of(s) - m) / z;
                    if (j0 < kq) dk += w * (dw_of(s) - sumw) * a.scale * qkv[(int64_t)s * P + h * kq + j0];
                    if (j0 < vd) ag += w * qkv[(int64_t)s * P + 2 * nq + j0];
                }
            }
            dk = wave_sum(dk);
            ag = wave_sum(ag);
            if (lane == 0) {
                if (j0 < kq) a.dqkv[net][(int64_t)r * P + nq + h * kq + j0] = dk;
                if (j0 < vd) a.agg[net][(int64_t)r * NV + h * vd + j0] = ag;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_attn_bwd_send(const AttnBwdArgs a) {
    const int net = blockIdx.y;
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x * kRowsPerBlock + wave;
    if (u >= a.n) return;
    const float* qkv = a.qkv[net];
    const float* dagg = a.dagg[net];
    const float* stats = a.stats[net];
    const int beg = a.rowptr_t[u], end = a.rowptr_t[u + 1];
    const float* qrow = qkv + (int64_t)u * P;
    const float* vrow = qrow + 2 * nq;
    // per-edge softmax weight and dlogit, recomputed from the receiver's statistics
    auto edge = [&](int r, int h, float* w_out) -> float {
        const float* qh = qrow + h * kq;
        const float* kr = qkv + (int64_t)r * P + nq + h * kq;
        float d = 0.f;
        for (int j = 0; j < kq; ++j) d += qh[j] * kr[j];
        const float* st = stats + (int64_t)r * 3 * nh;
        const float w = expf(d * a.scale - st[h]) / st[nh + h];
        float dw = 0.f;
        for (int j = 0; j < vd; ++j) dw += dagg[(int64_t)r * NV + h * vd + j] * vrow[j];
        *w_out = w;
        return w * (dw - st[2 * nh + h]);
    };
    for (int h = 0; h < nh; ++h)
        for (int j0 = 0; j0 < kq; ++j0) {
            float dq = 0.f;
            for (int e0 = beg; e0 < end; e0 += 64) {
                const int e = e0 + lane;
                if (e < end) {
                    const int r = a.col_t[e];
                    float w;
                    dq += edge(r, h, &w) * a.scale * qkv[(int64_t)r * P + nq + h * kq + j0];
                }
            }
            dq = wave_sum(dq);
            if (lane == 0) a.dqkv[net][(int64_t)u * P + h * kq + j0] = dq;
        }
    for (int j0 = 0; j0 < vd; ++j0) {  // v is shared by the heads: one sum over (edge, head)
        float dv = 0.f;
        for (int e0 = beg; e0 < end; e0 += 64) {
            const int e = e0 + lane;
            if (e < end) {
                const int r = a.col_t[e];
                for (int h = 0; h < nh; ++h) {
                    float w;
                    edge(r, h, &w);
                    dv += w * dagg[(int64_t)r * NV + h * vd + j0];
                }
            }
        }
        dv = wave_sum(dv);
        if (lane == 0) a.dqkv[net][(int64_t)u * P + 2 * nq + j0] = dv;
    }
}

// g[r, f] += sum over nets of ( dq Wq^T + dk Wk^T + dv Wv^T )[r, f]  (+ dh0[r, f] when concatenated)
//            (+ g_s[r, f] + g_t[r, f] for residual blocks: s, t = MLP(h0) + x, gnn.py:547-548)
struct AttnDxArgs {
    const float* dqkv[2];
    const float* Wq[2];
    const float* Wk[2];
    const float* Wv[2];
    const float* dh0[2];
    const float* gst[2];  // NULL unless residual
    float* g;
    int64_t ldg;
    int32_t n, H, nq, v, in0, concat;
};

__global__ __launch_bounds__(256) void k_attn_bwd_dx(const AttnDxArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)a.n * a.H) return;
    const int64_t r = i / a.H;
    const int f = (int)(i - r * a.H);
    const int P = 2 * a.nq + a.v;
    float acc = 0.f;
    for (int net = 0; net < 2; ++net) {
        const float* d = a.dqkv[net] + r * P;
        const float* wq = a.Wq[net] + (int64_t)f * a.nq;
        const float* wk = a.Wk[net] + (int64_t)f * a.nq;
        const float* wv = a.Wv[net] + (int64_t)f * a.v;
        float s = 0.f;
        for (int c = 0; c < a.nq; ++c) s += d[c] * wq[c] + d[a.nq + c] * wk[c];
        for (int c = 0; c < a.v; ++c) s += d[2 * a.nq + c] * wv[c];
        if (a.concat) s += a.dh0[net][r * a.in0 + f];
        if (a.gst[net]) s += a.gst[net][r * a.H + f];
        acc += s;
    }
    a.g[r * a.ldg + f] += acc;
}

// at[2]: the two attention blocks; qkv: [2][N, P] forward projections (launch_attn_front's scratch);
// dh0 / gst: per net;  dqkv / agg: per net outputs kept for the dW GEMMs;  dagg / stats: scratch.
int launch_attn_backward(const GnfAttn* const* at, int64_t n, int32_t H, int32_t in0, const int32_t* rowptr,
                         const int32_t* col, const int32_t* rowptr_t, const int32_t* col_t, const float* const* qkv,
                         const float* const* dh0, const float* const* gst, float* const* dqkv, float* const* agg,
                         float* const* dagg, float* const* stats, float* g_cond, int64_t ldg, hipStream_t st) {
    if (n == 0) return GNF_OK;
    const GnfAttn* a0 = at[0];
    AttnBwdArgs a;
    for (int q = 0; q < 2; ++q) {
        a.qkv[q] = qkv[q];
        a.Wo[q] = at[q]->Wo;
        a.dh0[q] = dh0[q];
        a.dqkv[q] = dqkv[q];
        a.agg[q] = agg[q];
        a.dagg[q] = dagg[q];
        a.stats[q] = stats[q];
    }
    a.rowptr = rowptr;
    a.col = col;
    a.rowptr_t = rowptr_t;
    a.col_t = col_t;
    a.n = (int32_t)n;
    a.H = H;
    a.nh = a0->num_heads;
    a.kq = a0->kq_dim;
    a.v = a0->v_dim;
    a.C = a0->out_dim;
    a.concat = a0->concat ? 1 : 0;
    a.in0 = in0;
    a.scale = a0->kq_dim_division ? 1.f / sqrtf((float)a0->kq_dim) : 1.f;
    const int NV = a.nh * a.v;
    const size_t lds = ((size_t)NV * a.C + (size_t)kRowsPerBlock * (a.C + NV)) * sizeof(float);
    if (lds > 160 * 1024) {
        set_error("attention backward needs %zu bytes of LDS for Wo: unsupported", lds);
        return GNF_EUNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_recv),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const dim3 grid((unsigned)((n + kRowsPerBlock - 1) / kRowsPerBlock), 2);
    hipLaunchKernelGGL(k_attn_bwd_recv, grid, dim3(256), lds, st, a);
    GNF_LAUNCH_CHECK("k_attn_bwd_recv");
    hipLaunchKernelGGL(k_attn_bwd_send, grid, dim3(256), 0, st, a);
    GNF_LAUNCH_CHECK("k_attn_bwd_send");
    AttnDxArgs d;
    for (int q = 0; q < 2; ++q) {
        d.dqkv[q] = dqkv[q];
        d.Wq[q] = at[q]->Wq;
        d.Wk[q] = at[q]->Wk;
        d.Wv[q] = at[q]->Wv;
        d.dh0[q] = dh0[q];
        d.gst[q] = a0->residual ? gst[q] : nullptr;
    }
    d.g = g_cond;
    d.ldg = ldg;
    d.n = (int32_t)n;
    d.H = H;
    d.nq = a.nh * a.kq;
    d.v = a.v;
    d.in0 = in0;
    d.concat = a.concat;
    hipLaunchKernelGGL(k_attn_bwd_dx, dim3((unsigned)((n * H + 255) / 256)), dim3(256), 0, st, d);
    GNF_LAUNCH_CHECK("k_attn_bwd_dx");
    return GNF_OK;
}

}  // namespace gnf
