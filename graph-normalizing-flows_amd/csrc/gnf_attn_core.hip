// Attention core on the fp32 matrix cores for head geometries the per-(row, head) thread kernels do not hold - first of
// all the DATA driver's default (train_grevnet_with_data.py:40-46, 303-310: ONE head, kq = v = 64, complete graphs), where
// the edge softmax of DMSelfAttention (/root/reference/gnn.py:385-477) is a pair of real GEMMs per graph:
//   S^T[s, r]  = < q[s, h, :], k[r, h, :] > * scale       [senders x kq] x [kq x receivers]
//   P^T[s, r]  = mult[r, s] * exp(S^T[s, r] - max_r)       mult = how often edge s -> r occurs (0: no edge)
//   O^T[:, r]  = V^T[:, senders] P^T[senders, r]           [v x senders] x [senders x receivers];  agg = O / sum_s P
// a block-sparse "flash attention" over each 64-row tile's SENDER WINDOW (the node range its incoming edges point into:
// the graphs its rows belong to), taken in chunks of CH nodes with the running-max recurrence between chunks, so any
// graph size and any edge list (sparse, directed, with repeated edges or isolated nodes) is handled by the one kernel:
//   * a workgroup = 4 waves = 4 row tiles of 16 receivers; heads one after the other;
//   * per chunk the senders' q rows ([CH][kq], k contiguous) and v rows TRANSPOSED ([v][CH], senders contiguous) are
//     staged in LDS with every load of a phase issued before the first store, and the tile's edges are scattered into a
//     [64][CH] table of 16-bit multiplicities (LDS atomics);
//   * S^T = Q K^T with the sender rows as the MFMA's first operand: a lane then holds the logits of ONE receiver
//     (lane & 15) for four consecutive senders per 16-sender tile - the softmax statistics of a receiver are register
//     reductions plus two cross-lane steps per chunk, and the lane's four P values are exactly the second operand of
//     the O^T = V^T P^T product: no trip through LDS between the two GEMMs;
//   * empty receivers give agg = 0 (gnn.py:403); the running max / denominator go to `mz` for the backward pass.
// Around it (launch_attn_core): the q | k | v projection (k_attn_proj, or the generic GEMM tile where Wq | Wk | Wv do not
// fit the LDS) in front - it also copies the node's own features into h0 when the block concatenates them (gnn.py:542-543),
// from the registers that hold them anyway - and the output projection new = agg Wo (generic GEMM tile) behind.
#include "gnf_attn_core_dev.h"

namespace gnf {

struct AttnCoreArgs {
    const float* qkv[2];  // [N, P] q | k | v per net
    float* agg[2];        // [N, nh * v] attended values (normalised)
    float* mz[2];         // NULL, or [N, 3 nh]: running max at [h], denominator at [nh + h]
    float* h0[2];         // concat: the node's own features go to h0[r, 0:H)
    const float* x;
    int64_t ldx;
    const int32_t* rowptr;
    const int32_t* col;
    int32_t n, H, nh, kq, v, in0, concat;
    float scale;
};

static constexpr int kCoreRows = 64;

// KG: k-groups (16 q / k components each) a head's logits reduce over, VT: 16-column tiles of v; ST: 16-sender tiles per chunk
template <int KG, int VT, int ST>
__global__ __launch_bounds__(256) void k_attn_core(const AttnCoreArgs a) {
    constexpr int CH = 16 * ST, QS = 16 * KG + 4, VS = CH + 4, MW = CH / 2 + 1;  // row strides: floats, floats, 32-bit words
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* q_lds = sm;                                                     // [CH][QS]
    float* vt_lds = q_lds + CH * QS;                                       // [16 VT][VS]
    unsigned* mult = reinterpret_cast<unsigned*>(vt_lds + 16 * VT * VS);   // [64][MW] pairs of 16-bit counts
    int* s_rp = reinterpret_cast<int*>(mult + kCoreRows * MW);             // [65]
    int* s_hdr = s_rp + kCoreRows + 1;                                     // lo, hi, overflow
    int* s_col = s_hdr + 3;                                                // [kCap] the tile's slice of col
    constexpr int kCap = core_col_cap<KG>();
    const int net = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const int row0 = blockIdx.x * kCoreRows;
    const float* __restrict__ qkv = a.qkv[net];
    if (tid <= kCoreRows) {
        const int r = row0 + tid;
        s_rp[tid] = a.rowptr[r < a.n ? r : a.n];
    }
    if (tid == 0) s_hdr[0] = 0x7fffffff, s_hdr[1] = -1, s_hdr[2] = 0;
    // the receiver's k row as the logits' second operand (k-slot g of the q-th MFMA of k-group kg = component 16 kg + 4 g + q):
    // head 0's is requested here, in the shadow of the rowptr / col round trips, head h + 1's behind head h's last chunk
    const int r = row0 + 16 * (tid >> 6) + (tid & 15);  // this lane's receiver
    f32x4 kB[KG];
    auto load_k = [&](int h) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * kg + 4 * ((tid & 63) >> 4) + q;
                kB[kg][q] = (r < a.n && j < a.kq) ? qkv[(int64_t)r * (2 * a.nh * a.kq + a.v) + a.nh * a.kq + h * a.kq + j] : 0.f;
            }
    };
    load_k(0);
    __syncthreads();
    core_window_scan(a.col, s_rp[0], s_rp[kCoreRows], s_hdr, tid, lane, s_col, kCap);
    __syncthreads();
    const bool col_kept = s_rp[kCoreRows] - s_rp[0] <= kCap;
    const int32_t* cols = col_kept ? s_col : a.col;
    const int col_base = col_kept ? s_rp[0] : 0;
    const int win_lo = s_hdr[0], win_n = s_hdr[1] >= s_hdr[0] ? s_hdr[1] - s_hdr[0] + 1 : 0;
    const int n_chunks = (win_n + CH - 1) / CH;
    // 16-byte staging loads: every q / v segment of a row starts on a multiple of four floats of a 16-byte aligned array
    const bool vec4 = ((kq | vd | P) & 3) == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0;
    const int my_row = 16 * wave + lrow;

    for (int h = 0; h < nh; ++h) {
        float m_run = -INFINITY, z_run = 0.f;
        f32x4 O[VT];
#pragma unroll
        for (int t = 0; t < VT; ++t) O[t] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch = 0; ch < n_chunks; ++ch) {
            const int c0 = ch * CH;
            const int cn = win_n - c0 < CH ? win_n - c0 : CH;  // senders of this chunk
            const bool restage_shared = h == 0 || n_chunks > 1;  // v rows and the multiplicities do not depend on the head
            __syncthreads();  // (the previous chunk / head has been read)
            // q rows of head h ([CH][16 KG], zero beyond kq and beyond the chunk) and - once per chunk - the v rows transposed
            // (vt[j][s]) and a cleared multiplicity table.  Every load of the phase is in flight before the first LDS store
            // (a "load, store" loop is one memory round trip per iteration); 16-byte loads where the rows allow them.
            if (vec4) {
                constexpr int WQ = 4 * KG, WV = 4 * VT;                       // float4 per row
                constexpr int PQ = CH * WQ / 256, PV = CH * WV / 256;         // per thread
                f32x4 rq[PQ], rv[PV];
#pragma unroll
                for (int u = 0; u < PQ; ++u) {
                    const int i = tid + 256 * u, s = i / WQ, j = 4 * (i % WQ);
                    rq[u] = (s < cn && j < kq) ? *reinterpret_cast<const f32x4*>(qkv + (int64_t)(win_lo + c0 + s) * P + h * kq + j)
                                               : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (restage_shared) {
#pragma unroll
                    for (int u = 0; u < PV; ++u) {
                        const int i = tid + 256 * u, s = i / WV, j = 4 * (i % WV);
                        rv[u] = (s < cn && j < vd) ? *reinterpret_cast<const f32x4*>(qkv + (int64_t)(win_lo + c0 + s) * P + 2 * nq + j)
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    for (int i = tid; i < kCoreRows * MW; i += 256) mult[i] = 0u;
                }
#pragma unroll
                for (int u = 0; u < PQ; ++u) {
                    const int i = tid + 256 * u;
                    *reinterpret_cast<f32x4*>(q_lds + (i / WQ) * QS + 4 * (i % WQ)) = rq[u];
                }
                if (restage_shared) {
#pragma unroll
                    for (int u = 0; u < PV; ++u) {
                        const int i = tid + 256 * u, s = i / WV, j = 4 * (i % WV);
#pragma unroll
                        for (int c = 0; c < 4; ++c) vt_lds[(j + c) * VS + s] = rv[u][c];
                    }
                }
            } else {
                constexpr int W = 16 * KG, PER = CH * W / 256;
                static_assert(PER % 8 == 0, "staging rounds of eight");
                for (int b = 0; b < PER; b += 8) {
                    float reg[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = tid + 256 * (b + u);
                        const int s = i / W, j = i % W;
                        reg[u] = (s < cn && j < kq) ? qkv[(int64_t)(win_lo + c0 + s) * P + h * kq + j] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = tid + 256 * (b + u);
                        q_lds[(i / W) * QS + (i % W)] = reg[u];
                    }
                }
                if (restage_shared) {
                    constexpr int WV_ = 16 * VT, PERV = CH * WV_ / 256;
                    static_assert(PERV % 8 == 0, "staging rounds of eight");
                    for (int b = 0; b < PERV; b += 8) {
                        float reg[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = tid + 256 * (b + u);
                            const int s = i / WV_, j = i % WV_;
                            reg[u] = (s < cn && j < vd) ? qkv[(int64_t)(win_lo + c0 + s) * P + 2 * nq + j] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = tid + 256 * (b + u);
                            vt_lds[(i % WV_) * VS + (i / WV_)] = reg[u];
                        }
                    }
                    for (int i = tid; i < kCoreRows * MW; i += 256) mult[i] = 0u;
                }
            }
            __syncthreads();
            if (restage_shared) core_scatter_mult<CH>(mult, s_rp, cols, col_base, win_lo, c0, s_hdr, tid);
            __syncthreads();
            // ---- S^T tiles of the chunk: sender rows x this wave's 16 receivers ------------------------------------------
            f32x4 S[ST];
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (16 * t < cn) {
                    const float* qrow = q_lds + (16 * t + lrow) * QS + 4 * lgrp;
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) {
                        if (16 * kg < kq) {
                            const f32x4 qa = *reinterpret_cast<const f32x4*>(qrow + 16 * kg);
#pragma unroll
                            for (int q = 0; q < 4; ++q) S[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[q], kB[kg][q], S[t], 0, 0, 0);
                        }
                    }
                }
            }
            // lane holds S^T[sender 16 t + 4 lgrp + i][receiver lrow]; the matching multiplicities: two 32-bit words
            float cm = -INFINITY;
            float mlt[ST][4];
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                const unsigned* mw = mult + my_row * MW + 8 * t + 2 * lgrp;
                const unsigned w0 = 16 * t < cn ? mw[0] : 0u, w1 = 16 * t < cn ? mw[1] : 0u;
                mlt[t][0] = (float)(w0 & 0xffffu), mlt[t][1] = (float)(w0 >> 16);
                mlt[t][2] = (float)(w1 & 0xffffu), mlt[t][3] = (float)(w1 >> 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    S[t][i] *= a.scale;
                    if (mlt[t][i] > 0.f) cm = fmaxf(cm, S[t][i]);
                }
            }
            cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
            cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
            const float m_new = fmaxf(m_run, cm);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;  // (no edge so far: every weight below is 0)
            const float sc = __expf(m_run - m_safe);               // exp(-inf) = 0 while nothing has been accumulated
            float zs = 0.f;
#pragma unroll
            for (int t = 0; t < ST; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = mlt[t][i] > 0.f ? mlt[t][i] * __expf(S[t][i] - m_safe) : 0.f;
                    S[t][i] = p;
                    zs += p;
                }
            zs += __shfl_xor(zs, 16, 64);
            zs += __shfl_xor(zs, 32, 64);
            z_run = z_run * sc + zs;
            m_run = m_new;
            // ---- O^T += V^T P^T: the lane's P values are the second operand as they stand ----------------------------------
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                if (16 * vt < vd) {
                    O[vt] *= sc;
                    const float* vrow = vt_lds + (16 * vt + lrow) * VS + 4 * lgrp;
#pragma unroll
                    for (int t = 0; t < ST; ++t) {
                        if (16 * t < cn) {
                            const f32x4 va = *reinterpret_cast<const f32x4*>(vrow + 16 * t);
#pragma unroll
                            for (int q = 0; q < 4; ++q) O[vt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[q], S[t][q], O[vt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // ---- this head's attended values: lane holds O^T[v column 16 vt + 4 lgrp + i][receiver lrow] -----------------------
        if (r < a.n) {
            const bool poisoned = s_hdr[2] != 0;
            const float inv = poisoned ? NAN : (z_run > 0.f ? 1.f / z_run : 0.f);   // no incoming edge -> 0 (gnn.py:403)
            float* __restrict__ out = a.agg[net] + (int64_t)r * NV + h * vd;
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 16 * vt + 4 * lgrp + i;
                    if (j < vd) out[j] = O[vt][i] * inv;
                }
            if (a.mz[net] && lgrp == 0) {
                float* mz = a.mz[net] + (int64_t)r * 3 * nh;
                mz[h] = m_run;
                mz[nh + h] = z_run > 0.f ? z_run : 1.f;
            }
        }
        if (h + 1 < nh) load_k(h + 1);
    }
}

// ---- q | k | v = x [Wq | Wk | Wv] on the matrix cores -------------------------------------------------------------------
// k_attn_proj (gnf_attn.hip) stages all three weight matrices in LDS for every 16 rows and multiplies on the vector
// units: 38.6 us per half-step on the data driver's batch (2 718 nodes, H = 100, P = 192).  Here a workgroup owns 32 rows x
// all P columns: the rows go through LDS in chunks of 64 features, the weights come straight from memory (L2) as MFMA
// operands - lane (lrow, g) loads W[16 kg + 4 g + q][16 ct + lrow], 64-byte runs along a weight row - with every load of
// a chunk issued before the first MFMA; accumulators transposed (a lane holds four consecutive columns of a row).
struct AttnProjArgs {
    const float* Wq[2];
    const float* Wk[2];
    const float* Wv[2];
    float* qkv[2];
    float* h0[2];          // NULL, or the MLP's layer-0 rows [N, in0] of a concat block: h0[r, 0:H) = x[r, :] rides along (gnn.py:542-543)
    int32_t in0;
    const float* x;
    int64_t ldx;
    int32_t n, H, nq, v;
};
static constexpr int kPjRows = 32, kPjKC = 64, kPjKG = kPjKC / 16, kPjKS = kPjKC + 4;

__global__ __launch_bounds__(256) void k_attn_proj_mfma(const AttnProjArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[kPjRows * kPjKS];
    const int net = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int row0 = blockIdx.x * kPjRows;
    const int nq = a.nq, vd = a.v, P = 2 * nq + vd, H = a.H;
    const int n_ct = (P + 15) / 16;
    float* __restrict__ out = a.qkv[net];
    const bool xvec = (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
    const bool ovec = (P & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    for (int cp = 0; cp < n_ct; cp += 16) {  // column passes of 16 column tiles: wave w owns tiles cp + w + 4 b
        const float* wp[4];
        int ldw[4];
        bool live[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {  // this lane's weight column of tile b: which matrix, which column
            const int c = 16 * (cp + wave + 4 * b) + lrow;
            live[b] = c < P;
            if (c < nq)
                wp[b] = a.Wq[net] + c, ldw[b] = nq;
            else if (c < 2 * nq)
                wp[b] = a.Wk[net] + (c - nq), ldw[b] = nq;
            else
                wp[b] = a.Wv[net] + (live[b] ? c - 2 * nq : 0), ldw[b] = vd;
        }
        int nv = 0;  // live column tiles of this wave (wave-uniform)
#pragma unroll
        for (int b = 0; b < 4; ++b) nv += (cp + wave + 4 * b < n_ct) ? 1 : 0;
        f32x4 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[m][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < H; k0 += kPjKC) {
            const int kn = H - k0 < kPjKC ? H - k0 : kPjKC;
            // weights of the chunk: 4 k-groups x 4 column tiles x 4 = up to 64 loads per lane, all in flight together
            float bw[kPjKG][4][4];
#pragma unroll
            for (int kg = 0; kg < kPjKG; ++kg)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int k = 16 * kg + 4 * lgrp + q;
                        bw[kg][b][q] = (b < nv && live[b] && k < kn) ? wp[b][(int64_t)(k0 + k) * ldw[b]] : 0.f;
                    }
            // the rows' features of the chunk: 32 x 64 floats = 2 float4 per thread
            f32x4 xr[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + 256 * u, rl = i >> 4, c = 4 * (i & 15);
                const int r = row0 + rl;
                f32x4 v4 = {0.f, 0.f, 0.f, 0.f};
                if (r < a.n && c < kn) {
                    const float* px = a.x + (int64_t)r * a.ldx + k0 + c;
                    if (xvec && c + 3 < kn) {
                        v4 = *reinterpret_cast<const f32x4*>(px);
                    } else {
                        v4[0] = px[0];
                        if (c + 1 < kn) v4[1] = px[1];
                        if (c + 2 < kn) v4[2] = px[2];
                        if (c + 3 < kn) v4[3] = px[3];
                    }
                }
                xr[u] = v4;
            }
            __syncthreads();  // (the previous chunk / pass has been read)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + 256 * u;
                *reinterpret_cast<f32x4*>(xs + (i >> 4) * kPjKS + 4 * (i & 15)) = xr[u];
                if (cp == 0 && a.h0[net]) {   // the block's own features into its MLP's input rows, from the registers that hold them
                    const int rl = i >> 4, c = 4 * (i & 15), r = row0 + rl;
                    if (r < a.n) {
                        float* ph = a.h0[net] + (int64_t)r * a.in0 + k0 + c;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (c + q < kn) ph[q] = xr[u][q];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int kg = 0; kg < kPjKG; ++kg) {
                if (16 * kg < kn) {
                    f32x4 xa[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) xa[m] = *reinterpret_cast<const f32x4*>(xs + (16 * m + lrow) * kPjKS + 16 * kg + 4 * lgrp);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (b < nv)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[kg][b][q], xa[m][q], acc[m][b], 0, 0, 0);
                }
            }
        }
        // lane holds out[row 16 m + lrow][column 16 ct + 4 lgrp + i]
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b >= nv) continue;
            const int c = 16 * (cp + wave + 4 * b) + 4 * lgrp;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int r = row0 + 16 * m + lrow;
                if (r >= a.n || c >= P) continue;
                float* po = out + (int64_t)r * P + c;
                if (ovec && c + 3 < P) {
                    *reinterpret_cast<f32x4*>(po) = acc[m][b];
                } else {
                    po[0] = acc[m][b][0];
                    if (c + 1 < P) po[1] = acc[m][b][1];
                    if (c + 2 < P) po[2] = acc[m][b][2];
                    if (c + 3 < P) po[3] = acc[m][b][3];
                }
            }
        }
    }
}

int launch_attn_proj_mfma(const GnfAttn* const* at, int nets, int64_t n, const float* x, int64_t ldx, int32_t H, float* const* qkv,
                          hipStream_t st, float* const* h0_concat, int32_t in0) {
    if (n == 0) return GNF_OK;
    AttnProjArgs a;
    for (int q = 0; q < 2; ++q) {
        const GnfAttn* t = at[q < nets ? q : 0];
        a.Wq[q] = t->Wq, a.Wk[q] = t->Wk, a.Wv[q] = t->Wv, a.qkv[q] = qkv[q < nets ? q : 0];
        a.h0[q] = h0_concat ? h0_concat[q < nets ? q : 0] : nullptr;
    }
    a.in0 = in0;
    a.x = x, a.ldx = ldx, a.n = (int32_t)n, a.H = H, a.nq = at[0]->num_heads * at[0]->kq_dim, a.v = at[0]->v_dim;
    hipLaunchKernelGGL(k_attn_proj_mfma, dim3((unsigned)((n + kPjRows - 1) / kPjRows), (unsigned)nets), dim3(256), 0, st, a);
    GNF_LAUNCH_CHECK("k_attn_proj_mfma");
    return GNF_OK;
}

// is this head geometry the core kernel's? (else the edge-tiled kernel / the rows kernels of gnf_attn.hip)
bool attn_core_ok(const GnfAttn* at) { return at && at->kq_dim <= 256 && at->v_dim <= 256; }

template <int KG, int VT, int ST>
static size_t core_lds_bytes() {
    constexpr int CH = 16 * ST;
    return ((size_t)CH * (16 * KG + 4) + (size_t)16 * VT * (CH + 4)) * sizeof(float) + (size_t)kCoreRows * (CH / 2 + 1) * sizeof(unsigned) +
           (size_t)(kCoreRows + 1 + 3 + core_col_cap<KG>()) * sizeof(int);
}

// qkv[q]: the projections of net q ([N, P]); writes agg[q] ([N, nh v], normalised), mz[q] (nullable) and, for concat
// blocks, h0[q][:, 0:H) = x.  The output projection is the caller's next launch.
int launch_attn_core(const GnfAttn* a0, int nets, const int32_t* rowptr, const int32_t* col, int64_t n, const float* x, int64_t ldx,
                     int32_t H, int32_t in0, const float* const* qkv, float* const* agg, float* const* mz, float* const* h0,
                     hipStream_t st) {
    if (n == 0) return GNF_OK;
    AttnCoreArgs a;
    for (int q = 0; q < 2; ++q) {
        const int s = q < nets ? q : 0;
        a.qkv[q] = qkv[s], a.agg[q] = agg[s], a.mz[q] = mz ? mz[s] : nullptr, a.h0[q] = h0[s];
    }
    a.x = x, a.ldx = ldx, a.rowptr = rowptr, a.col = col;
    a.n = (int32_t)n, a.H = H, a.nh = a0->num_heads, a.kq = a0->kq_dim, a.v = a0->v_dim, a.in0 = in0, a.concat = a0->concat ? 1 : 0;
    a.scale = a0->kq_dim_division ? 1.f / sqrtf((float)a0->kq_dim) : 1.f;
    const dim3 grid((unsigned)((n + kCoreRows - 1) / kCoreRows), (unsigned)nets);
    if (a.kq <= 64 && a.v <= 64) {  // (the data driver's default: one k-group chain of 4, four v tiles, 128-sender chunks)
        const size_t lds = core_lds_bytes<4, 4, 8>();
        GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_core<4, 4, 8>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        hipLaunchKernelGGL((k_attn_core<4, 4, 8>), grid, dim3(256), lds, st, a);
    } else {
        const size_t lds = core_lds_bytes<16, 16, 4>();
        GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_core<16, 16, 4>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        hipLaunchKernelGGL((k_attn_core<16, 16, 4>), grid, dim3(256), lds, st, a);
    }
    GNF_LAUNCH_CHECK("k_attn_core");
    return GNF_OK;
}

}  // namespace gnf
