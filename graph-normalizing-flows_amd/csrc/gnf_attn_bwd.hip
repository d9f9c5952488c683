// Backward of the edge-list attention front-end (forward: gnf_attn.hip; reference gnn.py:385-553), so that the
// drivers' DEFAULT GNN (run_grevnet.py:56,199-211) trains.  Per net, with the forward quantities
//   q = x Wq, k = x Wk, v = x Wv;  logit[e,h] = scale <q[s_e,h,:], k[r_e,h,:]>;  w = softmax over the edges into r_e
//   agg[r,h,:] = sum_e w[e,h] v[s_e,:];  new = agg Wo;  h0 = [x || new] | new
// and dh0 = dL/dh0 coming back from the MLP:
//   dnew = dh0[:, off:]           dagg = dnew Wo^T                      dWo = agg^T dnew
//   dw[e,h] = <dagg[r_e,h,:], v[s_e,:]>      dlogit[e,h] = w[e,h] (dw[e,h] - sum_{e' into r_e} w[e',h] dw[e',h])
//   dk[r,h,:] = sum_{e into r}  scale dlogit[e,h] q[s_e,h,:]            (receiver side)
//   dq[u,h,:] = sum_{e out of u} scale dlogit[e,h] k[r_e,h,:]           (sender side,
//   dv[u,:]   = sum_{e out of u} sum_h w[e,h] dagg[r_e,h,:]              by-sender CSR, no atomics)
//   dx = dq Wq^T + dk Wk^T + dv Wv^T (+ dh0[:, :H] when the node is concatenated)   (k_attn_bwd_dx)
//   dWq = x^T dq, dWk = x^T dk, dWv = x^T dv                             (grouped dW GEMM, gnf_train.hip)
// The sender-side pass needs no edge ids: it recomputes w[e,h] from the per-(receiver, head) softmax max and
// normaliser the forward pass leaves in `stats`, and reads dagg per receiver.
// Two kernel families, by head geometry: heads <= 8 with kq, v <= 32 (the drivers' default 8 x 10 / 10) take the
// thread-per-(row, head) kernels below; everything else inside the ABI's limit takes the matrix-core attention core
// (gnf_attn_core_bwd.hip).  (Rounds 1 - 5 also carried a lane-per-feature pair, k_attn_bwd_recv / _send<FU>, for wider heads:
// unreachable since the core kernels took every such geometry in round 5, removed in round 6.)
#include "gnf_attn_dev.h"

#include <stdlib.h>
#include <type_traits>

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

// ------------------------------------------------------------------------------------------------
// thread = (row, head) versions of the two passes (wave w = head w, lane = row of a 64-row tile; gnf_attn_dev.h):
// the rows the tile's edges point at sit in the LDS window, the tile's col slice too, and a thread walks its own CSR
// row with everything in registers - no per-edge LDS scratch, no cross-lane reductions.  These run whenever
// heads <= 8 and kq, v <= 32.
// ------------------------------------------------------------------------------------------------
// The forward pass leaves, per (row, head), the softmax statistics (running max m, denominator Z) and the attended values
// O = sum_e w_e v_e (AttnArgs.agg_out / mz_out): the backward pass never rebuilds them.  With
//   delta[r, h] = sum_e w_e <dO[r,h,:], v_e> = <dO[r,h,:], O[r,h,:]>
// the receiver side is ONE sweep over a row's edges (it used to be two: statistics, then gradients) and the sender side
// depends on nothing the receiver side computes, so on sparse batches both run in one launch (k_attn_bwd_edges: a sender
// tile forms the delta of its window rows while staging them); as two launches the receiver pass leaves delta in the
// statistics' third block for the sender pass.
// PAR = 2: two lanes (lane, lane ^ 32) share a receiver row, each walks one half of its edges; the sums are added at the
// end and lane parity 0 writes.
template <int KQM, int VDM, bool WIN, int PAR = 1, bool FX = false>
__device__ __forceinline__ void attn_recv_thread(const AttnBwdArgs& a, int net, int r, int h, const float* win,
                                                 int win_lo, int WS, const int* cols, int col_base, bool v2, int par = 0) {
    const int nh = FX ? 8 : a.nh, kq = FX ? KQM : a.kq, vd = FX ? VDM : a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const float* qkv = a.qkv[net];
    float kreg[KQM], dreg[VDM];
#pragma unroll
    for (int j = 0; j < KQM; ++j) kreg[j] = j < kq ? qkv[(int64_t)r * P + nq + h * kq + j] : 0.f;
    float delta = 0.f;
#pragma unroll
    for (int j = 0; j < VDM; ++j) {
        dreg[j] = j < vd ? a.dagg[net][(int64_t)r * NV + h * vd + j] : 0.f;
        delta += dreg[j] * (j < vd ? a.agg[net][(int64_t)r * NV + h * vd + j] : 0.f);
    }
    float* st = a.stats[net] + (int64_t)r * 3 * nh;
    const float m = st[h], rz = 1.f / st[nh + h];
    if (par == 0) st[2 * nh + h] = delta;  // (what a sender pass in a launch of its own reads)
    int beg = a.rowptr[r], end = a.rowptr[r + 1];
    if (PAR == 2) {  // this lane's half of the row's edges
        const int mid = beg + (end - beg + 1) / 2;
        if (par == 0) end = mid; else beg = mid;
    }
    auto fetch = [&](int s_, float (&qv)[KQM], float (&vv)[VDM]) {
        const float *qrow, *vrow;
        if (WIN) {
            qrow = win + (s_ - win_lo) * WS + h * kq;
            vrow = win + (s_ - win_lo) * WS + nq;
        } else {
            qrow = qkv + (int64_t)s_ * P + h * kq;
            vrow = qkv + (int64_t)s_ * P + 2 * nq;
        }
        load_row<KQM>(qrow, kq, v2, qv);
        load_row<VDM>(vrow, vd, v2, vv);
    };
    float dk[KQM];
#pragma unroll
    for (int j = 0; j < KQM; ++j) dk[j] = 0.f;
    auto accum = [&](const float (&qv)[KQM], const float (&vv)[VDM]) {
        float lg = 0.f, dw = 0.f;
#pragma unroll
        for (int j = 0; j < KQM; ++j) lg += qv[j] * kreg[j];
#pragma unroll
        for (int j = 0; j < VDM; ++j) dw += vv[j] * dreg[j];
        const float dl = __expf(lg * a.scale - m) * rz * (dw - delta);
#pragma unroll
        for (int j = 0; j < KQM; ++j) dk[j] += dl * qv[j];
    };
    {   // edges four at a time: column indices and rows of a chunk are independent loads issued together
        int e = beg;
        for (; e + 4 <= end; e += 4) {
            int s4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] = cols[e + u - col_base];
            float qv[4][KQM], vv[4][VDM];
#pragma unroll
            for (int u = 0; u < 4; ++u) fetch(s4[u], qv[u], vv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(qv[u], vv[u]);
        }
        for (; e < end; ++e) {
            float qv[KQM], vv[VDM];
            fetch(cols[e - col_base], qv, vv);
            accum(qv, vv);
        }
    }
    if (PAR == 2) {
#pragma unroll
        for (int j = 0; j < KQM; ++j) dk[j] += __shfl_xor(dk[j], 32, 64);
        if (par != 0) return;
    }
#pragma unroll
    for (int j = 0; j < KQM; ++j)
        if (j < kq) a.dqkv[net][(int64_t)r * P + nq + h * kq + j] = dk[j] * a.scale;
}

// ROWS: receiver rows per workgroup (lanes ROWS .. 63 of every head's wave idle).  64 suits large batches of complete
// graphs (the window IS the graph); 32-row tiles put twice as many CUs to work and split every row's edges over two lanes.
template <int KQM, int VDM, int ROWS, bool FX = false>
__device__ __forceinline__ void attn_recv_tile(const AttnBwdArgs& a, int net, int tile, int win_cap, float* sm) {
    const int nh = FX ? 8 : a.nh, kq = FX ? KQM : a.kq, vd = FX ? VDM : a.v, nq = nh * kq, P = 2 * nq + vd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = tile * ROWS;
    int* s_rp = reinterpret_cast<int*>(sm);
    int* s_hdr = s_rp + ROWS + 1;
    int* s_col = s_hdr + 3;
    float* win = reinterpret_cast<float*>(s_col + kRowsColCap);
    const int WS = (nq + vd + 2) & ~1;  // even: rows stay 8-byte aligned
    if (tid <= ROWS) {
        const int r = row0 + tid;
        s_rp[tid] = a.rowptr[r < a.n ? r : a.n];
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
    if (wave < nh && row_l < ROWS && r < a.n) {
        const bool even = ((kq | vd | nq) & 1) == 0;
        if (lo >= 0)
            attn_recv_thread<KQM, VDM, true, PAR, FX>(a, net, r, wave, win, lo, WS, cols, col_base,
                                                  even && (reinterpret_cast<uintptr_t>(win) & 7) == 0, par);
        else
            attn_recv_thread<KQM, VDM, false, PAR, FX>(a, net, r, wave, win, 0, WS, cols, col_base,
                                                   even && (reinterpret_cast<uintptr_t>(qkv) & 7) == 0, par);
    }
}

template <int KQM, int VDM, int ROWS = 64>
__global__ __launch_bounds__(512) void k_attn_bwd_recv_rows(const AttnBwdArgs a, int win_cap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_recv_tile<KQM, VDM, ROWS>(a, blockIdx.y, blockIdx.x, win_cap, sm);
}

// sender side: window rows = the receivers' k | dagg (row stride WS), and with SW their m | Z | delta too (else those
// come from global memory)
// DG (one launch, window too wide for the LDS): delta formed per edge from global memory - nothing may be read that the
// receiver tiles of the same launch write
template <int KQM, int VDM, bool WIN, bool SW, bool DG, int PAR = 1, bool FX = false>   // PAR = 2: lanes (lane, lane ^ 32) share a sender row
__device__ __forceinline__ void attn_send_thread(const AttnBwdArgs& a, int net, int u_, int h, const float* win,
                                                 int win_lo, int WS, const int* cols, int col_base, bool v2, float* dvp_out,
                                                 int par = 0) {
    const int nh = FX ? 8 : a.nh, kq = FX ? KQM : a.kq, vd = FX ? VDM : a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const float* qkv = a.qkv[net];
    const float* dagg = a.dagg[net];
    const float* stats = a.stats[net];
    float qreg[KQM], vreg[VDM], dq[KQM], dvp[VDM];
#pragma unroll
    for (int j = 0; j < KQM; ++j) {
        qreg[j] = j < kq ? qkv[(int64_t)u_ * P + h * kq + j] : 0.f;
        dq[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < VDM; ++j) {
        vreg[j] = j < vd ? qkv[(int64_t)u_ * P + 2 * nq + j] : 0.f;
        dvp[j] = 0.f;
    }
    int beg = a.rowptr_t[u_], end = a.rowptr_t[u_ + 1];
    if (PAR == 2) {
        const int mid = beg + (end - beg + 1) / 2;
        if (par == 0) end = mid; else beg = mid;
    }
    auto fetch = [&](int r, float (&kv)[KQM], float (&dv_)[VDM], float (&st3)[3]) {
        const float *krow, *drow;
        if (WIN) {
            krow = win + (r - win_lo) * WS + h * kq;
            drow = win + (r - win_lo) * WS + nq + h * vd;
        } else {
            krow = qkv + (int64_t)r * P + nq + h * kq;
            drow = dagg + (int64_t)r * NV + h * vd;
        }
        if (WIN && SW) {
            const float* st = win + (r - win_lo) * WS + nq + NV;
            st3[0] = st[h];
            st3[1] = st[nh + h];
            st3[2] = st[2 * nh + h];
        } else {
            const float* st = stats + (int64_t)r * 3 * nh;
            st3[0] = st[h];
            st3[1] = st[nh + h];
            st3[2] = DG ? 0.f : st[2 * nh + h];
        }
        load_row<KQM>(krow, kq, v2, kv);
        load_row<VDM>(drow, vd, v2, dv_);
        if (DG) {
            float ag[VDM];
            load_row<VDM>(a.agg[net] + (int64_t)r * NV + h * vd, vd, false, ag);
#pragma unroll
            for (int j = 0; j < VDM; ++j) st3[2] += dv_[j] * ag[j];
        }
    };
    auto accum = [&](const float (&kv)[KQM], const float (&dv_)[VDM], const float (&st3)[3]) {
        float lg = 0.f, dw = 0.f;
#pragma unroll
        for (int j = 0; j < KQM; ++j) lg += kv[j] * qreg[j];
#pragma unroll
        for (int j = 0; j < VDM; ++j) dw += dv_[j] * vreg[j];
        const float w = __expf(lg * a.scale - st3[0]) / st3[1];
        const float dl = w * (dw - st3[2]);
#pragma unroll
        for (int j = 0; j < KQM; ++j) dq[j] += dl * kv[j];
#pragma unroll
        for (int j = 0; j < VDM; ++j) dvp[j] += w * dv_[j];
    };
    {
        int e = beg;
        for (; e + 4 <= end; e += 4) {  // four edges' loads in flight together
            int r4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r4[u] = cols[e + u - col_base];
            float kv[4][KQM], dv_[4][VDM], st3[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) fetch(r4[u], kv[u], dv_[u], st3[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(kv[u], dv_[u], st3[u]);
        }
        for (; e < end; ++e) {
            float kv[KQM], dv_[VDM], st3[3];
            fetch(cols[e - col_base], kv, dv_, st3);
            accum(kv, dv_, st3);
        }
    }
    if (PAR == 2) {
#pragma unroll
        for (int j = 0; j < KQM; ++j) dq[j] += __shfl_xor(dq[j], 32, 64);
#pragma unroll
        for (int j = 0; j < VDM; ++j) dvp[j] += __shfl_xor(dvp[j], 32, 64);
    }
#pragma unroll
    for (int j = 0; j < VDM; ++j) dvp_out[j] = dvp[j];
    if (par != 0) return;
#pragma unroll
    for (int j = 0; j < KQM; ++j)
        if (j < kq) a.dqkv[net][(int64_t)u_ * P + h * kq + j] = dq[j] * a.scale;
}

// SW (the one-launch form): the window rows also carry the receivers' m | Z | delta, delta formed here from the staged dagg
// rows and the forward pass's attended values - nothing the receiver tiles write is read.  (On the complete 100-node
// graphs of the drivers' datasets the window of a tile that straddles two graphs has no room for them: two launches.)
template <int KQM, int VDM, int ROWS, bool SW, bool FX = false>
__device__ __forceinline__ void attn_send_tile(const AttnBwdArgs& a, int net, int tile, int win_cap, float* sm) {
    const int nh = FX ? 8 : a.nh, kq = FX ? KQM : a.kq, vd = FX ? VDM : a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = tile * ROWS;
    int* s_rp = reinterpret_cast<int*>(sm);
    int* s_hdr = s_rp + ROWS + 1;
    int* s_col = s_hdr + 3;
    float* win = reinterpret_cast<float*>(s_col + kRowsColCap);
    const int WS = (nq + NV + (SW ? 3 * nh : 0) + 2) & ~1;  // even: rows stay 8-byte aligned
    if (tid <= ROWS) {
        const int r = row0 + tid;
        s_rp[tid] = a.rowptr_t[r < a.n ? r : a.n];
    }
    __syncthreads();
    const float* qkv = a.qkv[net];
    const float* dagg = a.dagg[net];
    const float* stats = a.stats[net];
    bool cols_in_lds = false;
    const int lo = stage_window(a.col_t, s_rp, ROWS, s_hdr, win_cap, tid, 512, [&](int lo_, int cnt) {
        // k | dagg (| m | Z) of a row; in 8-byte units when every segment starts on an even column (then P, NV and 3 nh
        // are even too: every row of the three arrays is 8-byte aligned)
        const bool pairs = ((nq | vd | NV | nh) & 1) == 0 && (reinterpret_cast<uintptr_t>(win) & 7) == 0 &&
                           ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dagg) | reinterpret_cast<uintptr_t>(stats)) & 7) == 0;
        if (pairs)
            window_copy2(win, WS, cnt, (nq + NV + (SW ? 2 * nh : 0)) >> 1, tid, 512, [&](int rr, int c2) {
                const int64_t r = lo_ + rr;
                const int c = 2 * c2;
                const float* src = c < nq ? qkv + r * P + nq + c : (c < nq + NV ? dagg + r * NV + (c - nq) : stats + r * 3 * nh + (c - nq - NV));
                return *reinterpret_cast<const f32x2_win*>(src);
            });
        else
            window_copy(win, WS, cnt, nq + NV + (SW ? 2 * nh : 0), tid, 512, [&](int rr, int c) {
                const int64_t r = lo_ + rr;
                if (c < nq) return qkv[r * P + nq + c];
                if (c < nq + NV) return dagg[r * NV + (c - nq)];
                return stats[r * 3 * nh + (c - nq - NV)];
            });
    }, s_col, kRowsColCap, &cols_in_lds);
    if (SW && lo >= 0) {  // delta of the window rows: <dagg (staged), agg> per head
        const float* agg = a.agg[net];
        const int cells = (s_hdr[1] - lo + 1) * nh;
        for (int i = tid; i < cells; i += 512) {
            const int rr = i / nh, h = i - rr * nh;
            const float* ap = agg + (int64_t)(lo + rr) * NV + h * vd;
            const float* dp = win + rr * WS + nq + h * vd;
            float d = 0.f;
            for (int j = 0; j < vd; ++j) d += dp[j] * ap[j];
            win[rr * WS + nq + NV + 2 * nh + h] = d;
        }
    }
    __syncthreads();
    const int* cols = cols_in_lds ? s_col : a.col_t;
    const int col_base = cols_in_lds ? s_rp[0] : 0;
    constexpr int PAR = ROWS == 32 ? 2 : 1;
    const int row_l = PAR == 2 ? (lane & 31) : lane, par = PAR == 2 ? (lane >> 5) : 0;
    const int u_ = row0 + row_l;
    float dvp[VDM];
#pragma unroll
    for (int j = 0; j < VDM; ++j) dvp[j] = 0.f;
    if (wave < nh && row_l < ROWS && u_ < a.n) {
        const bool even = ((kq | vd | nq | NV) & 1) == 0;
        if (lo >= 0)
            attn_send_thread<KQM, VDM, true, SW, false, PAR, FX>(a, net, u_, wave, win, lo, WS, cols, col_base,
                                                             even && (reinterpret_cast<uintptr_t>(win) & 7) == 0, dvp, par);
        else
            attn_send_thread<KQM, VDM, false, false, SW, PAR, FX>(a, net, u_, wave, win, 0, WS, cols, col_base,
                                                              even && ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dagg)) & 7) == 0, dvp, par);
    }
    // v is shared by the heads: dv[u, :] = sum over the head waves (through the LDS region the window occupied)
    __syncthreads();
    float* red = win;  // [8][64][VDM]
#pragma unroll
    for (int j = 0; j < VDM; ++j) red[(wave * 64 + lane) * VDM + j] = dvp[j];
    __syncthreads();
    if (lane < ROWS && u_ < a.n)
        for (int j = wave; j < vd; j += 8) {
            float s = 0.f;
            for (int h = 0; h < nh; ++h) s += red[(h * 64 + lane) * VDM + j];
            a.dqkv[net][(int64_t)u_ * P + 2 * nq + j] = s;
        }
}

template <int KQM, int VDM, int ROWS = 64>
__global__ __launch_bounds__(512) void k_attn_bwd_send_rows(const AttnBwdArgs a, int win_cap) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_send_tile<KQM, VDM, ROWS, false>(a, blockIdx.y, blockIdx.x, win_cap, sm);
}

// both passes of both nets in ONE launch: workgroups [0, 2 ts) are the sender tiles (RS rows each; the longer ones, first
// in dispatch order), [2 ts, 2 ts + 2 tr) the receiver tiles (RR rows each).  One workgroup per CU (the LDS window), so
// the launcher picks the tile sizes that make the launch one round of the chip where it can: the config-2 batch (2718
// nodes) is 170 sender tiles of 32 rows + 86 receiver tiles of 64 rows = 256 workgroups.
// FX: 8 heads of kq = KQM, v = VDM (the drivers' defaults, run_grevnet.py:74-76) as compile-time constants
template <int KQM, int VDM, int RS, int RR, bool FX = false>
__global__ __launch_bounds__(512) void k_attn_bwd_edges(const AttnBwdArgs a, int cap_recv, int cap_send, int ts) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x;
    if (b < 2 * ts)
        attn_send_tile<KQM, VDM, RS, true, FX>(a, b & 1, b >> 1, cap_send, sm);
    else
        attn_recv_tile<KQM, VDM, RR, FX>(a, (b - 2 * ts) & 1, (b - 2 * ts) >> 1, cap_recv, sm);
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
    // batch-norm bijector in front of the half-step (NULL bn_part: none): this kernel's rows of g are final, so it also
    // leaves sum G and sum G x^ (x^ = (y - beta) / gamma) over its rows, one [H][2] fp64 row per workgroup
    const float* bn_y;
    int64_t bn_ld;
    const float* bn_gamma;
    const float* bn_beta;
    double* bn_part;
    // the conditioning half as the attention dW GEMMs will read it (it changes under the bijector's backward pass and the
    // next half-step before they run): copied here, by the last kernel of the half-step that sees it intact (NULL: no copy)
    const float* xc_src;
    int64_t xc_ld;
    float* xc_dst;  // [n][H]
    const float* wct[2];  // k_attn_bwd_dx_mfma: [Wq | Wk | Wv]^T as packed fragments (attn_wct_floats), else unused
};

static constexpr int kDxRows = 8;  // rows per workgroup

__global__ __launch_bounds__(256) void k_attn_bwd_dx(const AttnDxArgs a) {
    extern __shared__ float sm[];  // Wcat [H][P + 1] (Wq | Wk | Wv of one net) | dqkv rows [kDxRows][P]
    const int P = 2 * a.nq + a.v, P1 = P + 1, H = a.H;
    float* wl = sm;
    float* dl = wl + H * P1;
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * kDxRows;
    const int rows = (int)((a.n - row0) < kDxRows ? (a.n - row0) : kDxRows);
    if (a.xc_dst)
        for (int i = tid; i < rows * H; i += 256) {
            const int rl = i / H, f = i - rl * H;
            a.xc_dst[(row0 + rl) * H + f] = a.xc_src[(row0 + rl) * a.xc_ld + f];
        }
    for (int net = 0; net < 2; ++net) {
        __syncthreads();
        for (int base = 0; base < H * P; base += 256 * 8) {  // loads first, stores after
            float reg[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i0 = base + tid + q * 256;
                const int i = i0 < H * P ? i0 : 0;
                const int f = i / P, c = i - f * P;
                if (c < a.nq)
                    reg[q] = a.Wq[net][(int64_t)f * a.nq + c];
                else if (c < 2 * a.nq)
                    reg[q] = a.Wk[net][(int64_t)f * a.nq + (c - a.nq)];
                else
                    reg[q] = a.Wv[net][(int64_t)f * a.v + (c - 2 * a.nq)];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = base + tid + q * 256;
                if (i < H * P) wl[(i / P) * P1 + (i % P)] = reg[q];
            }
        }
        for (int i = tid; i < rows * P; i += 256) dl[i] = a.dqkv[net][row0 * P + i];
        __syncthreads();
        for (int i = tid; i < rows * H; i += 256) {
            const int rl = i / H, f = i - rl * H;
            const float* d = dl + rl * P;
            const float* w = wl + f * P1;
            float s = 0.f;
#pragma unroll 16
            for (int c = 0; c < P; ++c) s += d[c] * w[c];
            const int64_t r = row0 + rl;
            if (a.concat) s += a.dh0[net][r * a.in0 + f];
            if (a.gst[net]) s += a.gst[net][r * H + f];
            a.g[r * a.ldg + f] += s;
        }
    }
    if (a.bn_part) {  // (the same thread wrote g[r, f] in both passes: it re-reads its own final value)
        float* bg = dl + kDxRows * P;   // [kDxRows][H] G | [kDxRows][H] G x^
        float* bx = bg + kDxRows * H;
        for (int i = tid; i < kDxRows * H; i += 256) {
            const int rl = i / H, f = i - rl * H;
            float gv = 0.f, gx = 0.f;
            if (rl < rows) {
                const int64_t r = row0 + rl;
                gv = a.g[r * a.ldg + f];
                gx = gv * ((a.bn_y[r * a.bn_ld + f] - a.bn_beta[f]) / a.bn_gamma[f]);
            }
            bg[i] = gv;
            bx[i] = gx;
        }
        __syncthreads();
        for (int i = tid; i < 2 * H; i += 256) {
            const int f = i >> 1, which = i & 1;
            const float* src = which ? bx : bg;
            double acc = 0.0;
            for (int rl = 0; rl < kDxRows; ++rl) acc += (double)src[rl * H + f];
            a.bn_part[((int64_t)blockIdx.x * H + f) * 2 + which] = acc;
        }
    }
}

// The same with BOTH nets' operands in LDS at once: every global load of the workgroup (2 x Wcat, 2 x kDxRows dqkv rows,
// g, dh0, gst) is in flight before the first LDS store, then one barrier, one pass over 2 P products per thread and one
// store of g - two round trips to memory instead of ten.  NQ / VD: compile-time heads*kq and v (0 = run-time values; the
// reference's geometry gets an instance whose index arithmetic has no divisions).  Needs kDxRows * H <= 256.
static constexpr int kDx2Loads = 24;  // W elements per thread per net held in registers (256 x 24 >= H * P)
template <int NQ, int VD>
__global__ __launch_bounds__(256) void k_attn_bwd_dx2(const AttnDxArgs a) {
    extern __shared__ float sm[];  // [2] Wcat [H][P + 1] | [2] dqkv rows [kDxRows][P] | bn scratch [2][kDxRows][H]
    const int nq = NQ ? NQ : a.nq, vd = VD ? VD : a.v;
    const int P = 2 * nq + vd, P1 = P + 1, H = a.H;
    float* wl = sm;
    float* dl = wl + 2 * H * P1;
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * kDxRows;
    const int rows = (int)((a.n - row0) < kDxRows ? (a.n - row0) : kDxRows);
    float wreg[2][kDx2Loads], dreg[2][6];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const float *Wq = a.Wq[net], *Wk = a.Wk[net], *Wv = a.Wv[net];
#pragma unroll
        for (int q = 0; q < kDx2Loads; ++q) {
            const int i0 = tid + q * 256;
            const int i = i0 < H * P ? i0 : 0;
            const int f = i / P, c = i - f * P;
            wreg[net][q] = c < nq ? Wq[f * nq + c] : (c < 2 * nq ? Wk[f * nq + (c - nq)] : Wv[f * vd + (c - 2 * nq)]);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int i = tid + q * 256;
            dreg[net][q] = a.dqkv[net][row0 * P + (i < rows * P ? i : 0)];
        }
    }
    const int rl0 = tid / H, f0 = tid - rl0 * H;
    const bool live = tid < rows * H;
    float s0 = 0.f, xc = 0.f;
    if (live) {
        const int64_t r = row0 + rl0;
        s0 = a.g[r * a.ldg + f0];
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            if (a.concat) s0 += a.dh0[net][r * a.in0 + f0];
            if (a.gst[net]) s0 += a.gst[net][r * H + f0];
        }
        if (a.xc_dst) xc = a.xc_src[r * a.xc_ld + f0];
    }
#pragma unroll
    for (int net = 0; net < 2; ++net) {
#pragma unroll
        for (int q = 0; q < kDx2Loads; ++q) {
            const int i = tid + q * 256;
            if (i < H * P) wl[net * H * P1 + (i / P) * P1 + (i % P)] = wreg[net][q];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int i = tid + q * 256;
            if (i < rows * P) dl[net * kDxRows * P + i] = dreg[net][q];
        }
    }
    for (int i = tid + 6 * 256; i < rows * P; i += 256)  // (P > 192: the rest of the rows, plainly)
        for (int net = 0; net < 2; ++net) dl[net * kDxRows * P + i] = a.dqkv[net][row0 * P + i];
    __syncthreads();
    float gv = 0.f;
    if (live) {
        float s = 0.f;
#pragma unroll
        for (int net = 0; net < 2; ++net) {
            const float* d = dl + net * kDxRows * P + rl0 * P;
            const float* w = wl + net * H * P1 + f0 * P1;
            float sn = 0.f;
#pragma unroll 10
            for (int c = 0; c < P; ++c) sn += d[c] * w[c];
            s += sn;
        }
        gv = s0 + s;
        a.g[(row0 + rl0) * a.ldg + f0] = gv;
        if (a.xc_dst) a.xc_dst[(row0 + rl0) * H + f0] = xc;
    }
    if (a.bn_part) {
        float* bg = dl + 2 * kDxRows * P;   // [kDxRows][H] G | [kDxRows][H] G x^
        float* bx = bg + kDxRows * H;
        if (tid < kDxRows * H) {
            float gx = 0.f;
            if (live) gx = gv * ((a.bn_y[(row0 + rl0) * a.bn_ld + f0] - a.bn_beta[f0]) / a.bn_gamma[f0]);
            bg[tid] = live ? gv : 0.f;
            bx[tid] = gx;
        }
        __syncthreads();
        for (int i = tid; i < 2 * H; i += 256) {
            const int f = i >> 1, which = i & 1;
            const float* src = which ? bx : bg;
            double acc = 0.0;
            for (int rl = 0; rl < kDxRows; ++rl) acc += (double)src[rl * H + f];
            a.bn_part[((int64_t)blockIdx.x * H + f) * 2 + which] = acc;
        }
    }
}

// The same product on the matrix cores: a workgroup = 16 rows, wave = (net, column tile of H): out = dqkv rows [16 x P] x
// Wcat^T [P x H] with the weights as pre-packed fragments (one 16-byte load per lane and k-group, all requested before the
// first MFMA), the dqkv rows staged in LDS with coalesced loads; then the same epilogue as k_attn_bwd_dx2 (g, the copy of
// the conditioning half, the batch-norm partial sums - one [H][2] fp64 row per workgroup of SIXTEEN rows here).
// 14.7 -> 9 us per launch on the config-2 batch (4.5 of them the launch itself): k_attn_bwd_dx2 re-reads both nets' 43 KB
// of weights element by element in every 8-row workgroup and walks 2 x 170 LDS products per thread.
static constexpr int kDxmRows = 16;
static constexpr int kDxmMaxKG = 12;  // k-groups of fragments a wave holds in registers at a time
typedef float f32x4_dx __attribute__((ext_vector_type(4)));
size_t attn_wct_floats(const GnfAttn* at, int32_t H) {
    if (!at) return 0;
    const size_t P = 2 * (size_t)at->num_heads * at->kq_dim + at->v_dim;
    return ((P + 15) & ~(size_t)15) * (((size_t)H + 15) & ~(size_t)15);
}
__global__ __launch_bounds__(256) void k_attn_bwd_dx_mfma(const AttnDxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int P = 2 * a.nq + a.v, H = a.H;
    const int Pp = (P + 15) & ~15, Hp = (H + 15) & ~15, kgs = Pp >> 4, nts = Hp >> 4;
    const int lda = Pp + 4, ldo = Hp + 4;
    float* A = sm;                              // [2][16][lda]
    float* out = A + 2 * kDxmRows * lda;        // [2][16][ldo]
    float* bg = out + 2 * kDxmRows * ldo;       // [16][H] G | [16][H] G x^
    float* bx = bg + kDxmRows * H;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lrow = lane & 15, lgrp = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * kDxmRows;
    const int rows = (int)((a.n - row0) < kDxmRows ? (a.n - row0) : kDxmRows);
    // Every global read of the workgroup is requested before the first one is consumed (one round trip to memory instead
    // of five): the weight fragments of this wave's first tile, the dqkv rows, and the epilogue's operands of the
    // thread's first two elements (all of them when H <= 32).
    // (buffer loads: one descriptor per array, a 32-bit VGPR offset per load - with 64-bit address arithmetic per load
    // the ISSUE of these ~50 loads alone took 7 k cycles; rows past the batch / columns past P come back as zeros by the
    // descriptor's range check)
    f32x4_dx b0[kDxmMaxKG];
    const int net0 = wave / nts, nt0 = wave - net0 * nts;  // (idx = wave: the first (net, column tile) of this wave)
    if (wave < 2 * nts) {
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(net0 ? a.wct[1] : a.wct[0]), 0,
                                                                             Pp * Hp * 4, 0x00020000);
        const int vo = (nt0 * 256 + lane * 4) * 4, kstride = nts * 1024;
#pragma unroll
        for (int u = 0; u < kDxmMaxKG; ++u)
            b0[u] = __builtin_bit_cast(f32x4_dx, __builtin_amdgcn_raw_buffer_load_b128(wr, vo + (u < kgs ? u : kgs - 1) * kstride, 0, 0));
    }
    // (thread = column, a loop over the 2 x 16 rows: no integer divisions in the index arithmetic - with i / Pp per element
    // this phase took 19 k cycles of the kernel's 26 k)
    float av_[2 * kDxmRows];
    {
        const int live_bytes = rows * P * 4;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dqkv[0] + row0 * P), 0, live_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dqkv[1] + row0 * P), 0, live_bytes, 0x00020000);
        const int vcol = tid < P ? tid * 4 : 0x7ffffff0;  // (out of range: zero)
#pragma unroll
        for (int rl = 0; rl < kDxmRows; ++rl) {
            const int vo = tid < P ? vcol + rl * P * 4 : vcol;
            av_[rl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, vo, 0, 0));
            av_[kDxmRows + rl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, vo, 0, 0));
        }
    }
    float e_s0[2] = {0.f, 0.f}, e_y[2] = {0.f, 0.f}, e_xc[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;
        const int rl = i / H, f = i - rl * H;
        if (i < kDxmRows * H && rl < rows) {
            const int64_t r = row0 + rl;
            float s0 = a.g[r * a.ldg + f];
#pragma unroll
            for (int net = 0; net < 2; ++net) {
                if (a.concat) s0 += a.dh0[net][r * a.in0 + f];
                if (a.gst[net]) s0 += a.gst[net][r * H + f];
            }
            e_s0[u] = s0;
            if (a.bn_part) e_y[u] = a.bn_y[r * a.bn_ld + f];
            if (a.xc_dst) e_xc[u] = a.xc_src[r * a.xc_ld + f];
        }
    }
    if (tid < Pp) {
#pragma unroll
        for (int rr = 0; rr < 2 * kDxmRows; ++rr) A[rr * lda + tid] = av_[rr];
    }
    for (int c = tid + 256; c < Pp; c += 256)  // (more than 256 columns: the rest, plainly)
        for (int rr = 0; rr < 2 * kDxmRows; ++rr) {
            const int rl = rr & (kDxmRows - 1);
            A[rr * lda + c] = (c < P && rl < rows) ? (rr >= kDxmRows ? a.dqkv[1] : a.dqkv[0])[(row0 + rl) * P + c] : 0.f;
        }
    __syncthreads();
    // ---- wave = (net, column tile): 4 MFMAs per k-group on one accumulator
    for (int idx = wave; idx < 2 * nts; idx += 4) {
        const int net = idx / nts, nt = idx - net * nts;
        const float* wp = (net ? a.wct[1] : a.wct[0]) + (size_t)nt * 256 + lane * 4;
        const float* ab = A + (net * kDxmRows + lrow) * lda + 4 * lgrp;
        f32x4_dx acc = {0.f, 0.f, 0.f, 0.f};  // (one chain: two interleaved ones measured no faster)
        for (int kg0 = 0; kg0 < kgs; kg0 += kDxmMaxKG) {
            f32x4_dx b[kDxmMaxKG];
#pragma unroll
            for (int u = 0; u < kDxmMaxKG; ++u) {
                if (idx == wave && kg0 == 0) {
                    b[u] = b0[u];
                } else {
                    const int kg = kg0 + u < kgs ? kg0 + u : kgs - 1;
                    b[u] = *reinterpret_cast<const f32x4_dx*>(wp + (size_t)kg * nts * 256);
                }
            }
#pragma unroll
            for (int u = 0; u < kDxmMaxKG; ++u) {
                if (kg0 + u < kgs) {  // wave-uniform
                    const f32x4_dx av = *reinterpret_cast<const f32x4_dx*>(ab + 16 * (kg0 + u));
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], b[u][q], acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(net * kDxmRows + 4 * lgrp + r) * ldo + 16 * nt + lrow] = acc[r];
    }
    __syncthreads();
    // ---- epilogue: g += both nets' products (+ the direct terms), copy of the conditioning half, batch-norm partials
    for (int i = tid, u = 0; i < kDxmRows * H; i += 256, ++u) {
        const int rl = i / H, f = i - rl * H;
        const bool live = rl < rows;
        float gv = 0.f, yv = 0.f;
        if (live) {
            const int64_t r = row0 + rl;
            float s0, xc;
            if (u < 2) {
                s0 = e_s0[u & 1], yv = e_y[u & 1], xc = e_xc[u & 1];
            } else {
                s0 = a.g[r * a.ldg + f];
#pragma unroll
                for (int net = 0; net < 2; ++net) {
                    if (a.concat) s0 += a.dh0[net][r * a.in0 + f];
                    if (a.gst[net]) s0 += a.gst[net][r * H + f];
                }
                yv = a.bn_part ? a.bn_y[r * a.bn_ld + f] : 0.f;
                xc = a.xc_dst ? a.xc_src[r * a.xc_ld + f] : 0.f;
            }
            gv = s0 + (out[rl * ldo + f] + out[(kDxmRows + rl) * ldo + f]);
            a.g[r * a.ldg + f] = gv;
            if (a.xc_dst) a.xc_dst[r * H + f] = xc;
        }
        if (a.bn_part) {
            bg[i] = gv;
            bx[i] = live ? gv * ((yv - a.bn_beta[f]) / a.bn_gamma[f]) : 0.f;
        }
    }
    if (a.bn_part) {
        __syncthreads();
        for (int i = tid; i < 2 * H; i += 256) {
            const int f = i >> 1, which = i & 1;
            const float* src = which ? bx : bg;
            double acc = 0.0;
            for (int rl = 0; rl < kDxmRows; ++rl) acc += (double)src[rl * H + f];
            a.bn_part[((int64_t)blockIdx.x * H + f) * 2 + which] = acc;
        }
    }
}

// at[2]: the two attention blocks; qkv: [2][N, P] forward projections (launch_attn_front's scratch);
// dh0 / gst: per net;  dqkv: per net outputs kept for the dW GEMMs;  dagg = dnew Wo^T [N, heads*v] (computed by the
// caller with the matrix-core GEMM);  agg / stats: the forward pass's attended values and softmax statistics
// (launch_attn_front's agg_out / mz_out; the lane-per-feature kernels for wide heads rebuild both in place).
int launch_attn_backward(const GnfAttn* const* at, int64_t n, int32_t H, int32_t in0, const int32_t* rowptr,
                         const int32_t* col, const int32_t* rowptr_t, const int32_t* col_t, const float* const* qkv,
                         const float* const* dh0, const float* const* gst, float* const* dqkv, float* const* agg,
                         float* const* dagg, float* const* stats, float* g_cond, int64_t ldg, hipStream_t st, int64_t n_edges,
                         const AttnBnFold* bn, const float* xc_src, int64_t xc_ld, float* xc_dst, const float* const* wct) {
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
    const int NV = a.nh * a.v, nq = a.nh * a.kq, P = 2 * nq + a.v;
    const int wmax = nq > NV ? nq : NV;
    if (wmax > kAttnMaxWidth || a.nh > kAttnMaxHeads || !attn_geometry_ok(a.nh, a.kq, a.v, H)) {  // (validate_attn's limit: the entry point has rejected this already)
        set_error("attention backward: heads=%d kq=%d v=%d on H=%d outside heads <= %d, heads*kq <= %d, heads*v <= %d, "
                  "pad16(2 heads kq + v) + pad16(H) + H <= %d", a.nh, a.kq, a.v, H, kAttnMaxHeads, kAttnMaxWidth, kAttnMaxWidth,
                  kAttnMaxRowFloats);
        return GNF_ESHAPE;
    }
    // dL/dx_cond: the matrix-core form (the caller's packed [Wq | Wk | Wv]^T; fits every geometry inside the limit by the
    // limit's last clause), else both nets' whole weights in LDS
    const size_t lds_x = ((size_t)H * (P + 1) + (size_t)kDxRows * P + (size_t)2 * kDxRows * H) * sizeof(float);
    const size_t lds_m = ((size_t)2 * kDxmRows * (((P + 15) & ~15) + 4) + (size_t)2 * kDxmRows * (((H + 15) & ~15) + 4) +
                          (size_t)2 * kDxmRows * H) * sizeof(float);
    const bool dx_mfma = wct && wct[0] && wct[1] && lds_m <= 160 * 1024;
    if (!dx_mfma && lds_x > 160 * 1024) {
        set_error("attention backward: heads=%d kq=%d v=%d on H=%d without the packed [Wq | Wk | Wv]^T needs more LDS than a CU has",
                  a.nh, a.kq, a.v, H);
        return GNF_EUNSUPPORTED;
    }
    GNF_ONCE_PER_DEVICE(
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_dx),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_dx_mfma),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
);
    if (a.nh <= kRowsMaxHeads && a.kq <= 32 && a.v <= 32) {
        const size_t fixed_r = (size_t)(kRowsTile + 1 + 3 + kRowsColCap) * sizeof(int);
        const int capr = (int)((kRowsLdsBudget - fixed_r) / ((size_t)(nq + a.v + 2) * sizeof(float)));
        const int caps = (int)((kRowsLdsBudget - fixed_r) / ((size_t)(nq + NV + 2) * sizeof(float)));
        const int capw = (int)((kRowsLdsBudget - fixed_r) / ((size_t)(nq + NV + 3 * a.nh + 2) * sizeof(float)));  // with statistics
        // the sender pass re-uses the window region for its head reduction: 8 x 64 x VDM floats must fit
        if (capr >= 64 && (size_t)capw * (nq + NV) >= (size_t)8 * 64 * 32) {
            GNF_ONCE_PER_DEVICE(
                const void* ks[18] = {
                    reinterpret_cast<const void*>(k_attn_bwd_recv_rows<10, 10, 64>), reinterpret_cast<const void*>(k_attn_bwd_recv_rows<32, 32, 64>),
                    reinterpret_cast<const void*>(k_attn_bwd_send_rows<10, 10, 64>), reinterpret_cast<const void*>(k_attn_bwd_send_rows<32, 32, 64>),
                    reinterpret_cast<const void*>(k_attn_bwd_recv_rows<10, 10, 32>), reinterpret_cast<const void*>(k_attn_bwd_recv_rows<32, 32, 32>),
                    reinterpret_cast<const void*>(k_attn_bwd_send_rows<10, 10, 32>), reinterpret_cast<const void*>(k_attn_bwd_send_rows<32, 32, 32>),
                    reinterpret_cast<const void*>(k_attn_bwd_recv_rows<10, 10, 16>), reinterpret_cast<const void*>(k_attn_bwd_recv_rows<32, 32, 16>),
                    reinterpret_cast<const void*>(k_attn_bwd_send_rows<10, 10, 16>), reinterpret_cast<const void*>(k_attn_bwd_send_rows<32, 32, 16>),
                    reinterpret_cast<const void*>(k_attn_bwd_edges<10, 10, 64, 64>), reinterpret_cast<const void*>(k_attn_bwd_edges<32, 32, 64, 64>),
                    reinterpret_cast<const void*>(k_attn_bwd_edges<10, 10, 32, 32>), reinterpret_cast<const void*>(k_attn_bwd_edges<32, 32, 32, 32>),
                    reinterpret_cast<const void*>(k_attn_bwd_edges<10, 10, 32, 64>), reinterpret_cast<const void*>(k_attn_bwd_edges<32, 32, 32, 64>)};
                for (const void* k : ks) GNF_HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
            // Tile sizes.  Two launches (dense batches): 32-row tiles on sparse batches (default-flags
            // training step on the config-2 batch: 4.44 (64) / 4.24 (32) / 4.74 ms (16: the window staging per workgroup
            // takes over)) and on any batch whose 64-row tiles would be fewer workgroups than the chip has CUs (the drivers'
            // default batch, 32 complete 100-node graphs: 10.25 -> 9.9 ms per iteration of examples/run_grevnet.py).
            // One launch (sparse batches): the first of (32, 32), (32 sender / 64 receiver), (64, 64) that is at most one
            // workgroup per CU; beyond that 32-row tiles.  Option attn_bwd_rows forces a size (16: two launches).
            const bool sparse = n_edges > 0 && n_edges < 24 * n;
            int rows = (sparse || 2 * ((n + 63) / 64) < 256) ? 32 : 64;
            const int64_t force = opt(OPT_ATTN_BWD_ROWS);
            if (force == 64 || force == 32 || force == 16) rows = (int)force;
            const bool one_launch = sparse && rows != 16;
            const bool small = a.kq <= 10 && a.v <= 10;
            if (one_launch) {
                const int64_t t32 = (n + 31) / 32, t64 = (n + 63) / 64;
                int rs = rows, rr = rows;
                if (!force) {
                    if (4 * t32 <= 256) rs = rr = 32;
                    else if (2 * t32 + 2 * t64 <= 256) rs = 32, rr = 64;
                    else if (4 * t64 <= 256) rs = rr = 64;
                } else if (force == 3264) {
                    rs = 32, rr = 64;
                }
                const int ts = (int)((n + rs - 1) / rs), tr = (int)((n + rr - 1) / rr);
                const dim3 egrid((unsigned)(2 * ts + 2 * tr));
                auto go1 = [&](auto rs_c, auto rr_c) {
                    constexpr int RS = decltype(rs_c)::value, RR = decltype(rr_c)::value;
                    if (a.nh == 8 && a.kq == 10 && a.v == 10 && RS == 32 && RR == 64)
                        hipLaunchKernelGGL((k_attn_bwd_edges<10, 10, RS, RR, true>), egrid, dim3(512), (size_t)kRowsLdsBudget, st, a, capr, capw, ts);
                    else if (small)
                        hipLaunchKernelGGL((k_attn_bwd_edges<10, 10, RS, RR>), egrid, dim3(512), (size_t)kRowsLdsBudget, st, a, capr, capw, ts);
                    else
                        hipLaunchKernelGGL((k_attn_bwd_edges<32, 32, RS, RR>), egrid, dim3(512), (size_t)kRowsLdsBudget, st, a, capr, capw, ts);
                };
                if (rs == 32 && rr == 64)
                    go1(std::integral_constant<int, 32>{}, std::integral_constant<int, 64>{});
                else if (rs == 32)
                    go1(std::integral_constant<int, 32>{}, std::integral_constant<int, 32>{});
                else
                    go1(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{});
                GNF_LAUNCH_CHECK("k_attn_bwd_edges");
                goto dx_pass;
            }
            const dim3 rgrid((unsigned)((n + rows - 1) / rows), 2);
            auto go = [&](auto rows_c) {
                constexpr int R = decltype(rows_c)::value;
                if (small) {
                    hipLaunchKernelGGL((k_attn_bwd_recv_rows<10, 10, R>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, capr);
                    hipLaunchKernelGGL((k_attn_bwd_send_rows<10, 10, R>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, caps);
                } else {
                    hipLaunchKernelGGL((k_attn_bwd_recv_rows<32, 32, R>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, capr);
                    hipLaunchKernelGGL((k_attn_bwd_send_rows<32, 32, R>), rgrid, dim3(512), (size_t)kRowsLdsBudget, st, a, caps);
                }
            };
            if (rows == 16)
                go(std::integral_constant<int, 16>{});
            else if (rows == 32)
                go(std::integral_constant<int, 32>{});
            else
                go(std::integral_constant<int, 64>{});
            GNF_LAUNCH_CHECK("k_attn_bwd_recv_rows / k_attn_bwd_send_rows");
            goto dx_pass;
        }
    }
    {
    // wide heads (the data driver's one head of 64 / 64, train_grevnet_with_data.py:40-46): both passes on the matrix cores
    // (gnf_attn_core_bwd.hip), from the statistics and attended values the forward pass left
    const int rc_core = launch_attn_core_backward(a0, n, rowptr, col, rowptr_t, col_t, qkv, dagg, agg, stats, dqkv, st);
    if (rc_core != GNF_OK) {
        if (rc_core == 1) {   // (kq or v above 256: outside validate_attn's limit, the entry point has rejected it)
            set_error("attention backward: heads=%d kq=%d v=%d has no kernel", a.nh, a.kq, a.v);
            return GNF_EUNSUPPORTED;
        }
        return rc_core;
    }
    }
dx_pass:
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
    d.xc_src = xc_src, d.xc_ld = xc_ld, d.xc_dst = xc_dst;
    d.wct[0] = wct ? wct[0] : nullptr, d.wct[1] = wct ? wct[1] : nullptr;
    {   // the matrix-core form when the caller has the packed weights
        const int64_t blocks_m = (n + kDxmRows - 1) / kDxmRows;
        if (dx_mfma) {
            d.bn_y = nullptr, d.bn_ld = 0, d.bn_gamma = d.bn_beta = nullptr, d.bn_part = nullptr;
            if (bn && bn->n_parts) *bn->n_parts = 0;
            if (bn && bn->part && blocks_m <= kBnPartRowsMax) {
                d.bn_y = bn->y, d.bn_ld = bn->ld, d.bn_gamma = bn->gamma, d.bn_beta = bn->beta, d.bn_part = bn->part;
                if (bn->n_parts) *bn->n_parts = (int32_t)blocks_m;
            }
            hipLaunchKernelGGL(k_attn_bwd_dx_mfma, dim3((unsigned)blocks_m), dim3(256), lds_m, st, d);
            GNF_LAUNCH_CHECK("k_attn_bwd_dx_mfma");
            return GNF_OK;
        }
    }
    const int64_t dx_blocks = (n + kDxRows - 1) / kDxRows;
    d.bn_y = nullptr, d.bn_ld = 0, d.bn_gamma = d.bn_beta = nullptr, d.bn_part = nullptr;
    if (bn && bn->n_parts) *bn->n_parts = 0;
    if (bn && bn->part && dx_blocks <= kBnPartRowsMax) {
        d.bn_y = bn->y, d.bn_ld = bn->ld, d.bn_gamma = bn->gamma, d.bn_beta = bn->beta, d.bn_part = bn->part;
        if (bn->n_parts) *bn->n_parts = (int32_t)dx_blocks;
    }
    const size_t lds_x2 = (2 * ((size_t)H * (P + 1) + (size_t)kDxRows * P) + (size_t)2 * kDxRows * H) * sizeof(float);
    if (kDxRows * H <= 256 && H * P <= 256 * kDx2Loads && lds_x2 <= 64 * 1024) {
        if (d.nq == 80 && d.v == 10)  // the reference's head geometry (run_grevnet.py:74-76)
            hipLaunchKernelGGL((k_attn_bwd_dx2<80, 10>), dim3((unsigned)dx_blocks), dim3(256), lds_x2, st, d);
        else
            hipLaunchKernelGGL((k_attn_bwd_dx2<0, 0>), dim3((unsigned)dx_blocks), dim3(256), lds_x2, st, d);
    } else {
        hipLaunchKernelGGL(k_attn_bwd_dx, dim3((unsigned)dx_blocks), dim3(256), lds_x, st, d);
    }
    GNF_LAUNCH_CHECK("k_attn_bwd_dx");
    return GNF_OK;
}

}  // namespace gnf
