// Backward of the matrix-core attention core (forward: gnf_attn_core.hip; reference gnn.py:385-477 under tf.gradients) for the
// head geometries the per-(row, head) thread kernels of gnf_attn_bwd.hip do not hold - first of all the DATA driver's default,
// ONE head with kq = v = 64 on complete graphs (train_grevnet_with_data.py:40-46), where the lane-per-feature kernels took
// 639 + 366 us per half-step.  With the forward quantities of gnf_attn_bwd.hip's header and the statistics the forward pass
// leaves per (receiver, head) - running max m, denominator Z, attended values O:
//   P[s, r]   = mult[r, s] exp(scale <q_s, k_r> - m_r) / Z_r          the softmax weights, rebuilt on the matrix cores
//   dW[s, r]  = < v_s, dO_r >                                         [senders x v] x [v x receivers]
//   delta_r   = sum_s P dW = < dO_r, O_r >                            (no sweep of its own)
//   dS[s, r]  = P (dW - delta_r)
//   dk_r = scale sum_s dS q_s          receiver side (k_attn_core_bwd_recv: a tile of 64 receivers, chunks of its sender window)
//   dq_s = scale sum_r dS k_r,  dv_s = sum_h sum_r P dO_r            sender side (k_attn_core_bwd_send: a tile of 64 senders,
//                                                                     chunks of its receiver window, by-sender CSR; no atomics)
// Both kernels are the forward kernel's shape: the tile's own rows are MFMA second operands held in registers (a lane
// owns ONE node of the tile, lane & 15, and four nodes of the other side per 16-node tile), the other side's rows sit
// TRANSPOSED in LDS ([component][node]) so that the two accumulating products read 16-byte first operands along the node
// axis and the two logit-side products four 4-byte ones; P and dS never leave the registers.
#include "gnf_attn_core_dev.h"

namespace gnf {

struct AttnCoreBwdArgs {
    const float* qkv[2];   // [N, P] q | k | v
    const float* dagg[2];  // [N, nh v]  dL/d(attended values)
    const float* agg[2];   // [N, nh v]  attended values of the forward pass
    float* stats[2];       // [N, 3 nh]  m | Z | delta (delta: written by the receiver pass, read by the sender pass)
    float* dqkv[2];        // [N, P]     dq | dk | dv
    const int32_t* rowptr;    // by receiver (receiver pass) or by sender (sender pass)
    const int32_t* col;
    int32_t n, nh, kq, v;
    float scale;
};

static constexpr int kCbRows = 64;

// ---- receiver side ------------------------------------------------------------------------------------------------------
template <int KG, int VT, int ST>
__global__ __launch_bounds__(256) void k_attn_core_bwd_recv(const AttnCoreBwdArgs a) {
    constexpr int CH = 16 * ST, VS = CH + 4, MW = CH / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qt = sm;                                                        // [16 KG][VS]  q of the chunk's senders, transposed
    float* vt = qt + 16 * KG * VS;                                         // [16 VT][VS]  v likewise
    unsigned* mult = reinterpret_cast<unsigned*>(vt + 16 * VT * VS);       // [64][MW]
    int* s_rp = reinterpret_cast<int*>(mult + kCbRows * MW);
    int* s_hdr = s_rp + kCbRows + 1;
    int* s_col = s_hdr + 3;                                                // [kCap] the tile's slice of col
    constexpr int kCap = core_col_cap<KG>();
    const int net = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lgrp = lane >> 4;
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const int row0 = blockIdx.x * kCbRows;
    const float* __restrict__ qkv = a.qkv[net];
    if (tid <= kCbRows) {
        const int r = row0 + tid;
        s_rp[tid] = a.rowptr[r < a.n ? r : a.n];
    }
    if (tid == 0) s_hdr[0] = 0x7fffffff, s_hdr[1] = -1, s_hdr[2] = 0;
    __syncthreads();
    core_window_scan(a.col, s_rp[0], s_rp[kCbRows], s_hdr, tid, lane, s_col, kCap);
    __syncthreads();
    const bool col_kept = s_rp[kCbRows] - s_rp[0] <= kCap;
    const int32_t* cols = col_kept ? s_col : a.col;
    const int col_base = col_kept ? s_rp[0] : 0;
    const int win_lo = s_hdr[0], win_n = s_hdr[1] >= s_hdr[0] ? s_hdr[1] - s_hdr[0] + 1 : 0;
    const int n_chunks = (win_n + CH - 1) / CH;
    const bool vec4 = ((kq | vd | P) & 3) == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0;
    const int r = row0 + 16 * wave + lrow, my_row = 16 * wave + lrow;
    const bool live = r < a.n;
    for (int h = 0; h < nh; ++h) {
        f32x4 kB[KG], dB[VT];
        float dpart = 0.f;
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * g + 4 * lgrp + q;
                kB[g][q] = (live && j < kq) ? qkv[(int64_t)r * P + nq + h * kq + j] : 0.f;
            }
#pragma unroll
        for (int g = 0; g < VT; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * g + 4 * lgrp + q;
                const bool ok = live && j < vd;
                dB[g][q] = ok ? a.dagg[net][(int64_t)r * NV + h * vd + j] : 0.f;
                dpart += ok ? dB[g][q] * a.agg[net][(int64_t)r * NV + h * vd + j] : 0.f;
            }
        dpart += __shfl_xor(dpart, 16, 64);
        dpart += __shfl_xor(dpart, 32, 64);
        const float delta = dpart;
        float* st = a.stats[net] + (int64_t)(live ? r : 0) * 3 * nh;
        const float m = live ? st[h] : 0.f, rz = live ? 1.f / st[nh + h] : 0.f;
        if (live && lgrp == 0) st[2 * nh + h] = delta;   // for the sender pass
        f32x4 dK[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) dK[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int c0 = ch * CH, cn = win_n - c0 < CH ? win_n - c0 : CH;
            const bool restage_shared = h == 0 || n_chunks > 1;
            __syncthreads();
            core_stage_t<KG, CH>(qt, qkv + h * kq, P, kq, win_lo + c0, cn, tid, vec4);
            if (restage_shared) {
                core_stage_t<VT, CH>(vt, qkv + 2 * nq, P, vd, win_lo + c0, cn, tid, vec4);
                for (int i = tid; i < kCbRows * MW; i += 256) mult[i] = 0u;
            }
            __syncthreads();
            if (restage_shared) core_scatter_mult<CH>(mult, s_rp, cols, col_base, win_lo, c0, s_hdr, tid);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                if (16 * t < cn) {
                    const f32x4 S = core_dot_tile<KG>(qt, VS, t, lrow, lgrp, kq, kB);    // S^T[sender 16 t + 4 lgrp + i][receiver lrow]
                    const f32x4 dW = core_dot_tile<VT>(vt, VS, t, lrow, lgrp, vd, dB);
                    const unsigned* mw = mult + my_row * MW + 8 * t + 2 * lgrp;
                    const unsigned w0 = mw[0], w1 = mw[1];
                    const float ml[4] = {(float)(w0 & 0xffffu), (float)(w0 >> 16), (float)(w1 & 0xffffu), (float)(w1 >> 16)};
                    f32x4 dS;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float p = ml[i] > 0.f ? ml[i] * __expf(S[i] * a.scale - m) * rz : 0.f;
                        dS[i] = p * (dW[i] - delta);
                    }
                    core_acc_tile<KG>(qt, VS, t, lrow, lgrp, kq, dS, dK);
                }
            }
        }
        if (live) {
            const bool poisoned = s_hdr[2] != 0;
            float* __restrict__ out = a.dqkv[net] + (int64_t)r * P + nq + h * kq;
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 16 * g + 4 * lgrp + i;
                    if (j < kq) out[j] = poisoned ? NAN : dK[g][i] * a.scale;
                }
        }
    }
}

// ---- sender side --------------------------------------------------------------------------------------------------------
template <int KG, int VT, int ST>
__global__ __launch_bounds__(256) void k_attn_core_bwd_send(const AttnCoreBwdArgs a) {
    constexpr int CH = 16 * ST, VS = CH + 4, MW = CH / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* kt = sm;                                                        // [16 KG][VS]  k of the chunk's receivers, transposed
    float* dt = kt + 16 * KG * VS;                                         // [16 VT][VS]  their dO rows (head h), transposed
    float* stl = dt + 16 * VT * VS;                                        // [3][CH]      m | 1 / Z | delta of head h
    unsigned* mult = reinterpret_cast<unsigned*>(stl + 3 * CH);            // [64][MW]
    int* s_rp = reinterpret_cast<int*>(mult + kCbRows * MW);
    int* s_hdr = s_rp + kCbRows + 1;
    int* s_col = s_hdr + 3;                                                // [kCap] the tile's slice of col
    constexpr int kCap = core_col_cap<KG>();
    const int net = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, lgrp = lane >> 4;
    const int nh = a.nh, kq = a.kq, vd = a.v, nq = nh * kq, P = 2 * nq + vd, NV = nh * vd;
    const int row0 = blockIdx.x * kCbRows;
    const float* __restrict__ qkv = a.qkv[net];
    const float* __restrict__ dagg = a.dagg[net];
    if (tid <= kCbRows) {
        const int r = row0 + tid;
        s_rp[tid] = a.rowptr[r < a.n ? r : a.n];
    }
    if (tid == 0) s_hdr[0] = 0x7fffffff, s_hdr[1] = -1, s_hdr[2] = 0;
    __syncthreads();
    core_window_scan(a.col, s_rp[0], s_rp[kCbRows], s_hdr, tid, lane, s_col, kCap);
    __syncthreads();
    const bool col_kept = s_rp[kCbRows] - s_rp[0] <= kCap;
    const int32_t* cols = col_kept ? s_col : a.col;
    const int col_base = col_kept ? s_rp[0] : 0;
    const int win_lo = s_hdr[0], win_n = s_hdr[1] >= s_hdr[0] ? s_hdr[1] - s_hdr[0] + 1 : 0;
    const int n_chunks = (win_n + CH - 1) / CH;
    const bool vec4k = ((kq | P) & 3) == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0;
    const bool vec4d = ((vd | NV) & 3) == 0 && (reinterpret_cast<uintptr_t>(dagg) & 15) == 0;
    const int u_ = row0 + 16 * wave + lrow, my_row = 16 * wave + lrow;
    const bool live = u_ < a.n;
    f32x4 vB[VT], dV[VT];
#pragma unroll
    for (int g = 0; g < VT; ++g) {
        dV[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 16 * g + 4 * lgrp + q;
            vB[g][q] = (live && j < vd) ? qkv[(int64_t)u_ * P + 2 * nq + j] : 0.f;
        }
    }
    for (int h = 0; h < nh; ++h) {
        f32x4 qB[KG], dQ[KG];
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            dQ[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 16 * g + 4 * lgrp + q;
                qB[g][q] = (live && j < kq) ? qkv[(int64_t)u_ * P + h * kq + j] : 0.f;
            }
        }
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int c0 = ch * CH, cn = win_n - c0 < CH ? win_n - c0 : CH;
            const bool restage_shared = h == 0 || n_chunks > 1;
            __syncthreads();
            core_stage_t<KG, CH>(kt, qkv + nq + h * kq, P, kq, win_lo + c0, cn, tid, vec4k);
            core_stage_t<VT, CH>(dt, dagg + h * vd, NV, vd, win_lo + c0, cn, tid, vec4d);
            if (tid < CH) {   // the receivers' softmax statistics of head h
                const bool ok = tid < cn;
                const float* st = a.stats[net] + (int64_t)(win_lo + c0 + (ok ? tid : 0)) * 3 * nh;
                stl[tid] = ok ? st[h] : 0.f;
                stl[CH + tid] = ok ? 1.f / st[nh + h] : 0.f;
                stl[2 * CH + tid] = ok ? st[2 * nh + h] : 0.f;
            }
            if (restage_shared)
                for (int i = tid; i < kCbRows * MW; i += 256) mult[i] = 0u;
            __syncthreads();
            if (restage_shared) core_scatter_mult<CH>(mult, s_rp, cols, col_base, win_lo, c0, s_hdr, tid);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                if (16 * t < cn) {
                    const f32x4 S = core_dot_tile<KG>(kt, VS, t, lrow, lgrp, kq, qB);    // S[receiver 16 t + 4 lgrp + i][sender lrow]
                    const f32x4 dW = core_dot_tile<VT>(dt, VS, t, lrow, lgrp, vd, vB);
                    const unsigned* mw = mult + my_row * MW + 8 * t + 2 * lgrp;
                    const unsigned w0 = mw[0], w1 = mw[1];
                    const float ml[4] = {(float)(w0 & 0xffffu), (float)(w0 >> 16), (float)(w1 & 0xffffu), (float)(w1 >> 16)};
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(stl + 16 * t + 4 * lgrp);
                    const f32x4 z4 = *reinterpret_cast<const f32x4*>(stl + CH + 16 * t + 4 * lgrp);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(stl + 2 * CH + 16 * t + 4 * lgrp);
                    f32x4 pw, dS;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pw[i] = ml[i] > 0.f ? ml[i] * __expf(S[i] * a.scale - m4[i]) * z4[i] : 0.f;
                        dS[i] = pw[i] * (dW[i] - d4[i]);
                    }
                    core_acc_tile<KG>(kt, VS, t, lrow, lgrp, kq, dS, dQ);
                    core_acc_tile<VT>(dt, VS, t, lrow, lgrp, vd, pw, dV);
                }
            }
        }
        if (live) {
            const bool poisoned = s_hdr[2] != 0;
            float* __restrict__ out = a.dqkv[net] + (int64_t)u_ * P + h * kq;
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 16 * g + 4 * lgrp + i;
                    if (j < kq) out[j] = poisoned ? NAN : dQ[g][i] * a.scale;
                }
        }
    }
    if (live) {
        float* __restrict__ out = a.dqkv[net] + (int64_t)u_ * P + 2 * nq;
#pragma unroll
        for (int g = 0; g < VT; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = 16 * g + 4 * lgrp + i;
                if (j < vd) out[j] = dV[g][i];
            }
    }
}

template <int KG, int VT, int ST>
static size_t core_bwd_lds_bytes() {
    constexpr int CH = 16 * ST;
    return ((size_t)16 * (KG + VT) * (CH + 4) + 3 * (size_t)CH) * sizeof(float) + (size_t)kCbRows * (CH / 2 + 1) * sizeof(unsigned) +
           (size_t)(kCbRows + 1 + 3 + core_col_cap<KG>()) * sizeof(int);
}

// dq | dk | dv of both nets from dagg, the forward's q | k | v, attended values and statistics (stats' third block is
// written here).  Every geometry the thread-per-(row, head) kernels do not take comes here (heads above 8 or kq / v above 32 -
// many narrow heads too: each head is padded to whole 16-wide k-groups, one head at a time); the probabilities are rebuilt with
// __expf from the forward's (max, Z) - the rows / edge-tiled forward kernels formed Z with expf, the core forward with __expf:
// a relative 1e-7 per weight either way, inside the gradient pins of tests/test_data_driver_gpu.py for both.  1 = kq or v above
// 256 (outside validate_attn's limit: no kernel).
int launch_attn_core_backward(const GnfAttn* a0, int64_t n, const int32_t* rowptr, const int32_t* col, const int32_t* rowptr_t,
                              const int32_t* col_t, const float* const* qkv, const float* const* dagg, const float* const* agg,
                              float* const* stats, float* const* dqkv, hipStream_t st) {
    if (n == 0) return GNF_OK;
    if (a0->kq_dim > 256 || a0->v_dim > 256) return 1;
    AttnCoreBwdArgs a;
    for (int q = 0; q < 2; ++q) a.qkv[q] = qkv[q], a.dagg[q] = dagg[q], a.agg[q] = agg[q], a.stats[q] = stats[q], a.dqkv[q] = dqkv[q];
    a.n = (int32_t)n, a.nh = a0->num_heads, a.kq = a0->kq_dim, a.v = a0->v_dim;
    a.scale = a0->kq_dim_division ? 1.f / sqrtf((float)a0->kq_dim) : 1.f;
    const dim3 grid((unsigned)((n + kCbRows - 1) / kCbRows), 2);
    auto go = [&](auto kr, auto ks, size_t lds) -> int {
        a.rowptr = rowptr, a.col = col;
        hipLaunchKernelGGL(kr, grid, dim3(256), lds, st, a);
        GNF_LAUNCH_CHECK("k_attn_core_bwd_recv");
        a.rowptr = rowptr_t, a.col = col_t;
        hipLaunchKernelGGL(ks, grid, dim3(256), lds, st, a);
        GNF_LAUNCH_CHECK("k_attn_core_bwd_send");
        return GNF_OK;
    };
    if (a.kq <= 64 && a.v <= 64) {
        GNF_ONCE_PER_DEVICE(
            GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_core_bwd_recv<4, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_core_bwd_send<4, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        return go(k_attn_core_bwd_recv<4, 4, 8>, k_attn_core_bwd_send<4, 4, 8>, core_bwd_lds_bytes<4, 4, 8>());
    }
    GNF_ONCE_PER_DEVICE(
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_core_bwd_recv<16, 16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_core_bwd_send<16, 16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    return go(k_attn_core_bwd_recv<16, 16, 4>, k_attn_core_bwd_send<16, 16, 4>, core_bwd_lds_bytes<16, 16, 4>());
}

}  // namespace gnf
