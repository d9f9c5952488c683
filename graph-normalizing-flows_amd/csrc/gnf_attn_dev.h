// Device helpers shared by the attention kernels (gnf_attn.hip, gnf_attn_bwd.hip): the LDS row window and the
// thread = (row, head) kernels' common geometry.
#pragma once
#include "gnf_common.h"

namespace gnf {

// THE ROW WINDOW.  The rows a workgroup's edges point at (senders for a receiver-side pass, receivers for the
// sender-side pass) lie in a narrow node range - at most the graphs its rows belong to.  When that range fits the
// LDS budget the needed columns of those rows are staged ONCE per workgroup and the edge loops read LDS.
// s_rp: rowptr slice [nrows + 1] already in LDS; s_hdr: 2 ints of LDS scratch.  stage_row(lo, count) copies the
// rows (all threads call it; no barriers inside).  Returns the first node of the window (>= 0) or -1 (range too wide
// or no edges): block-uniform.
// s_col != NULL: the same pass also leaves the slice in LDS (when it fits col_cap ints: *cols_in_lds, block-uniform) -
// what stage_cols would re-read from memory in a second round trip.
template <typename StageRow>
__device__ __forceinline__ int stage_window(const int32_t* __restrict__ col, const int* s_rp, int nrows, int* s_hdr,
                                            int win_cap, int tid, int nthr, StageRow stage_row, int* s_col = nullptr,
                                            int col_cap = 0, bool* cols_in_lds = nullptr) {
    if (tid == 0) {
        s_hdr[0] = 0x7fffffff;
        s_hdr[1] = -1;
    }
    __syncthreads();
    const int e0 = s_rp[0], e1 = s_rp[nrows];
    const bool keep = s_col != nullptr && e1 - e0 <= col_cap;
    if (cols_in_lds) *cols_in_lds = keep;
    int lo = 0x7fffffff, hi = -1;
    for (int base = e0; base < e1; base += nthr * 8) {  // eight loads in flight per thread (a load-compare loop is one
        int reg[8];                                     // memory round trip per iteration)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = base + tid + q * nthr;
            reg[q] = col[e < e1 ? e : e1 - 1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = base + tid + q * nthr;
            if (keep && e < e1) s_col[e - e0] = reg[q];
            lo = reg[q] < lo ? reg[q] : lo;
            hi = reg[q] > hi ? reg[q] : hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((tid & 63) == 0) {
        atomicMin(&s_hdr[0], lo);
        atomicMax(&s_hdr[1], hi);
    }
    __syncthreads();
    lo = s_hdr[0];
    hi = s_hdr[1];
    if (hi < lo || hi - lo + 1 > win_cap) return -1;
    stage_row(lo, hi - lo + 1);
    __syncthreads();
    return lo;
}

// coalesced copy of `cnt` rows x W columns into an LDS window with row stride WS; src(row, c) returns the element
template <typename Src>
__device__ __forceinline__ void window_copy(float* __restrict__ win, int WS, int cnt, int W, int tid, int nthr, Src src) {
    // (row, column) of a thread's elements advance by a fixed step: two integer divisions per call instead of two per
    // element (36 elements per thread on a 200-row window: the index arithmetic was a third of the staging time)
    const int total = cnt * W;
    int r = tid / W, c = tid - r * W;
    const int dr = nthr / W, dc = nthr - dr * W;
    for (int base = 0; base < total; base += nthr * 8) {  // eight loads in flight per thread, then the stores
        float reg[8];
        int at[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = base + tid + q * nthr < total;
            at[q] = in ? r * WS + c : -1;
            reg[q] = src(in ? r : 0, in ? c : 0);
            r += dr, c += dc;
            if (c >= W) c -= W, ++r;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (at[q] >= 0) win[at[q]] = reg[q];
    }
}

// the same in 8-byte units (W2 float pairs per row; src(row, c2) returns the pair at float column 2 c2): for rows whose
// segments all start on even columns of 8-byte aligned arrays - half the loads and half the index arithmetic
typedef float f32x2_win __attribute__((ext_vector_type(2)));
template <typename Src>
__device__ __forceinline__ void window_copy2(float* __restrict__ win, int WS, int cnt, int W2, int tid, int nthr, Src src) {
    const int total = cnt * W2;
    int r = tid / W2, c = tid - r * W2;
    const int dr = nthr / W2, dc = nthr - dr * W2;
    for (int base = 0; base < total; base += nthr * 8) {
        f32x2_win reg[8];
        int at[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = base + tid + q * nthr < total;
            at[q] = in ? r * WS + 2 * c : -1;
            reg[q] = src(in ? r : 0, in ? c : 0);
            r += dr, c += dc;
            if (c >= W2) c -= W2, ++r;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (at[q] >= 0) *reinterpret_cast<f32x2_win*>(win + at[q]) = reg[q];
    }
}

// the tile's slice of the CSR column array, staged coalesced (a thread walking its own row would otherwise wait one
// memory round trip per edge); returns false when it does not fit (the caller then reads col from global memory)
__device__ __forceinline__ bool stage_cols(const int32_t* __restrict__ col, const int* s_rp, int nrows, int* s_col,
                                           int col_cap, int tid, int nthr) {
    const int e0 = s_rp[0], cnt = s_rp[nrows] - e0;
    if (cnt > col_cap) return false;
    for (int base = 0; base < cnt; base += nthr * 8) {
        int reg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = base + tid + q * nthr;
            reg[q] = col[e0 + (i < cnt ? i : 0)];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = base + tid + q * nthr;
            if (i < cnt) s_col[i] = reg[q];
        }
    }
    return true;
}

// out[0 .. NM) <- p[0 .. cnt), zero beyond.  v2 (block-uniform): p is 8-byte aligned and cnt even -> 8-byte reads (half the
// LDS / memory instructions of the edge loops, which are issue-bound on dense graphs).
typedef float f32x2_attn __attribute__((ext_vector_type(2)));
template <int NM>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int cnt, bool v2, float (&out)[NM]) {
    if (cnt == NM) {  // the width the instance was built for (block-uniform): no per-element predicates
        if (v2) {
#pragma unroll
            for (int j = 0; j + 1 < NM; j += 2) {
                const f32x2_attn t = *reinterpret_cast<const f32x2_attn*>(p + j);
                out[j] = t[0];
                out[j + 1] = t[1];
            }
            if (NM & 1) out[NM - 1] = p[NM - 1];
        } else {
#pragma unroll
            for (int j = 0; j < NM; ++j) out[j] = p[j];
        }
    } else if (v2) {
#pragma unroll
        for (int j = 0; j < NM; j += 2) {
            f32x2_attn t = {0.f, 0.f};
            if (j < cnt) t = *reinterpret_cast<const f32x2_attn*>(p + j);
            out[j] = t[0];
            if (j + 1 < NM) out[j + 1] = t[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NM; ++j) out[j] = j < cnt ? p[j] : 0.f;
    }
}

// thread = (row, head) kernels: wave w = head w, lane = row inside the 64-row tile
static constexpr int kRowsColCap = 6656;                 // ints of LDS for the tile's col slice (64 rows x degree 104)
static constexpr int kRowsTile = 64;
static constexpr int kRowsMaxHeads = 8;                  // 8 waves
static constexpr int kRowsLdsBudget = 159 * 1024;        // dynamic LDS given to every such workgroup

}  // namespace gnf
