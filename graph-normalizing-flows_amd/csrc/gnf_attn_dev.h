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
template <typename StageRow>
__device__ __forceinline__ int stage_window(const int32_t* __restrict__ col, const int* s_rp, int nrows, int* s_hdr,
                                            int win_cap, int tid, int nthr, StageRow stage_row) {
    if (tid == 0) {
        s_hdr[0] = 0x7fffffff;
        s_hdr[1] = -1;
    }
    __syncthreads();
    const int e0 = s_rp[0], e1 = s_rp[nrows];
    int lo = 0x7fffffff, hi = -1;
    for (int e = e0 + tid; e < e1; e += nthr) {
        const int c = col[e];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
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
    for (int base = 0; base < cnt * W; base += nthr * 8) {  // eight loads in flight per thread, then the stores
        float reg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i0 = base + tid + q * nthr;
            const int i = i0 < cnt * W ? i0 : 0;
            reg[q] = src(i / W, i % W);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = base + tid + q * nthr;
            if (i < cnt * W) win[(i / W) * WS + (i % W)] = reg[q];
        }
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

// thread = (row, head) kernels: wave w = head w, lane = row inside the 64-row tile
static constexpr int kRowsColCap = 6656;                 // ints of LDS for the tile's col slice (64 rows x degree 104)
static constexpr int kRowsTile = 64;
static constexpr int kRowsMaxHeads = 8;                  // 8 waves
static constexpr int kRowsLdsBudget = 159 * 1024;        // dynamic LDS given to every such workgroup

}  // namespace gnf
