// Device helpers shared by the matrix-core attention kernels (gnf_attn_core.hip, gnf_attn_core_bwd.hip): the tile's window
// scan, the multiplicity table, the transposed LDS slabs and the two MFMA products every one of those kernels is made of.
#pragma once
#include "gnf_attn_dev.h"
#include "gnf_fused_dev.h"

namespace gnf {

// ints of LDS for the tile's slice of col (64 rows x mean degree 64; the wide-head instances have less room)
template <int KG>
constexpr int core_col_cap() { return KG <= 4 ? 4096 : 1024; }

// sender / receiver window of the tile's edges: min / max of its slice of col -> s_hdr[0], s_hdr[1] (caller: barriers around).
// The same pass leaves the slice in LDS (s_col, when it fits cap ints): the multiplicity scatter of every chunk then reads it
// there instead of making three more dependent trips to memory per chunk.
__device__ __forceinline__ void core_window_scan(const int32_t* __restrict__ col, int e0, int e1, int* s_hdr, int tid, int lane,
                                                 int* s_col, int cap) {
    const bool keep = e1 - e0 <= cap;
    int lo = 0x7fffffff, hi = -1;
    for (int base = e0; base < e1; base += 256 * 16) {  // sixteen loads in flight per thread: one round trip for 64 rows of degree 64
        int reg[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = base + tid + 256 * u;
            reg[u] = col[e < e1 ? e : e1 - 1];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = base + tid + 256 * u;
            if (keep && e < e1) s_col[e - e0] = reg[u];
            lo = reg[u] < lo ? reg[u] : lo;
            hi = reg[u] > hi ? reg[u] : hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0 && e1 > e0) {
        atomicMin(&s_hdr[0], lo);
        atomicMax(&s_hdr[1], hi);
    }
}

// the tile's edges into the [64][MW] table of 16-bit multiplicities: four threads per tile row (table cleared, barriers by the caller)
// cols: the slice in LDS (index e - col_base) when the window scan kept it, else the global array (col_base = 0)
template <int CH>
__device__ __forceinline__ void core_scatter_mult(unsigned* mult, const int* s_rp, const int32_t* __restrict__ cols, int col_base, int win_lo,
                                                  int c0, int* s_hdr, int tid) {
    constexpr int MW = CH / 2 + 1;
    const int rl = tid >> 2, sub = tid & 3;
    const int beg = s_rp[rl], end = s_rp[rl + 1];
    for (int e = beg + sub; e < end; e += 16) {
        int sreg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sreg[u] = cols[(e + 4 * u < end ? e + 4 * u : end - 1) - col_base];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = sreg[u] - win_lo - c0;
            if (e + 4 * u < end && s >= 0 && s < CH) {
                const unsigned sh = 16u * (unsigned)(s & 1);
                const unsigned old = atomicAdd(&mult[rl * MW + (s >> 1)], 1u << sh);
                if (((old >> sh) & 0xffffu) == 0xffffu) s_hdr[2] = 1;  // an edge repeated 65536 times: not representable
            }
        }
    }
}

// rows [row_lo, row_lo + cn) x columns [0, width) of a row-major array (row pitch `pitch`; src = its first column) ->
// TRANSPOSED LDS slab dst[j][s] (j < 16 WT, s < CH; row stride CH + 4), zero beyond width / cn; every load before the first store
template <int WT, int CH>
__device__ __forceinline__ void core_stage_t(float* __restrict__ dst, const float* __restrict__ src, int64_t pitch, int width, int row_lo,
                                             int cn, int tid, bool vec4) {
    constexpr int VS = CH + 4;
    if (vec4) {
        constexpr int W4 = 4 * WT, PER = CH * W4 / 256;
        f32x4 reg[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + 256 * u, s = i / W4, j = 4 * (i % W4);
            reg[u] = (s < cn && j < width) ? *reinterpret_cast<const f32x4*>(src + (int64_t)(row_lo + s) * pitch + j) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + 256 * u, s = i / W4, j = 4 * (i % W4);
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[(j + c) * VS + s] = reg[u][c];
        }
    } else {
        constexpr int W = 16 * WT, PER = CH * W / 256;
        for (int b = 0; b < PER; b += 8) {
            float reg[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = tid + 256 * (b + u), s = i / W, j = i % W;
                reg[u] = (s < cn && j < width) ? src[(int64_t)(row_lo + s) * pitch + j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = tid + 256 * (b + u);
                dst[(i % W) * VS + (i / W)] = reg[u];
            }
        }
    }
}

// one 16-node tile of "other side" logits-type product: D[m = other node 16 t + ..][n = own node] = sum_j X[other][j] B[j]
// with X^T in LDS (xt[j][node]) read as four 4-byte first operands per k-group and B the own node's row in registers
template <int NG>
__device__ __forceinline__ f32x4 core_dot_tile(const float* __restrict__ xt, int VS, int t, int lrow, int lgrp, int width, const f32x4 (&B)[NG]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (16 * g < width) {
            const float* p = xt + (16 * g + 4 * lgrp) * VS + 16 * t + lrow;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p[q * VS], B[g][q], acc, 0, 0, 0);
        }
    }
    return acc;
}

// acc[g] += X^T[16 g + ..][nodes of tile t] * w  (the accumulating products: 16-byte first operands along the node axis)
template <int NG>
__device__ __forceinline__ void core_acc_tile(const float* __restrict__ xt, int VS, int t, int lrow, int lgrp, int width, const f32x4& w,
                                              f32x4 (&acc)[NG]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (16 * g < width) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xt + (16 * g + lrow) * VS + 16 * t + 4 * lgrp);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], w[q], acc[g], 0, 0, 0);
        }
    }
}

}  // namespace gnf
