// Training step of the GRevNet hot path on gfx950: gradient of
//     total_loss = -(sum_n log N(z_n; 0, I) + log_det_jacobian)            run_grevnet.py:291-295
// with respect to every MLP weight and bias, by REVERSIBLE back-propagation: the backward pass walks
// the coupling half-steps of f (gnn.py:304-341) in reverse, rebuilds each half-step's input from its
// output with the inverse update (gnn.py:359,372) and recomputes the two GNNs, so no activation of
// the forward pass is kept - what the drivers' `use_efficient_backprop` flag asks of the missing
// GNFBlock (run_grevnet.py:46,282-288).  The reference gets the same numbers from tf.gradients
// through optimizer.compute_gradients(total_loss) (run_grevnet.py:361-362).
//
// One half-step, given its output y (state) and g = dL/dy:
//   recompute  h0 = combine(x_a, agg(x_a)); s, t = MLP_s(h0), MLP_t(h0)  (all layer outputs kept in the workspace)
//   coupling   x_b = (y_b - t) exp(-s);  g_t = g_b;  g_s = g_b (y_b - t) - 1;  g_b <- g_b exp(s)
//   per layer  dW_j = h_j^T dP_j (split over node chunks, fixed-order reduce), db_j = colsum(dP_j),
//              dP_{j-1} = (dP_j W_j^T) * act'(h_j)
//   message    g_a += eps dh0 + A^T (dh0 / max(deg, 1))      (transposed CSR; concat: the two column blocks)
// All three GEMM shapes run on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32) through ONE kernel
// whose operands are staged in LDS in their natural global layout, so no transposed copy of anything
// is ever made.  Also here: the multi-tensor weight re-pack after an optimiser step, Adam
// (tf.train.AdamOptimizer, run_grevnet.py:352-356) and the two gradient clippers (run_grevnet.py:363-373).
#include "gnf_common.h"
#include "gnf_fused_bwd_dev.h"

#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <string.h>

namespace gnf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));


static constexpr int TGM = 128, TGN = 64, TGK = 32;
static constexpr int kGemmThreads = 512;

// operand storage: KC = rows indexed by the GEMM's m (or n), k contiguous;  MC = rows indexed by k
enum { OPND_KC = 0, OPND_MC = 1 };
enum { EPI_BIAS_ACT = 0, EPI_MASK = 1, EPI_SLAB = 2 };

struct GemmJob {
    const float* A;
    const float* B;
    float* C;
    const float* aux;  // EPI_BIAS_ACT: bias [N];  EPI_MASK: activation matrix [M, N] or NULL
    float* aux_out;    // EPI_SLAB: column sums of B per k-chunk [chunks, N] or NULL
};

struct GemmShape {
    int64_t lda, ldb, ldc, ldaux;
    int64_t M, K;    // rows of C, reduction length
    int32_t N;       // columns of C
    int32_t chunks;  // split of K (EPI_SLAB only; 1 otherwise)
    int64_t kchunk;  // multiple of TGK
    int32_t act;
    float alpha;
    int32_t apply_act;
};

// One R x C tile (C contiguous) of a row-major matrix -> registers (U float4 per thread), zero filled
// outside [rlim, clim).
template <int R, int C, int U, int NT = kGemmThreads>
__device__ __forceinline__ void tile_fetch(const float* __restrict__ base, int64_t ld, int64_t rlim, int64_t clim,
                                           bool vec, int tid, f32x4_t (&v)[U]) {
    constexpr int C4 = C / 4;
#pragma unroll
    for (int q = 0; q < U; ++q) {
        const int u = tid + q * NT;
        const int r = u / C4, c = (u % C4) * 4;
        f32x4_t w = {0.f, 0.f, 0.f, 0.f};
        if (r < rlim && c < clim) {
            const float* p = base + (int64_t)r * ld + c;
            if (vec && c + 3 < clim) {
                w = *reinterpret_cast<const f32x4_t*>(p);
            } else {
                w[0] = p[0];
                if (c + 1 < clim) w[1] = p[1];
                if (c + 2 < clim) w[2] = p[2];
                if (c + 3 < clim) w[3] = p[3];
            }
        }
        v[q] = w;
    }
}

template <int R, int C, int U, int NT = kGemmThreads>
__device__ __forceinline__ void tile_stash(float* __restrict__ lds, int tid, const f32x4_t (&v)[U]) {
    constexpr int C4 = C / 4;
#pragma unroll
    for (int q = 0; q < U; ++q) {
        const int u = tid + q * NT;
        const int r = u / C4, c = (u % C4) * 4;
        *reinterpret_cast<f32x4_t*>(lds + r * (C + 4) + c) = v[q];
    }
}

// C[M, N] = A . B.  128 x 64 output tile per workgroup, BK = 32, 8 waves (wave w: rows (w & 3) * 32,
// columns (w >> 2) * 32, 2 x 2 MFMA tiles); the next k-tile is fetched into registers while the matrix
// cores work on the current one.  blockIdx.z = job * chunks + chunk.
// BUF: the operand tiles are fetched with buffer loads whose descriptor ends at the operand's last valid row, so rows
// past it come back as zeros by the hardware range check and the fetch has no branch (the bounds-checked tile_fetch
// below compiles to ~450 instructions with vmcnt(0) waits inside its scalar tails).  Columns past the valid width are
// NOT zeroed by the range check: for an MC operand they are output columns that are never stored; for a KC operand
// they are the reduction tail, and the lane's offset is pushed out of range instead (needs K % 4 == 0).
template <int AK, int BK, int EPI, bool BUF = false, int TN = TGN>
__device__ __forceinline__ void gemm_tile(const GemmJob& job, const GemmShape& sh, const int bx, const int by,
                                          const int chunk) {
    constexpr int AR = AK == OPND_KC ? TGM : TGK, AC = AK == OPND_KC ? TGK : TGM;  // LDS tile rows x cols
    constexpr int BR = BK == OPND_KC ? TN : TGK, BC = BK == OPND_KC ? TGK : TN;
    constexpr int NW = TN / 32;  // 16-column MFMA tiles per wave: 2 (64-column tile) or 4 (128)
    constexpr int UB = TN / 64;  // float4 per thread of a B tile
    // two LDS stages: the next k-tile is written into the other stage behind this one's MFMAs - ONE barrier per step and
    // nothing waits for a stage to drain (with a single stage every step stalled twice; at a few hundred rows there is
    // one workgroup per CU and nothing else to fill those stalls)
    constexpr int kAs = AR * (AC + 4), kBs = BR * (BC + 4);
    __shared__ __attribute__((aligned(16))) float As2[2 * kAs];
    __shared__ __attribute__((aligned(16))) float Bs2[2 * kBs];
    const int64_t m0 = (int64_t)by * TGM;
    const int n0 = bx * TN;
    const int64_t kbeg = (int64_t)chunk * sh.kchunk;
    const int64_t kend = (EPI == EPI_SLAB) ? (kbeg + sh.kchunk < sh.K ? kbeg + sh.kchunk : sh.K) : sh.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int wm = (wave & 3) * 32, wn = (wave >> 2) * (TN / 2);
    const bool avec = (sh.lda % 4 == 0) && (reinterpret_cast<uintptr_t>(job.A) % 16 == 0);
    const bool bvec = (sh.ldb % 4 == 0) && (reinterpret_cast<uintptr_t>(job.B) % 16 == 0);

    f32x4_t acc[2][NW];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            float bv = 0.f;
            if (EPI == EPI_BIAS_ACT) {
                const int gc = n0 + wn + 16 * b + lrow;
                bv = (job.aux && gc < sh.N) ? job.aux[gc] : 0.f;  // (aux == NULL: no bias - the attention projections)
            }
            acc[m][b] = f32x4_t{bv, bv, bv, bv};
        }
    float colsum = 0.f;  // EPI_SLAB: sum over this chunk's k of B[k][n0 + tid]

    f32x4_t av[2], bvr[UB];
    // buffer path: descriptors over [tile's first row .. operand's last valid row], per-thread byte offsets of its float4s
    constexpr int AC4 = AC / 4, BC4 = BC / 4;
    const int64_t a_rows = AK == OPND_KC ? sh.M - m0 : kend - kbeg, b_rows = BK == OPND_KC ? (int64_t)sh.N - n0 : kend - kbeg;
    const float* a_base = AK == OPND_KC ? job.A + m0 * sh.lda : job.A + kbeg * sh.lda + m0;
    const float* b_base = BK == OPND_KC ? job.B + (int64_t)n0 * sh.ldb : job.B + kbeg * sh.ldb + n0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a_base), 0, BUF && a_rows > 0 ? (int)((a_rows * sh.lda - (AK == OPND_KC ? 0 : m0)) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(b_base), 0, BUF && b_rows > 0 ? (int)((b_rows * sh.ldb - (BK == OPND_KC ? 0 : n0)) * 4) : 0, 0x00020000);
    const int ar = tid / AC4, ac = (tid % AC4) * 4, br = tid / BC4, bc = (tid % BC4) * 4;
    const int va0 = (ar * (int)sh.lda + ac) * 4, va1 = ((ar + kGemmThreads / AC4) * (int)sh.lda + ac) * 4;
    const int vb0 = (br * (int)sh.ldb + bc) * 4, vb1 = ((br + kGemmThreads / BC4) * (int)sh.ldb + bc) * 4;
    constexpr int kOut = 0x7fffffff;  // beyond any descriptor: the load returns zeros
    auto fetch = [&](int64_t k0) {
        if (BUF) {
            if (AK == OPND_KC) {  // columns are k: offset by k0, lanes past the reduction length read nothing
                const int ko = (int)k0 * 4;
                const bool in = k0 + ac < kend;
                av[0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ra, in ? va0 + ko : kOut, 0, 0));
                av[1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ra, in ? va1 + ko : kOut, 0, 0));
            } else {              // rows are k: offset by whole rows, rows past the chunk are out of range by themselves
                const int ko = (int)(k0 - kbeg) * (int)sh.lda * 4;
                av[0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ra, va0 + ko, 0, 0));
                av[1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ra, va1 + ko, 0, 0));
            }
            if (BK == OPND_KC) {
                const bool in = k0 + bc < kend;
                bvr[0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rb, in ? vb0 + (int)k0 * 4 : kOut, 0, 0));
                if (UB > 1) bvr[UB - 1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rb, in ? vb1 + (int)k0 * 4 : kOut, 0, 0));
            } else {
                bvr[0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rb, vb0 + (int)(k0 - kbeg) * (int)sh.ldb * 4, 0, 0));
                if (UB > 1) bvr[UB - 1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rb, vb1 + (int)(k0 - kbeg) * (int)sh.ldb * 4, 0, 0));
            }
            return;
        }
        if (AK == OPND_KC)
            tile_fetch<AR, AC, 2>(job.A + m0 * sh.lda + k0, sh.lda, sh.M - m0, kend - k0, avec, tid, av);
        else
            tile_fetch<AR, AC, 2>(job.A + k0 * sh.lda + m0, sh.lda, kend - k0, sh.M - m0, avec, tid, av);
        if (BK == OPND_KC)
            tile_fetch<BR, BC, UB>(job.B + (int64_t)n0 * sh.ldb + k0, sh.ldb, sh.N - n0, kend - k0, bvec, tid, bvr);
        else
            tile_fetch<BR, BC, UB>(job.B + k0 * sh.ldb + n0, sh.ldb, kend - k0, sh.N - n0, bvec, tid, bvr);
    };
    int cur = 0;
    if (kbeg < kend) {
        fetch(kbeg);
        tile_stash<AR, AC, 2>(As2, tid, av);
        tile_stash<BR, BC, UB>(Bs2, tid, bvr);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += TGK) {
        const bool more = k0 + TGK < kend;
        if (more) fetch(k0 + TGK);  // in flight behind this step's MFMAs
        const float* As = As2 + cur * kAs;
        const float* Bs = Bs2 + cur * kBs;
        if (EPI == EPI_SLAB && BK == OPND_MC) {
            if (by == 0 && tid < TN) {
#pragma unroll 8
                for (int k = 0; k < TGK; ++k) colsum += Bs[k * (BC + 4) + tid];
            }
        }
#pragma unroll
        for (int kg = 0; kg < TGK / 16; ++kg) {
            float a[2][4], b[NW][4];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (AK == OPND_KC) {
                    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(As + (wm + 16 * m + lrow) * (AC + 4) + 16 * kg + 4 * lgrp);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[m][q] = t[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[m][q] = As[(16 * kg + 4 * lgrp + q) * (AC + 4) + wm + 16 * m + lrow];
                }
            }
#pragma unroll
            for (int n = 0; n < NW; ++n) {
                if (BK == OPND_KC) {
                    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(Bs + (wn + 16 * n + lrow) * (BC + 4) + 16 * kg + 4 * lgrp);
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[n][q] = t[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[n][q] = Bs[(16 * kg + 4 * lgrp + q) * (BC + 4) + wn + 16 * n + lrow];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int n = 0; n < NW; ++n)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][q], b[n][q], acc[m][n], 0, 0, 0);
        }
        if (more) {
            tile_stash<AR, AC, 2>(As2 + (cur ^ 1) * kAs, tid, av);
            tile_stash<BR, BC, UB>(Bs2 + (cur ^ 1) * kBs, tid, bvr);
        }
        __syncthreads();
        cur ^= 1;
    }
    // accumulator layout: col = lane & 15, row = 4 * (lane >> 4) + r
    float* __restrict__ Cp = job.C;
    if (EPI == EPI_SLAB) Cp += (int64_t)chunk * sh.M * sh.N;
    // EPI_MASK: the stored activations that pick act' - ALL of a lane's 8 NW values requested before the first is used (read
    // inside the store loop they were one dependent memory round trip per element: the 100 -> 2048 dX product of the data
    // driver's last layer took 223 us for 2.2 GFLOP)
    [[maybe_unused]] float hmask[2][NW][4];
    if (EPI == EPI_MASK && job.aux) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int b = 0; b < NW; ++b) {
                const int gc = n0 + wn + 16 * b + lrow;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t gr = m0 + wm + 16 * m + 4 * lgrp + r;
                    hmask[m][b][r] = (gr < sh.M && gc < sh.N) ? job.aux[gr * sh.ldaux + gc] : 0.f;
                }
            }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            const int gc = n0 + wn + 16 * b + lrow;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t gr = m0 + wm + 16 * m + 4 * lgrp + r;
                if (gr < sh.M && gc < sh.N) {
                    float v = acc[m][b][r];
                    if (EPI == EPI_BIAS_ACT) {
                        if (sh.apply_act) v = (sh.act == GNF_ACT_RELU) ? fmaxf(v, 0.f) : fmaxf(v, sh.alpha * v);
                    } else if (EPI == EPI_MASK) {
                        if (job.aux) {  // act'(pre) read off the stored activation: h > 0 <=> pre > 0
                            const float slope = (sh.act == GNF_ACT_RELU) ? 0.f : sh.alpha;
                            v = hmask[m][b][r] > 0.f ? v : v * slope;
                        }
                    }
                    Cp[gr * sh.ldc + gc] = v;
                }
            }
        }
    if (EPI == EPI_SLAB && job.aux_out && by == 0 && tid < TN && n0 + tid < sh.N)
        job.aux_out[(int64_t)chunk * sh.N + n0 + tid] = colsum;
}

// (An LDS-direct edition of this tile - `buffer_load ... lds` into fragment-order blocks, three stages - was built and
// measured in round 1: the wait for a tile's loads dropped from ~1050 to ~100 cycles per k-step, but ISSUING them took
// 1600 cycles per step; wide_fc 10.4 -> 10.8 ms, the 2048-wide trainer 12.0 -> 13.0 ms per iteration.  Removed in round 3
// with its developer option; DESIGN.md's changelog has the numbers.)

template <int AK, int BK, int EPI, bool BUF, int TN = TGN>
__global__ __launch_bounds__(kGemmThreads) void k_gemm(GemmJob j0, GemmJob j1, GemmShape sh) {
    const int jz = blockIdx.z / sh.chunks, chunk = blockIdx.z - jz * sh.chunks;
    const GemmJob job = jz ? j1 : j0;
    gemm_tile<AK, BK, EPI, BUF, TN>(job, sh, blockIdx.x, blockIdx.y, chunk);
}

// may the operands of a GEMM go through the buffer path? (32-bit byte offsets; a KC operand's reduction length in whole
// float4s)
static bool gemm_buf_ok(int ak, int bk, int64_t a_rows, int64_t lda, int64_t b_rows, int64_t ldb, int64_t K) {
    if ((a_rows + 256) * lda * 4 >= ((int64_t)1 << 31) || (b_rows + 256) * ldb * 4 >= ((int64_t)1 << 31)) return false;
    if ((ak == OPND_KC || bk == OPND_KC) && K % 4 != 0) return false;
    return true;
}

// Grouped split-K launch: up to kMaxGroup GEMMs of different M x N (the dW of every layer of both nets of
// a half-step) over the same K = node axis, in one grid.  blockIdx.z = job * chunks + chunk.
static constexpr int kMaxGroup = 2 * (GNF_MAX_LAYERS + 4);  // per net: K layers (+ Wq, Wk, Wv, Wo of an attention block)
struct GroupedGemm {
    GemmJob job[kMaxGroup];
    int64_t lda[kMaxGroup], ldb[kMaxGroup];
    int32_t M[kMaxGroup], N[kMaxGroup];
    int64_t K, kchunk;
    int32_t chunks;
};
// 1-D grid, XCD-aware: workgroup b runs on XCD b % 8, and the tiles of one (job, chunk) - which share the chunk's
// two operand panels - are all given to the same XCD (z % 8), so each panel is pulled into ONE L2 instead of eight
// (PMC before: 14 % L2 hit rate, 109 MiB fetched per dispatch for 46 MiB of operands).
// With few (job, chunk) pairs and many tiles each (wide layers: 6 pairs of 512 tiles), a pair's tile grid is cut into
// `groups` bands of `rpg` tile rows (tiles of a row share the A panel) and the bands are what is dealt to the XCDs -
// otherwise 6 pairs would occupy 6 of the 8 XCDs.
template <bool BUF>
__global__ __launch_bounds__(kGemmThreads) void k_gemm_dw_grouped(const GroupedGemm g, int gx, int gy, int nz, int groups,
                                                                  int rpg) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_v = gx * rpg;
    const int vi = slot / per_v, t = slot - vi * per_v;
    const int v = xcd + 8 * vi;
    const int z = v / groups, band = v - z * groups;
    if (z >= nz) return;
    const int bx = t % gx, by = band * rpg + t / gx;
    if (by >= gy) return;
    const int jz = z / g.chunks, chunk = z - jz * g.chunks;
    const int M = g.M[jz], N = g.N[jz];
    if (bx * TGN >= N || by * TGM >= M) return;  // the tile grid covers the largest job
    GemmShape sh;
    sh.lda = g.lda[jz], sh.ldb = g.ldb[jz], sh.ldc = N, sh.ldaux = 0;
    sh.M = M, sh.K = g.K, sh.N = N, sh.chunks = g.chunks, sh.kchunk = g.kchunk;
    sh.act = 0, sh.alpha = 0.f, sh.apply_act = 0;
    const GemmJob job = g.job[jz];
    gemm_tile<OPND_MC, OPND_MC, EPI_SLAB, BUF>(job, sh, bx, by, chunk);
}

// ---- wide split-K weight-gradient GEMM -----------------------------------------------------------------------
// Same contract as k_gemm_dw_grouped (slabs + column sums per node chunk), built to run at FULL matrix-core rate with
// ONE workgroup per CU: 128 x 128 output tile, 4 waves each owning 64 x 64 (4 x 4 MFMA tiles, 64 accumulator
// registers), BK = 32 = 128 MFMAs per wave per step (~1.7 us) behind which the next step's eight float4 loads per
// thread are in flight; two LDS stages, one barrier per step.  With few, long workgroups the launch can be sized to
// the CUs the fused backward kernel leaves idle (see launch_weight_grads).
static constexpr int WGM = 128, WGN = 128, WGK = 32;
static constexpr int kWideThreads = 512;  // 8 waves: two per SIMD, each fills the other's issue gaps
static constexpr int kWideQ = WGK * 32 / kWideThreads;  // float4 per thread and operand tile
static constexpr int kWidePieces = 4 * kWideQ;
static constexpr int kWideLd = WGM + 4;
static constexpr int kWideStage = 2 * WGK * kWideLd;  // floats: A tile then B tile
static constexpr size_t kWideLdsMin = (size_t)2 * kWideStage * sizeof(float);
static constexpr int kWideLoadPolicy = 0;  // cache policy of the operand loads (streaming / non-temporal hints measured no different from the default)

template <int G>
struct WideGemmT {
    GemmJob job[G];  // largest first
    int64_t lda[G], ldb[G];
    int32_t M[G], N[G];
    int32_t unit_base[G + 1];  // prefix sums of workgroups (128 x 128 tiles x node chunks) per job
    int32_t gx[G];
    int32_t chunks[G];         // split of the node axis, per job: light jobs are cut less often
    int32_t kchunk[G];         // rows per chunk, a multiple of WGK
    int64_t K;
    int32_t njobs;
    // stream-K over the costliest jobs (sk_q > 0): their tiles x k-steps form ONE axis of sk_tiles * sk_steps steps, cut
    // into equal runs of sk_q steps, one run per workgroup (a run may end one tile and begin the next: at most two
    // items per workgroup since sk_q <= sk_steps).  The pieces of a tile are its slabs, numbered in workgroup order;
    // chunks[j] = the largest piece count among job j's tiles, and the workgroup holding a tile's piece 0 zero-fills
    // the slabs that tile does not reach, so the reduce can add chunks[j] slabs everywhere.  Jobs 0 .. sk_jobs-1.
    int32_t sk_q, sk_steps, sk_tiles, sk_jobs;
    int32_t sk_light_base;  // first unit of the jobs cut the uniform way (they ride behind, strided)
    // > 0: the first xcd_jobs jobs are whole-tile jobs (one chunk per tile) of a launch with a multiple of 8 workgroups, and
    // workgroup b runs on XCD b % 8: a job's tile grid is then dealt to the XCDs as 2 x 4 BLOCKS (gx / 2 columns x gy / 4 rows
    // of tiles each) instead of every eighth tile, so that an XCD's L2 holds gy / 4 A panels and gx / 2 B panels instead of
    // all gy A panels (PMC, wide_fc_train: 516 MB fetched per launch for ~100 MB of operands, L2 hit rate 0.67)
    int32_t xcd_jobs;
    // merged training launch, stream-K plans only: the uniformly cut (thin) jobs are NOT taken by the weight-gradient
    // workgroups behind their runs but by the launch's backward-tile workgroups once their tile and their share of the slab
    // reduce are done (they end ~11 - 17 us before the launch does: per-workgroup stamps, CHANGELOG round 6) - cut for THEIR
    // count, one short unit each
    int32_t light_on_tiles;
};
using WideGemm = WideGemmT<kMaxGroup>;
// what fits next to the backward kernel's arguments in one launch (4 KB of kernel arguments)
static constexpr int kMergedGroup = 20;  // message-passing nets of up to 8 layers (16 jobs), attention nets of up to 6 (2 x (6 + 4))
using WideGemmS = WideGemmT<kMergedGroup>;

// One BK = 32 step of a wave's MTW x NTW block of MFMA tiles.  A wave is alone on its SIMD here and issues in order,
// so everything that is not an MFMA has to sit BETWEEN MFMAs to be free (an instruction placed behind a run of MFMAs
// waits for all of them to issue): after every MFMA one other instruction is slotted in - first the LDS fragment
// loads of k-slice ks + 2 (three register sets), then the `piece`s the caller hands in (16 of them: the LDS writes of
// the next step's operand tiles and the global loads of the step after).  Order pinned with sched_barrier.
// cs[n] += this lane's B fragments: the column sums of B (the bias gradient) fall out of the registers.
// acc += a (x) b with the accumulator pinned to ONE AGPR tuple: through the builtin the register allocator gave the
// loop-carried accumulators a different tuple at the end of the step than at its start and paid ~100 moves per step.
// (No software hazard here: an accumulator is touched again 16 MFMAs later, and read out only after the last barrier.)
__device__ __forceinline__ void mfma_inplace(f32x4_t& c, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

struct WideLayout {
    bool along_n, along_m;  // 1 x 8 waves of 32 x 16 along N / 8 x 1 of 16 x 32 along M / (neither) 4 x 2 waves of 32 x 64
};
__host__ __device__ inline WideLayout wide_layout(int m_left, int n_left) {  // extents of the tile inside the matrix
    WideLayout l;
    l.along_n = m_left <= 32;
    l.along_m = !l.along_n && n_left <= 32;
    return l;
}

template <int MTW, int NTW, typename Piece>
__device__ __forceinline__ void wide_compute(const float* __restrict__ As, const float* __restrict__ Bs,
                                             f32x4_t (&acc)[4][4], float (&cs)[4], int wm, int wn, int lrow, int lgrp,
                                             Piece&& piece) {
    constexpr int NS = WGK / 4, NL = MTW + NTW;
    float a[3][MTW], b[3][NTW];
    const float* ap = As + 4 * lgrp * kWideLd + wm + lrow;
    const float* bp = Bs + 4 * lgrp * kWideLd + wn + lrow;
    auto load1 = [&](int ks, int i) {  // slice ks: k rows 16 * (ks / 4) + 4 * lgrp + (ks % 4)
        const int row = 16 * (ks >> 2) + (ks & 3);
        if (i < MTW)
            a[ks % 3][i] = ap[row * kWideLd + 16 * i];
        else
            b[ks % 3][i - MTW] = bp[row * kWideLd + 16 * (i - MTW)];
    };
#pragma unroll
    for (int i = 0; i < NL; ++i) load1(0, i);
#pragma unroll
    for (int i = 0; i < NL; ++i) load1(1, i);
    __builtin_amdgcn_sched_barrier(0);
    int pc = 0;  // pieces handed out so far (a constant everywhere once the loops are unrolled)
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
        int li = 0;
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                mfma_inplace(acc[m][n], a[ks % 3][m], b[ks % 3][n]);
                if (ks + 2 < NS && li < NL)
                    load1(ks + 2, li++);
                else if (ks >= 1 && pc < kWidePieces)
                    piece(pc++);
                __builtin_amdgcn_sched_barrier(0);
            }
        if (ks + 2 < NS)
            for (; li < NL; ++li) load1(ks + 2, li);  // narrow blocks: more loads than MFMA slots
#pragma unroll
        for (int n = 0; n < NTW; ++n) cs[n] += b[ks % 3][n];
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; pc < kWidePieces; ++pc) piece(pc);
}


// workgroup barrier that only waits for this wave's LDS traffic (not for the global loads in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// BUF: every operand is 16-byte aligned with a row pitch that is a multiple of 4 floats -> the tiles are read with
// buffer loads through a descriptor that ends at the chunk's last row: rows past the chunk come back as zeros by the
// hardware range check (on the VGPR offset), so the fetch is branch-free and can be cut into the pieces wide_compute
// interleaves; tiles past the chunk's end are fetched and stashed like any other (zeros, never read).  Columns past
// M / N read whatever follows in the row; they only reach accumulators that are never stored.  The generic path
// (any pitch / alignment) keeps the bounds-checked fetch in front of the MFMA block.
// vblock / vgrid: this workgroup's index among the vgrid workgroups that share the launch's weight-gradient work
// items [it0, it1) of the workgroup's list: all of them (0, INT_MAX), only its stream-K run (0, 2), or only its share of the
// uniformly cut jobs (2, INT_MAX) - see WideGemmT.light_on_tiles
template <bool BUF, class WG>
__device__ __forceinline__ void dw_wide_body(const WG& g, const int vblock, const int vgrid, const int it0 = 0,
                                             const int it1 = 0x7fffffff) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    // a workgroup takes unit vblock (a chunk of one of the costliest tiles; those are cut equal) and then, strided,
    // its share of the cheap units that follow them in the list
    for (int it = it0; it < it1; ++it) {
    int j = 0, lt, chunk, zero_from = 0, zero_to = 0;
    int64_t kbeg, kend;
    if (g.sk_q > 0 && it < 2) {  // stream-K: this workgroup's run [a, b) of the costliest jobs' step axis
        const int64_t total = (int64_t)g.sk_tiles * g.sk_steps;
        const int64_t a = (int64_t)vblock * g.sk_q, b = a + g.sk_q < total ? a + g.sk_q : total;
        if (a >= b) continue;
        const int t0 = (int)(a / g.sk_steps);
        const int64_t e0 = b < (int64_t)(t0 + 1) * g.sk_steps ? b : (int64_t)(t0 + 1) * g.sk_steps;
        int t;
        int64_t s0, s1;
        if (it == 0) {
            t = t0, s0 = a - (int64_t)t0 * g.sk_steps, s1 = e0 - (int64_t)t0 * g.sk_steps;
        } else {
            if (b <= e0) continue;
            t = t0 + 1, s0 = 0, s1 = b - e0;
        }
        while (j + 1 < g.sk_jobs && t >= g.unit_base[j + 1]) ++j;  // unit_base of these jobs counts TILES
        lt = t - g.unit_base[j];
        const int first_w = (int)(((int64_t)t * g.sk_steps) / g.sk_q);
        const int last_w = (int)((((int64_t)(t + 1) * g.sk_steps) - 1) / g.sk_q);
        chunk = vblock - first_w;
        kbeg = s0 * WGK;
        kend = s1 * WGK < g.K ? s1 * WGK : g.K;
        if (chunk == 0) zero_from = last_w - first_w + 1, zero_to = g.chunks[j];
    } else {
        const int first = g.sk_q > 0 ? g.sk_light_base : 0;
        const int u = first + vblock + (it - (g.sk_q > 0 ? 2 : 0)) * vgrid;
        if (u >= g.unit_base[g.njobs]) break;
        j = g.sk_q > 0 ? g.sk_jobs : 0;
        while (j + 1 < g.njobs && u >= g.unit_base[j + 1]) ++j;
        const int lu = u - g.unit_base[j];
        const int nchunk = g.chunks[j];
        lt = lu / nchunk, chunk = lu - lt * nchunk;
        kbeg = (int64_t)chunk * g.kchunk[j];
        kend = kbeg + g.kchunk[j] < g.K ? kbeg + g.kchunk[j] : g.K;
    }
    const int gxj = g.gx[j];
    int bx = lt % gxj, by = lt / gxj;
    const int M = g.M[j], N = g.N[j];
    if (j < g.xcd_jobs) {
        const int gyj = (M + WGM - 1) / WGM;
        if ((gxj & 1) == 0 && (gyj & 3) == 0) {  // tile lt of the job runs on XCD lt % 8 (unit_base and the grid are multiples of 8)
            const int xcd = lt & 7, slot = lt >> 3, bw = gxj >> 1, bh = gyj >> 2;   // block width / height in tiles
            bx = (xcd & 1) * bw + slot % bw;
            by = (xcd >> 1) * bh + slot / bw;
        }
    }
    const int64_t lda = g.lda[j], ldb = g.ldb[j];
    const GemmJob job = g.job[j];
    const int m0 = by * WGM, n0 = bx * WGN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, lgrp = lane >> 4;
    // the eight waves tile the 128 x 128 block 4 x 2 (32 x 64 each) - or, when the tile is a thin strip (a layer with
    // few inputs or outputs), 1 x 8 / 8 x 1 blocks along the long side, so that every SIMD has work
    const WideLayout lay = wide_layout(M - m0, N - n0);
    const int wext_m = lay.along_m ? 16 : 32, wext_n = lay.along_n ? 16 : (lay.along_m ? 32 : 64);
    const int wm = lay.along_n ? 0 : (lay.along_m ? wave * 16 : (wave & 3) * 32);
    const int wn = lay.along_n ? wave * 16 : (lay.along_m ? 0 : (wave >> 2) * 64);
    int mt = (M - m0 - wm + 15) / 16, nt = (N - n0 - wn + 15) / 16;  // MFMA tiles of this wave that hold anything
    mt = mt < 0 ? 0 : (mt > wext_m / 16 ? wext_m / 16 : mt);
    nt = nt < 0 ? 0 : (nt > wext_n / 16 ? wext_n / 16 : nt);
    const int shape = (mt == 0 || nt == 0) ? 0 : (((mt > 2 ? 4 : mt) << 4) | (nt > 2 ? 4 : nt));
    const bool avec = (lda % 4 == 0) && (reinterpret_cast<uintptr_t>(job.A) % 16 == 0);
    const bool bvec = (ldb % 4 == 0) && (reinterpret_cast<uintptr_t>(job.B) % 16 == 0);
    const bool want_cs = job.aux_out != nullptr && by == 0 && wm == 0;

    f32x4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};  // this lane's share of the column sums of B, columns wn + 16 n + lrow

    // operand tiles in flight: 4 + 4 float4 per thread.  Thread t moves rows (t >> 5) + 8 q, columns 4 (t & 31) ..
    f32x4_t av[kWideQ], bv[kWideQ];
    const int64_t krows = kend > kbeg ? kend - kbeg : 0;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(job.A + kbeg * lda + m0), 0, BUF ? (int)((krows * lda - m0) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(job.B + kbeg * ldb + n0), 0, BUF ? (int)((krows * ldb - n0) * 4) : 0, 0x00020000);
    constexpr int kRowsPass = kWideThreads / 32;  // rows one pass of the workgroup covers
    const int va = ((tid >> 5) * (int)lda + (tid & 31) * 4) * 4, vb = ((tid >> 5) * (int)ldb + (tid & 31) * 4) * 4;
    const int qa = kRowsPass * (int)lda * 4, qb = kRowsPass * (int)ldb * 4;         // bytes between a thread's q-th and (q+1)-th row
    const int lw = (tid >> 5) * kWideLd + (tid & 31) * 4;           // LDS float offset of the thread's q = 0 float4
    auto fetch_buf = [&](int voa, int vob, int q, bool b_side) {
        if (!b_side)
            av[q] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ra, voa + q * qa, 0, kWideLoadPolicy));
        else
            bv[q] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rb, vob + q * qb, 0, kWideLoadPolicy));
    };
    auto stash1 = [&](float* stage, int q, bool b_side) {
        float* d = stage + (b_side ? WGK * kWideLd : 0) + lw + q * kRowsPass * kWideLd;
        *reinterpret_cast<f32x4_t*>(d) = b_side ? bv[q] : av[q];
    };
    auto fetch_any = [&](int64_t k0) {
        if (BUF) {
            const int voa = va + (int)(k0 - kbeg) * (int)lda * 4, vob = vb + (int)(k0 - kbeg) * (int)ldb * 4;
#pragma unroll
            for (int q = 0; q < kWideQ; ++q) fetch_buf(voa, vob, q, false), fetch_buf(voa, vob, q, true);
        } else {
            tile_fetch<WGK, WGM, kWideQ, kWideThreads>(job.A + k0 * lda + m0, lda, kend - k0, M - m0, avec, tid, av);
            tile_fetch<WGK, WGN, kWideQ, kWideThreads>(job.B + k0 * ldb + n0, ldb, kend - k0, N - n0, bvec, tid, bv);
        }
    };
    auto stash_all = [&](float* stage) {
#pragma unroll
        for (int q = 0; q < kWideQ; ++q) stash1(stage, q, false), stash1(stage, q, true);
    };

    int cur = 0;
    if (BUF || kbeg < kend) {
        fetch_any(kbeg);
        stash_all(wl);
        if (BUF || kbeg + WGK < kend) fetch_any(kbeg + WGK);  // stays in flight into step 0
    }
    lds_barrier();
    // The K loop exists once per block shape (dispatch OUTSIDE the loop: with the switch inside, the accumulators
    // crossed it in VGPRs and were copied to and from the AGPRs the MFMAs use on every step - 128 moves per step).
    auto k_loop = [&](auto mtw_c, auto ntw_c) {
        constexpr int MTW = decltype(mtw_c)::value, NTW = decltype(ntw_c)::value;
        for (int64_t k0 = kbeg; k0 < kend; k0 += WGK) {
            const float* As = wl + cur * kWideStage;
            const float* Bs = As + WGK * kWideLd;
            float* nstage = wl + (cur ^ 1) * kWideStage;
            // the registers hold the tiles of step k0 + WGK: they go to the other stage, then receive step k0 + 2 WGK
            if (!BUF) {
                if (k0 + WGK < kend) stash_all(nstage);
                if (k0 + 2 * WGK < kend) fetch_any(k0 + 2 * WGK);
            }
            const int voa = va + (int)(k0 + 2 * WGK - kbeg) * (int)lda * 4, vob = vb + (int)(k0 + 2 * WGK - kbeg) * (int)ldb * 4;
            auto piece = [&](int i) {
                if (!BUF) return;
                if (i < 2 * kWideQ)
                    stash1(nstage, i % kWideQ, i >= kWideQ);
                else
                    fetch_buf(voa, vob, i % kWideQ, i >= 3 * kWideQ);
            };
            if constexpr (MTW > 0) {
                wide_compute<MTW, NTW>(As, Bs, acc, cs, wm, wn, lrow, lgrp, piece);
            } else {  // nothing of this wave's block lies inside the matrix; its share of the tile traffic remains
#pragma unroll
                for (int i = 0; i < kWidePieces; ++i) piece(i);
            }
            lds_barrier();
            cur ^= 1;
        }
    };
    // partial blocks round up to 1 / 2 / 4 tiles (the stages are zero filled beyond M, N)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    switch (shape) {
        case 0x44: k_loop(I4{}, I4{}); break;
        case 0x42: k_loop(I4{}, I2{}); break;
        case 0x41: k_loop(I4{}, I1{}); break;
        case 0x24: k_loop(I2{}, I4{}); break;
        case 0x22: k_loop(I2{}, I2{}); break;
        case 0x21: k_loop(I2{}, I1{}); break;
        case 0x14: k_loop(I1{}, I4{}); break;
        case 0x12: k_loop(I1{}, I2{}); break;
        case 0x11: k_loop(I1{}, I1{}); break;
        default: k_loop(I0{}, I0{}); break;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs have left the pipe before acc is read
    // accumulator layout: col = lane & 15, row = 4 * (lane >> 4) + r
    float* __restrict__ Cp = job.C + (int64_t)chunk * M * N;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int gc = n0 + wn + 16 * n + lrow;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = m0 + wm + 16 * m + 4 * lgrp + r;
                if (16 * m < wext_m && 16 * n < wext_n && gr < M && gc < N) Cp[(int64_t)gr * N + gc] = acc[m][n][r];
            }
        }
    if (want_cs) {  // the four lane groups hold the four k-residues of every column
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v = cs[n];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int gc = n0 + wn + 16 * n + lrow;
            if (lgrp == 0 && 16 * n < wext_n && gc < N) job.aux_out[(int64_t)chunk * N + gc] = v;
        }
    }
    for (int zs = zero_from; zs < zero_to; ++zs) {  // stream-K: the slabs this tile has no piece for
        float* __restrict__ Zp = job.C + (int64_t)zs * M * N;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int gc = n0 + wn + 16 * n + lrow;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = m0 + wm + 16 * m + 4 * lgrp + r;
                    if (16 * m < wext_m && 16 * n < wext_n && gr < M && gc < N) Zp[(int64_t)gr * N + gc] = 0.f;
                }
            }
        if (want_cs) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int gc = n0 + wn + 16 * n + lrow;
                if (lgrp == 0 && 16 * n < wext_n && gc < N) job.aux_out[(int64_t)zs * N + gc] = 0.f;
            }
        }
    }
    }  // items of this workgroup
}

template <bool BUF>
// (waves_per_eu(2, 3), not (2, 2): with a maximum of two the compiler PADS the register allocation to 169 so that a third wave
// cannot fit - and then nothing of the main stream's kernels fits beside this one either; its own 140 registers leave them 224)
__global__ __launch_bounds__(kWideThreads) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_gemm_dw_wide(const WideGemm g) {
    dw_wide_body<BUF>(g, (int)blockIdx.x, (int)gridDim.x);
}

// Thin outputs with a long reduction (a 2048 -> 100 layer on a 440-node batch: 16 tiles, 64 k-steps each, 80 us on 16
// CUs): the reduction is split over `chunks` workgroups per tile into slabs, and this kernel sums the slabs in chunk
// order and applies the epilogue the unsplit kernel would have applied.
template <int EPI>
__global__ __launch_bounds__(256) void k_splitk_epilogue(GemmJob j0, GemmJob j1, const float* s0, const float* s1,
                                                         int chunks, GemmShape sh) {
    const GemmJob job = blockIdx.y ? j1 : j0;
    const float* __restrict__ slab = blockIdx.y ? s1 : s0;
    const int64_t mn = sh.M * sh.N;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= mn) return;
    const int64_t m = e / sh.N;
    const int n = (int)(e - m * sh.N);
    float v = 0.f;
    for (int c = 0; c < chunks; ++c) v += slab[(int64_t)c * mn + e];
    if (EPI == EPI_BIAS_ACT) {
        if (job.aux) v += job.aux[n];
        if (sh.apply_act) v = (sh.act == GNF_ACT_RELU) ? fmaxf(v, 0.f) : fmaxf(v, sh.alpha * v);
    } else if (EPI == EPI_MASK) {
        if (job.aux) {
            const float h = job.aux[m * sh.ldaux + n];
            const float slope = (sh.act == GNF_ACT_RELU) ? 0.f : sh.alpha;
            v = h > 0.f ? v : v * slope;
        }
    }
    job.C[m * sh.ldc + n] = v;
}

// sk[q] (nullable): scratch of sk_floats floats for job q's slabs when the launch is thin enough to be split
template <int AK, int BK, int EPI>
static int launch_gemm(const GemmJob* jobs, int nj, const GemmShape& sh, hipStream_t st, float* const* sk = nullptr,
                       size_t sk_floats = 0) {
    if (sh.M == 0 || sh.N == 0) return GNF_OK;
    {
        const int64_t tiles = (int64_t)((sh.N + TGN - 1) / TGN) * ((sh.M + TGM - 1) / TGM) * nj;
        if (sk && sk[0] && sk[nj - 1] && EPI != EPI_SLAB && sh.chunks == 1 && tiles < 192 && sh.K >= 512) {
            int64_t chunks = 2 * (int64_t)big_cu_count() / tiles;  // two 54 KB workgroups fit a CU
            if (chunks > 16) chunks = 16;
            if (chunks > sh.K / 128) chunks = sh.K / 128;
            while (chunks > 1 && (size_t)chunks * sh.M * sh.N > sk_floats) --chunks;
            if (chunks > 1) {
                GemmShape s2 = sh;
                s2.kchunk = ((sh.K + chunks - 1) / chunks + TGK - 1) / TGK * TGK;
                s2.chunks = (int32_t)((sh.K + s2.kchunk - 1) / s2.kchunk);
                s2.ldc = sh.N;
                GemmJob sj[2];
                for (int q = 0; q < nj; ++q) sj[q] = GemmJob{jobs[q].A, jobs[q].B, sk[q], nullptr, nullptr};
                dim3 grid((unsigned)((sh.N + TGN - 1) / TGN), (unsigned)((sh.M + TGM - 1) / TGM), (unsigned)(nj * s2.chunks));
                const bool buf = gemm_buf_ok(AK, BK, AK == OPND_KC ? sh.M : sh.K, sh.lda, BK == OPND_KC ? (int64_t)sh.N : sh.K, sh.ldb, sh.K) &&
                                 (AK != OPND_KC && BK != OPND_KC ? true : s2.kchunk % 4 == 0);
                if (buf)
                    hipLaunchKernelGGL((k_gemm<AK, BK, EPI_SLAB, true>), grid, dim3(kGemmThreads), 0, st, sj[0], sj[nj - 1], s2);
                else
                    hipLaunchKernelGGL((k_gemm<AK, BK, EPI_SLAB, false>), grid, dim3(kGemmThreads), 0, st, sj[0], sj[nj - 1], s2);
                GNF_LAUNCH_CHECK("k_gemm (split-K)");
                dim3 eg((unsigned)((sh.M * sh.N + 255) / 256), (unsigned)nj);
                hipLaunchKernelGGL((k_splitk_epilogue<EPI>), eg, dim3(256), 0, st, jobs[0], jobs[nj - 1], (const float*)sk[0],
                                   (const float*)sk[nj - 1], (int)s2.chunks, sh);
                GNF_LAUNCH_CHECK("k_splitk_epilogue");
                return GNF_OK;
            }
        }
    }
    const bool buf = gemm_buf_ok(AK, BK, AK == OPND_KC ? sh.M : sh.K, sh.lda, BK == OPND_KC ? (int64_t)sh.N : sh.K, sh.ldb, sh.K);
    // 128 x 128 tiles once they fill every CU's two slots (2 x 73.7 KB of LDS)
    // (the masked dX of a wide net's thin last layer keeps the 64-column tile: it runs beside the weight-gradient launch of the
    // previous half-step, and a 128-column workgroup's 240 registers per SIMD find no room there - round 6's kernel timeline)
    if (buf && EPI != EPI_SLAB && EPI != EPI_MASK && (int64_t)((sh.N + 127) / 128) * ((sh.M + TGM - 1) / TGM) * nj * sh.chunks >= 2 * (int64_t)big_cu_count()) {
        dim3 grid2((unsigned)((sh.N + 127) / 128), (unsigned)((sh.M + TGM - 1) / TGM), (unsigned)(nj * sh.chunks));
        hipLaunchKernelGGL((k_gemm<AK, BK, EPI, true, 128>), grid2, dim3(kGemmThreads), 0, st, jobs[0], jobs[nj - 1], sh);
        GNF_LAUNCH_CHECK("k_gemm (128-column tiles)");
        return GNF_OK;
    }
    dim3 grid((unsigned)((sh.N + TGN - 1) / TGN), (unsigned)((sh.M + TGM - 1) / TGM), (unsigned)(nj * sh.chunks));
    if (buf)
        hipLaunchKernelGGL((k_gemm<AK, BK, EPI, true>), grid, dim3(kGemmThreads), 0, st, jobs[0], jobs[nj - 1], sh);
    else
        hipLaunchKernelGGL((k_gemm<AK, BK, EPI, false>), grid, dim3(kGemmThreads), 0, st, jobs[0], jobs[nj - 1], sh);
    GNF_LAUNCH_CHECK("k_gemm");
    return GNF_OK;
}

// y = act(x W + b) for up to two nets sharing shapes: the layered forward path's matrix-core layers run through the
// same GEMM tile as the generic backward (split over the reduction when the layer is thin and the caller has a free
// ping-pong activation buffer to lend as slab scratch; sk may be NULL).
int launch_linear_splitk(const float* const* x, int64_t ldx, const float* const* W, const float* const* b, float* const* y,
                         int64_t ldy, int nj, int64_t n, int32_t I, int32_t O, int act, float alpha, int apply_act,
                         float* const* sk, size_t sk_floats, hipStream_t st) {
    if (n == 0) return GNF_OK;
    GemmJob jobs[2];
    for (int q = 0; q < nj; ++q) jobs[q] = GemmJob{x[q], W[q], y[q], b[q], nullptr};
    GemmShape sh;
    memset(&sh, 0, sizeof(sh));
    sh.lda = ldx, sh.ldb = O, sh.ldc = ldy;
    sh.M = n, sh.K = I, sh.N = O, sh.chunks = 1, sh.kchunk = TGK;
    sh.act = act, sh.alpha = alpha, sh.apply_act = apply_act;
    return launch_gemm<OPND_KC, OPND_MC, EPI_BIAS_ACT>(jobs, nj, sh, st, sk, sk_floats);
}

// G[e] (+)= sum over chunks of slab[chunk][e]   (fixed order)
struct ReduceJob {
    const float* wslab;
    const float* bslab;
    float* gw;
    float* gb;
};
template <int G>
struct GroupedReduceT {
    ReduceJob job[G];
    int64_t nw[G];
    int32_t nb[G];
    int32_t chunks[G];
    int32_t accumulate;
    // the index space of reduce_jobs_strided, laid out by the host (reduce_index_space): qpre[k] = 16-byte items of the jobs
    // in front of job k (k >= the job count: all of them), singles = the elements that go one by one.  (Every thread used
    // to add these up and then walk the job list for its item with scalar loads, one dependent trip to the kernel
    // arguments per job: 0.4 us per job - 4 of the 5.4 us a tile workgroup of the merged launch spent on its reduce share
    // with 10 jobs, 7 of 8.5 with 18: stamps, round 6.)
    int32_t qpre[G + 1];
    int32_t singles;
};
using GroupedReduce = GroupedReduceT<kMaxGroup>;
using GroupedReduceS = GroupedReduceT<kMergedGroup>;
// element e of job j (the weight gradient first, then the bias gradient)
// All jobs of a reduce as ONE index space, strided over `nthr` threads (this one is thread `t`): every item (four
// consecutive weight-gradient elements, or one element where a job cannot be read as float4) sums its slabs with up to
// eight chunk loads in flight, and a thread's items are independent of each other.  (The first version walked job after
// job and chunk after chunk, one dependent load at a time: 18 us for the config-2 nets on 85 workgroups against 5 us
// for k_reduce_grouped on the whole chip.)  The sum keeps chunk order, so the gradients are bitwise the same.
template <class T>
__device__ __forceinline__ T sum_slabs(const float* __restrict__ slab, const int64_t stride, const int64_t e, const int nchunk) {
    T s = T(0.f);
    for (int c = 0; c < nchunk; c += 8) {
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {  // indices clamped, adds predicated: one round trip for up to eight chunks
            const int cc = c + q < nchunk ? c + q : nchunk - 1;
            v[q] = *reinterpret_cast<const T*>(slab + (int64_t)cc * stride + e);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (c + q < nchunk) s += v[q];
    }
    return s;
}
__host__ __device__ inline bool reduce_job_vec_ok(const ReduceJob& job, int64_t nw) {
    return (nw & 3) == 0 && ((reinterpret_cast<uintptr_t>(job.wslab) | reinterpret_cast<uintptr_t>(job.gw)) & 15) == 0;
}
// fills GroupedReduceT.qpre / singles for the first nj jobs (call after the last change to job / nw / nb); false: the index
// space does not fit 31 bits (more than 8.6 G weight-gradient elements in one launch)
template <class GR>
static bool reduce_index_space(GR* g, int nj) {
    constexpr int G = (int)(sizeof(g->chunks) / sizeof(g->chunks[0]));
    int64_t quads = 0, singles = 0;
    for (int j = 0; j < G; ++j) {
        g->qpre[j] = (int32_t)quads;
        if (j >= nj) continue;
        if (reduce_job_vec_ok(g->job[j], g->nw[j]))
            quads += g->nw[j] >> 2;
        else
            singles += g->nw[j];
        singles += g->nb[j];
    }
    g->qpre[G] = (int32_t)quads;
    g->singles = (int32_t)singles;
    return quads <= INT32_MAX && singles <= INT32_MAX;
}
template <class GR>
__device__ __forceinline__ void reduce_jobs_strided(const GR& g, const int nj, const int64_t t, const int64_t nthr) {
    typedef float V4 __attribute__((ext_vector_type(4)));
    auto vec_ok = [&](int j) { return reduce_job_vec_ok(g.job[j], g.nw[j]); };
    constexpr int G = (int)(sizeof(g.chunks) / sizeof(g.chunks[0]));
    const int64_t quads = g.qpre[G], singles = g.singles;
    for (int64_t q = t; q < quads; q += nthr) {
        int j = 0;
        int64_t lq = q;
#pragma unroll
        for (int k = 1; k < G; ++k) {  // (the prefix sums arrive in a few wide scalar loads; the last k with qpre[k] <= q wins)
            const int64_t pk = g.qpre[k];
            if (q >= pk) j = k, lq = q - pk;
        }
        const ReduceJob job = g.job[j];
        const V4 s = sum_slabs<V4>(job.wslab, g.nw[j], 4 * lq, g.chunks[j]);
        V4* dst = reinterpret_cast<V4*>(job.gw + 4 * lq);
        *dst = g.accumulate ? *dst + s : s;
    }
    for (int64_t q = t; q < singles; q += nthr) {
        int j = 0;
        int64_t lq = q;
        bool bias = false;
        for (;; ++j) {
            const int64_t n1 = vec_ok(j) ? 0 : g.nw[j];
            if (lq < n1) break;
            lq -= n1;
            if (lq < g.nb[j]) {
                bias = true;
                break;
            }
            lq -= g.nb[j];
        }
        const ReduceJob job = g.job[j];
        const float s = bias ? sum_slabs<float>(job.bslab, g.nb[j], lq, g.chunks[j])
                             : sum_slabs<float>(job.wslab, g.nw[j], lq, g.chunks[j]);
        float* dst = (bias ? job.gb : job.gw) + lq;
        *dst = g.accumulate ? *dst + s : s;
    }
}

// (one axis over every job's elements, 16 bytes per thread and eight slabs in flight - reduce_jobs_strided, as the merged
// launch's tile workgroups do it; the per-element form, one 4-byte load per slab in a dependent loop, took 21.7 us per
// half-step on the 3 200-node batches of the drivers' default loop)
__global__ __launch_bounds__(256) void k_reduce_grouped(const GroupedReduce g, int nj) {
    reduce_jobs_strided(g, nj, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// g[N, D] <- z[N, D]: dL/dz of L = 1/2 sum z^2 + const - logdet
// (+ when rowptr != NULL: invdeg[i] = 1 / max(indeg(i), 1) for the mean aggregator's backward, in the same launch)
__global__ __launch_bounds__(256) void k_copy_rows(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                   int64_t ldd, int64_t n, int W, const int32_t* __restrict__ rowptr,
                                                   float* __restrict__ invdeg) {
    const int64_t total = n * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / W;
        const int f = (int)(i - r * W);
        dst[r * ldd + f] = src[r * lds_ + f];
    }
    if (rowptr)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const int dg = rowptr[i + 1] - rowptr[i];
            invdeg[i] = 1.f / (float)(dg > 1 ? dg : 1);
        }
}

// coupling backward (see the file header): y, g in place; gs, gt dense [N, H]
__global__ __launch_bounds__(256) void k_coupling_bwd(const float* __restrict__ s, const float* __restrict__ t,
                                                      float* __restrict__ y, int64_t ldy, float* __restrict__ g,
                                                      int64_t ldg, float* __restrict__ gs, float* __restrict__ gt,
                                                      int64_t n, int H, const float* __restrict__ xres, int64_t ldx) {
    const int64_t total = n * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / H;
        const int f = (int)(i - r * H);
        float sv = s[i], tv = t[i];
        if (xres) {  // residual attention block: s, t = MLP(h0) + x_cond (gnn.py:547-548)
            const float xr = xres[r * ldx + f];
            sv += xr;
            tv += xr;
        }
        const float yv = y[r * ldy + f], gv = g[r * ldg + f];
        const float d = yv - tv;
        y[r * ldy + f] = d * expf(-sv);   // the half-step's input, gnn.py:359,372
        g[r * ldg + f] = gv * expf(sv);
        gs[i] = gv * d - 1.f;             // through x_b exp(s), and -1 from -logdet
        gt[i] = gv;
    }
}

// message-passing backward: g_cond[u, f] += base + sum over edges u -> v of dh[v, aggcol + f] * w(v)
//   dh = dh_s + dh_t ([N, in0] each);  agg-combine: base = eps * dh[u, f], aggcol = 0;
//   concat: base = dh[u, f], aggcol = H;  w(v) = 1 / max(indeg(v), 1) for the mean aggregator (invdeg, or NULL).
// rowptr_t / col_t: CSR by SENDER (row u lists the receivers v of u's out-edges, in edge order).
// Same shape as kernel A (gnf_layered.hip): a group of G lanes owns a row, VEC floats per lane, four
// neighbour rows in flight; the adds stay in edge order.
template <int VEC>
__global__ __launch_bounds__(256) void k_aggregate_bwd(const int32_t* __restrict__ rowptr_t,
                                                       const int32_t* __restrict__ col_t,
                                                       const float* __restrict__ invdeg, int64_t n,
                                                       const float* __restrict__ dhs, const float* __restrict__ dht,
                                                       int in0, int H, int concat, float eps,
                                                       float* __restrict__ g, int64_t ldg, int G) {
    typedef float V __attribute__((ext_vector_type(VEC)));
    const int64_t nwg = gridDim.x, bid = blockIdx.x;
    const int64_t xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;  // XCD-aware remap as in kernel A
    const int64_t blk = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int64_t u = (blk * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (u >= n) return;
    const int beg = rowptr_t[u], end = rowptr_t[u + 1];
    const int aggcol = concat ? H : 0;
    auto ld2 = [&](int64_t o) -> V {
        return *reinterpret_cast<const V*>(dhs + o) + *reinterpret_cast<const V*>(dht + o);
    };
    for (int f = gl * VEC; f < H; f += G * VEC) {
        V acc = V(0.f);
        int e = beg;
        for (; e + 4 <= end; e += 4) {
            int vi[4];
            float w[4];
            V vv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) vi[q] = col_t[e + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w[q] = invdeg ? invdeg[vi[q]] : 1.f;
                vv[q] = ld2((int64_t)vi[q] * in0 + aggcol + f);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += vv[q] * w[q];
        }
        for (; e < end; ++e) {
            const int v = col_t[e];
            acc += ld2((int64_t)v * in0 + aggcol + f) * (invdeg ? invdeg[v] : 1.f);
        }
        const V own = ld2(u * in0 + f);
        V* pg = reinterpret_cast<V*>(g + u * ldg + f);
        *pg = *pg + (concat ? own : own * eps) + acc;
    }
}

// ---- workspace ---------------------------------------------------------------------------------
struct BwdPlan {
    int K, in0, lmax, H, chunks;
    int slab_chunks;  // slabs there is room for per job (>= chunks: thin jobs of a merged walk are cut finer, see plan_weight_grads)
    int64_t n, kchunk;
    int64_t wsum, osum;  // floats of dW / db slabs per chunk, both nets, attention weights included
    // attention geometry (0 when the nets are message-passing GNNs)
    int nh, kq, vd, C, P, NV;
    // float offsets.  single: g, invdeg, bnpart, st, wslab, bslab, qkv, dagg, stats;  per set: the rest
    size_t g, invdeg, bnpart, st, wslab, bslab, qkv, dagg, stats, splitk, lny, lnpart;
    size_t wot, wot_each;  // attention nets: Wo of every net of the flow as packed transposed fragments (k_pack_wot), floats per net
    size_t wct, wct_each;  // ... and [Wq | Wk | Wv]^T (attn_wct_floats) for the matrix-core form of dL/dx_cond
    size_t splitk_each;  // floats of split-K scratch per net
    size_t h0, h0b, acts, gst, dpb, dh0, xc, dqkv, agg;
    int n_sets;
    size_t set_stride;  // the dW operands exist n_sets times: half-step k's dW GEMMs run on the auxiliary stream while
                        // half-steps k+1, k+2 refill the other sets (with two sets the walk waited ~10 us per half-step
                        // for the dW launch of two half-steps earlier to release the set it was about to refill)
    int slab_sets;      // 2 where the merged backward + dW launch may run (the dW GEMMs of half-step k-1 write one
    size_t slab_stride; // set while the reduce of half-step k-2 reads the other), else 1;  floats between the sets
    size_t total;
};

static inline size_t al64(size_t v) { return (v + 63) / 64 * 64; }  // keep every region 256-byte aligned
// Operand sets of the dW GEMMs.  A set written by half-step k's backward kernel is read by its dW GEMMs on the auxiliary
// stream; before a later half-step may refill it, the main stream has to wait for that dW launch.  Three sets: the wait
// always finds its event fired.  (Measured, profiles/r2u_train_timeline.txt: each event record / wait between two
// dependent kernels shows as ~6 us of command-processor latency in the rocprofv3 timeline, fired or not - but one set
// per half-step, i.e. no wait on the main stream at all, changed nothing in the step time (2.41 vs 2.38 ms), and a
// captured-graph replay of the whole step is 2 % faster than eager launches: the gaps belong to the cross-queue
// dependency itself, not to the host or to the wait packets.  The generic code below still takes any set count.)
static constexpr int kTileLightChunks = 24;  // most pieces a thin job is cut into for the tile workgroups of a merged launch
static constexpr int kMergedMaxTiles = 192;  // backward tiles of a launch that still leaves CUs for the dW GEMMs
static constexpr int kLnBwdRows = 64;  // rows per workgroup of the layer-norm backward kernel
static constexpr int kBwdMaxSets = 64;
static constexpr int kBwdSetsDefault = 3;

// n_sets: operand sets (one per half-step of the walk, capped: see kBwdMaxSets)
static BwdPlan plan_backward(int64_t n, int32_t D, const GnfMlp* net, int n_sets, int n_nets_each = 0) {
    BwdPlan p;
    memset(&p, 0, sizeof(p));
    p.n = n;
    p.H = D / 2;
    p.K = net->num_layers;
    p.in0 = net->dims[0];
    int lmax = 1;
    int64_t wsum = 0, osum = 0;
    for (int j = 0; j < p.K; ++j) {
        if (j >= 1) lmax = lmax > net->dims[j] ? lmax : net->dims[j];
        wsum += (int64_t)net->dims[j] * net->dims[j + 1];
        osum += net->dims[j + 1];
    }
    if (net->attn) {
        const GnfAttn* at = net->attn;
        p.nh = at->num_heads, p.kq = at->kq_dim, p.vd = at->v_dim, p.C = at->out_dim;
        p.P = 2 * p.nh * p.kq + p.vd;
        p.NV = p.nh * p.vd;
        wsum += (int64_t)p.H * p.P + (int64_t)p.NV * p.C;
    }
    p.wsum = 2 * wsum;
    p.osum = 2 * osum;
    p.lmax = lmax;
    // split of the node axis for dW: every weight gradient of the half-step goes out in ONE grouped launch, so a
    // few chunks already fill the chip; fewer chunks = fewer slabs to write and reduce
    // (measured on the config-2 batch: chunks of 64 / 128 / 256 / 512 nodes give 3.11 / 2.94 / 2.94 / 2.99 ms per step)
    int64_t chunks = (n + 255) / 256;
    if (chunks > 64) chunks = 64;
    {   // ... but no more than it takes to put ~4 workgroups on every CU: wide layers have the tiles already (2048-wide:
        // 3072 tiles per chunk - a second chunk only added 200 MB of slab traffic per half-step)
        int64_t tiles = 0;
        for (int j = 0; j < p.K; ++j)
            tiles += (int64_t)((net->dims[j] + TGM - 1) / TGM) * ((net->dims[j + 1] + TGN - 1) / TGN);
        tiles *= 2;
        int64_t enough = (1024 + tiles - 1) / (tiles > 0 ? tiles : 1);
        // ... except that nets with wide hidden layers AND thin first / last layers need slab room for the thin layers' tiles:
        // the wide kernel gives every CU one workgroup that takes whole hidden-layer tiles by stride (one chunk each: the slab
        // is the gradient) and cuts the thin layers' tiles along the node axis so that they ride behind in equal pieces.  With
        // ONE chunk planned they could not be cut: on the data driver's nets 64 of the 256 workgroups carried a third
        // full-length tile (time stamps, round 6: 1 216 k cycles against 813 k for the other 192)
        if (lmax >= 512 && enough < 4) enough = 4;
        if (chunks > enough) chunks = enough;
    }
    if (chunks < 1) chunks = 1;
    int64_t kchunk = (n + chunks - 1) / chunks;
    kchunk = (kchunk + TGK - 1) / TGK * TGK;
    if (kchunk < TGK) kchunk = TGK;
    chunks = n > 0 ? (n + kchunk - 1) / kchunk : 1;
    p.chunks = (int)chunks;
    p.kchunk = kchunk;
    // merged walk (few tiles, nets the fused kernels hold): room for one short unit of the thin jobs per tile workgroup
    p.slab_chunks = p.chunks;
    if ((n + 15) / 16 <= kMergedMaxTiles && lmax < 512 && p.slab_chunks < kTileLightChunks) p.slab_chunks = kTileLightChunks;
    const size_t nk1 = (size_t)(p.K > 1 ? p.K - 1 : 0);
    size_t off = 0;
    p.g = off, off += al64((size_t)n * D);
    p.invdeg = off, off += al64((size_t)n);
    p.bnpart = off, off += al64(((size_t)kBnPartRowsMax + 1) * (size_t)p.H * 4);  // fp64 pairs of the batch-norm backward (+ one row: this rank's sums under cross-rank moments)
    p.st = off, off += 2 * al64((size_t)n * p.H);
    p.wslab = off, off += al64((size_t)p.slab_chunks * p.wsum);
    p.bslab = off, off += al64((size_t)p.slab_chunks * p.osum);
    p.slab_stride = off - p.wslab;
    p.slab_sets = (n + 15) / 16 <= kMergedMaxTiles ? 2 : 1;
    off += (size_t)(p.slab_sets - 1) * p.slab_stride;
    p.qkv = off, off += 2 * al64((size_t)n * p.P);
    p.dagg = off, off += 2 * al64((size_t)n * p.NV);
    p.stats = off, off += 2 * al64((size_t)n * 3 * p.nh);
    if (net->attn) {  // (s-nets first, then t-nets: n_nets_each of either)
        p.wot_each = al64((size_t)((p.C + 15) & ~15) * (size_t)((p.NV + 15) & ~15));
        p.wot = off, off += 2 * (size_t)n_nets_each * p.wot_each;
        p.wct_each = al64(attn_wct_floats(net->attn, p.H));
        p.wct = off, off += 2 * (size_t)n_nets_each * p.wct_each;
    }
    // slabs of the split-K path of thin generic-path GEMMs (launch_gemm): only launches with < 96 tiles take it, i.e.
    // fewer than 48 row tiles per net and one or two column tiles; 16 chunks at most
    p.splitk_each = al64((size_t)16 * (size_t)(n < 48 * TGM ? n : 48 * TGM) * (size_t)(2 * TGN));
    p.splitk = off, off += 2 * p.splitk_each;
    if (net->attn && net->attn->layer_norm) {  // snt.LayerNorm outputs (s, t) + per-workgroup partials of d gamma / d beta
        p.lny = off, off += 2 * al64((size_t)n * p.H);
        p.lnpart = off, off += al64((size_t)((n + kLnBwdRows - 1) / kLnBwdRows) * 4 * (size_t)p.H);
    }
    const size_t set0 = off;
    p.h0 = off, off += al64((size_t)n * p.in0);
    p.h0b = off, off += net->attn ? al64((size_t)n * p.in0) : 0;  // attention: one layer-0 input per net
    p.acts = off, off += 2 * nk1 * al64((size_t)n * lmax);   // [net][j = 1..K-1]: input of layer j
    p.gst = off, off += 2 * al64((size_t)n * p.H);           // dP of the last layer (g_s, g_t)
    p.dpb = off, off += 2 * nk1 * al64((size_t)n * lmax);    // [net][j = 0..K-2]: dP_j = dL/d(pre-activation of layer j)
    p.dh0 = off, off += 2 * al64((size_t)n * p.in0);
    p.xc = off, off += net->attn ? al64((size_t)n * p.H) : 0;  // the conditioning half as the attention dW GEMMs read it
    p.dqkv = off, off += 2 * al64((size_t)n * p.P);
    p.agg = off, off += 2 * al64((size_t)n * p.NV);
    p.set_stride = off - set0;
    // ... as long as the sets stay within ~4 GiB (large batches: the kernels are long, the event latency is noise)
    const int64_t fit = (int64_t)(((size_t)1 << 30) / (p.set_stride ? p.set_stride : 1));  // 2^30 floats
    if (n_sets > fit) n_sets = (int)fit;
    p.n_sets = n_sets < 3 ? 3 : (n_sets > kBwdMaxSets ? kBwdMaxSets : n_sets);
    off += (size_t)(p.n_sets - 1) * p.set_stride;             // the other sets
    p.total = off;
    return p;
}

// Wo ([NV, C], row major) of up to 64 attention blocks as packed TRANSPOSED fragments - k_pack_layer's wtout layout: k-groups
// over the C axis, column tiles over the NV axis - for the dagg row of the backward tile kernel (bwd_args_add_dagg_row)
struct PackWot {
    const float* wo[64];
    const float* wq[64];
    const float* wk[64];
    const float* wv[64];
    float* out[64];
    float* out_ct[64];  // [Wq | Wk | Wv]^T fragments: Bp[kg][nt][lane][q] = Wcat[16 nt + (lane & 15)][16 kg + 4 (lane >> 4) + q]
    int NV, C, NVp, Cp;
    int H, nq, vd, Pp, Hp;
};
__global__ __launch_bounds__(256) void k_pack_wot(const PackWot b) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    const int q = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
    if (i < b.NVp * b.Cp) {
        const int nts = b.NVp >> 4;
        const int kg = blk / nts, nt = blk - kg * nts;
        const int ko = 16 * kg + 4 * (lane >> 4) + q;  // along C
        const int ci = 16 * nt + (lane & 15);           // along NV
        b.out[blockIdx.y][i] = (ci < b.NV && ko < b.C) ? b.wo[blockIdx.y][(int64_t)ci * b.C + ko] : 0.f;
    }
    if (i < b.Pp * b.Hp) {
        const int nts = b.Hp >> 4;
        const int kg = blk / nts, nt = blk - kg * nts;
        const int c = 16 * kg + 4 * (lane >> 4) + q;   // along P: q columns, k columns, v columns
        const int f = 16 * nt + (lane & 15);            // along H
        float v = 0.f;
        if (f < b.H) {
            if (c < b.nq)
                v = b.wq[blockIdx.y][f * b.nq + c];
            else if (c < 2 * b.nq)
                v = b.wk[blockIdx.y][f * b.nq + (c - b.nq)];
            else if (c < 2 * b.nq + b.vd)
                v = b.wv[blockIdx.y][f * b.vd + (c - 2 * b.nq)];
        }
        b.out_ct[blockIdx.y][i] = v;
    }
}

static const GnfMlp* pick_net(const GnfFlow* f, const GnfMlp* nets, int half, int i) {
    return f->weight_sharing ? &nets[half] : &nets[half * f->num_timesteps + i];
}

// Buffers of one half-step, resolved for one operand set.
struct BwdOperands {
    float* hin[2 * GNF_MAX_LAYERS];   // [net * K + j]: input of layer j
    float* dPs[2 * GNF_MAX_LAYERS];   // [net * K + j]: dL/d(pre-activation of layer j)
    int64_t ldh[2 * GNF_MAX_LAYERS], lddp[2 * GNF_MAX_LAYERS];
    float* gst[2];
    float* dh0[2];
    float* h0[2];                     // message passing: both point at the shared h0
    float* stb[2];                    // s, t of the recompute (generic path)
    float* xc;
    float* dqkv[2];
    float* agg[2];
    float* qkv[2];
    float* dagg[2];
    float* stats[2];
};

static BwdOperands bwd_operands(const BwdPlan& p, float* ws, int set, bool attn) {
    BwdOperands o;
    memset(&o, 0, sizeof(o));
    const int64_t n = p.n;
    const int K = p.K;
    float* wss = ws + (size_t)set * p.set_stride;
    const size_t act_sz = al64((size_t)n * p.lmax);
    o.h0[0] = wss + p.h0;
    o.h0[1] = attn ? wss + p.h0b : o.h0[0];
    o.gst[0] = wss + p.gst;
    o.gst[1] = wss + p.gst + al64((size_t)n * p.H);
    o.dh0[0] = wss + p.dh0;
    o.dh0[1] = wss + p.dh0 + al64((size_t)n * p.in0);
    o.stb[0] = ws + p.st;
    o.stb[1] = ws + p.st + al64((size_t)n * p.H);
    o.xc = wss + p.xc;
    for (int q = 0; q < 2; ++q) {
        o.dqkv[q] = wss + p.dqkv + q * al64((size_t)n * p.P);
        o.agg[q] = wss + p.agg + q * al64((size_t)n * p.NV);
        o.qkv[q] = ws + p.qkv + (size_t)q * n * p.P;   // launch_attn_front's layout: net q at scratch + q * n * P
        o.dagg[q] = ws + p.dagg + q * al64((size_t)n * p.NV);
        o.stats[q] = ws + p.stats + q * al64((size_t)n * 3 * p.nh);
        for (int j = 0; j < K; ++j) {
            const int e = q * K + j;
            o.hin[e] = j == 0 ? o.h0[q] : wss + p.acts + ((size_t)q * (K - 1) + (j - 1)) * act_sz;
            o.ldh[e] = j == 0 ? p.in0 : p.lmax;
            o.dPs[e] = j == K - 1 ? o.gst[q] : wss + p.dpb + ((size_t)q * (K - 1) + j) * act_sz;
            o.lddp[e] = j == K - 1 ? p.H : p.lmax;
        }
    }
    return o;
}

// Every weight gradient of a half-step: dW = A^T B over the node axis (A, B row-major [nodes, *]) and, where
// gb != NULL, db = colsum(B).  One grouped split-K GEMM launch into per-chunk slabs + one fixed-order reduce
// launch into the gradient buffers.
struct WGJob {
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    int32_t M, N;
    float* gw;
    float* gb;
};

// How the dW launch is shaped.  max_units == 0: the 128 x 64 grouped kernel over the plan's chunks (many short
// workgroups that share CUs with whatever else runs).  max_units > 0: the wide kernel in at most that many
// workgroups, each asking for `lds` bytes of LDS - unless the cut this allows would take longer than `budget_us`,
// in which case the grouped kernel runs after all.
struct DwPolicy {
    int max_units;
    size_t lds;
    double budget_us;
    bool big_tiles_only = false;  // only the jobs with the most tiles count as "costliest" (see dw_policy, generic backward)
    int tile_wgs = 0;  // > 0: the launch that carries this plan has that many backward-tile workgroups with time to spare
                       // (merged walk with the MLP-row stash): a stream-K plan hands them the thin jobs (WideGemmT.light_on_tiles)
};

// The fused backward kernel of the NEXT half-step runs beside this half-step's dW GEMMs (one 16-node tile per
// workgroup, one workgroup per CU by its LDS footprint).  While it leaves CUs idle (config-2 batch: 170 tiles on 256
// CUs) the dW launch is sized to exactly those: long, equal workgroups, no more of them than there are idle CUs on
// every XCD (workgroups are dealt round-robin to the 8 XCDs of 32 CUs), with enough LDS padding that none fits beside
// a backward workgroup.  The backward kernel (the critical path) then runs undisturbed and the weight gradients
// finish inside its shadow.  One workgroup too many is costly - a backward workgroup that finds no CU waits for a whole
// dW workgroup - hence the strict cap.  Without enough idle CUs to finish in the backward kernel's time, the grouped
// kernel's many short workgroups spread the contention evenly instead.
static DwPolicy dw_policy(const GnfMlp* net, int64_t bwd_tiles, size_t bwd_lds) {
    const int64_t env_g = opt(OPT_DW_GROUPED), env_u = opt(OPT_DW_WIDE_UNITS);  // shape forcing for the parity tests (gnf_set_option)
    DwPolicy pol{0, kWideLdsMin, 0.0};
    if (env_g) return pol;
    if (bwd_tiles > 0 && bwd_tiles <= 192 && bwd_lds > 80 * 1024) {
        // workgroups are dealt round-robin to 32 shader engines of 8 CUs: no engine may get more exclusive workgroups
        // (backward + dW) than it has CUs, or a backward workgroup waits for a whole dW workgroup (measured: the
        // backward kernel then takes 150 instead of 84 us every other half-step)
        pol.max_units = 32 * (8 - (int)((bwd_tiles + 31) / 32));
        const size_t excl = (size_t)160 * 1024 - bwd_lds + 1024;
        if (excl > pol.lds) pol.lds = excl;
        // what the backward kernel takes: recompute + dP chain = 2 x 2 nets x 16 rows x sum(d_j d_j+1) MACs per tile at
        // ~60 % of a CU's fp32 matrix rate (measured: 84 us at 5 x 256-wide layers)
        double macs = 0.0;
        for (int j = 0; j < net->num_layers; ++j) macs += (double)net->dims[j] * net->dims[j + 1];
        // the alternative (grouped kernel sharing every CU) stretches the backward kernel by ~1.3 x
        pol.budget_us = 1.4 * (2.0 * 2.0 * 16.0 * macs * 2.0) / (614e9 * 0.6) * 1e6 + 10.0;
    }
    if (bwd_tiles == 0) {
        // generic backward (nets too wide for the fused kernels: the data driver's 2048 x 3 MLPs): a hidden layer's dW is
        // hundreds of full 128 x 128 tiles over the whole node axis - the wide kernel with ONE chunk per tile (the slab IS
        // the gradient: no reduce pass) where three workgroups per CU hold them all, against the grouped kernel's 128 x 64
        // tiles + slabs + reduce: wide_fc_train 34.2 -> 29.8 ms per step (measured with dw_wide_units = 640 .. 4096)
        int lmax = 1;
        for (int j = 1; j < net->num_layers; ++j) lmax = lmax > net->dims[j] ? lmax : net->dims[j];
        // ONE workgroup per CU, each taking its share of the hidden layers' 2 x 256 full tiles by stride (one chunk per tile:
        // the slab is the gradient) with the thin first / last layers' tiles cut into pieces that ride behind through slabs of
        // their own.  The count hardly matters (measured 128 .. 512 workgroups: 29.56 .. 29.90 ms per wide_fc_train step,
        // best at one per CU): the step is bound by the matrix cores under these kernels' efficiencies, and one workgroup per
        // CU leaves the other LDS slot of every CU to the main stream's kernels
        if (lmax >= 512) pol.max_units = big_cu_count(), pol.budget_us = 1e30, pol.big_tiles_only = true;
    }
    if (env_u > 0) pol.max_units = (int)env_u, pol.budget_us = 1e30;
    return pol;
}

// Everything launch_weight_grads decides, kept so that the GEMM launch and the reduce launch can be issued apart (or
// ride in another launch: the merged backward + dW kernel below).
struct DwLaunch {
    bool wide, buf, direct;
    WideGemm wg;
    GroupedGemm gg;
    GroupedReduce gr;
    int units, nj;
    size_t lds;
    int64_t maxred;
    int gx, gy, nz, groups, rpg;  // grouped kernel geometry
    unsigned blocks;
    bool gbuf;
};

// slab_set: which of the plan's slab sets (BwdPlan.slab_sets) the GEMM writes and the reduce reads
static int plan_weight_grads(const BwdPlan& p, const DwPolicy& pol, const WGJob* jobs, int nj, bool accumulate,
                             float* ws, int slab_set, DwLaunch* L) {
    GroupedGemm& gg = L->gg;
    GroupedReduce& gr = L->gr;
    memset(&gg, 0, sizeof(gg));
    memset(&gr, 0, sizeof(gr));
    L->nj = nj;
    L->lds = pol.lds;
    L->direct = false;
    ws += (size_t)slab_set * p.slab_stride;
    int maxM = 1, maxN = 1;
    int64_t maxred = 1;
    int64_t woff = 0, boff = 0;
    for (int e = 0; e < nj; ++e) {
        const WGJob& j = jobs[e];
        float* wsl = ws + p.wslab + (size_t)p.slab_chunks * woff;
        float* bsl = j.gb ? ws + p.bslab + (size_t)p.slab_chunks * boff : nullptr;
        gg.job[e] = GemmJob{j.A, j.B, wsl, nullptr, bsl};
        gg.lda[e] = j.lda;
        gg.ldb[e] = j.ldb;
        gg.M[e] = j.M;
        gg.N[e] = j.N;
        gr.job[e] = ReduceJob{wsl, bsl, j.gw, j.gb};
        gr.nw[e] = (int64_t)j.M * j.N;
        gr.nb[e] = j.gb ? j.N : 0;
        maxM = maxM > j.M ? maxM : j.M;
        maxN = maxN > j.N ? maxN : j.N;
        const int64_t red = (int64_t)j.M * j.N + (j.gb ? j.N : 0);
        maxred = maxred > red ? maxred : red;
        woff += (int64_t)j.M * j.N;
        if (j.gb) boff += j.N;
    }
    gg.K = p.n;
    gg.kchunk = p.kchunk;
    gg.chunks = p.chunks;
    for (int e = 0; e < nj; ++e) gr.chunks[e] = p.chunks;
    gr.accumulate = accumulate ? 1 : 0;
    // ---- wide kernel: cut every job along the node axis so that the workgroups carry about equal MFMA work --------
    bool wide = pol.max_units > 0 && p.n > 0;
    WideGemm& wg = L->wg;
    int units = 0;
    int64_t max_kchunk = WGK;
    if (wide) {
        memset(&wg, 0, sizeof(wg));
        // MFMA tiles per k-slice of the busiest wave of a job's busiest tile (what one step of such a workgroup costs;
        // 16 for a full 128 x 128 tile), rounded the way the kernel rounds (1 / 2 / 4 MFMA tiles per side)
        auto side = [](int len) {
            const int t16 = (len + 15) / 16;
            return t16 > 2 ? 4 : t16;
        };
        int cost[kMaxGroup], tiles_of[kMaxGroup], order[kMaxGroup], cj[kMaxGroup];
        for (int e = 0; e < nj; ++e) {
            const int mt_ = jobs[e].M < WGM ? jobs[e].M : WGM, nt_ = jobs[e].N < WGN ? jobs[e].N : WGN;
            const WideLayout l = wide_layout(mt_, nt_);
            const int wext_m = l.along_m ? 16 : 32, wext_n = l.along_n ? 16 : (l.along_m ? 32 : 64);
            cost[e] = 2 * side(mt_ < wext_m ? mt_ : wext_m) * side(nt_ < wext_n ? nt_ : wext_n);  // two waves per SIMD
            tiles_of[e] = ((jobs[e].M + WGM - 1) / WGM) * ((jobs[e].N + WGN - 1) / WGN);
            order[e] = e;
        }
        for (int a = 1; a < nj; ++a) {  // insertion sort, costliest tiles first (stable): they are dispatched first
            const int v = order[a];
            int b = a - 1;
            while (b >= 0 && cost[order[b]] < cost[v]) order[b + 1] = order[b], --b;
            order[b + 1] = v;
        }
        // The costliest jobs (the hidden-layer matrices) give the launch its grid: the smallest per-workgroup work T (in
        // rows of a full tile) whose cut of THEM fits max_units; candidates n / k.  Cheaper jobs (thin first / last
        // layers, attention projections) are cut into about as many units as there are workgroups and ride behind.
        if (pol.big_tiles_only) {  // a job with under a quarter of the largest job's tiles is not among the costliest
            int tmax = 1;
            for (int e = 0; e < nj; ++e) tmax = tmax > tiles_of[e] ? tmax : tiles_of[e];
            for (int e = 0; e < nj; ++e)
                if (4 * tiles_of[e] < tmax && cost[e] > 1) cost[e] -= 1;
            for (int a = 1; a < nj; ++a) {  // (re-sort: costliest first, stable)
                const int v = order[a];
                int b = a - 1;
                while (b >= 0 && cost[order[b]] < cost[v]) order[b + 1] = order[b], --b;
                order[b + 1] = v;
            }
        }
        const int cmax = nj > 0 ? cost[order[0]] : 16;
        int heavy_tiles = 0, light_tiles = 0;
        for (int e = 0; e < nj; ++e) (cost[e] == cmax ? heavy_tiles : light_tiles) += tiles_of[e];
        int c_heavy = 0;
        for (int k = p.chunks; k >= 1; --k)
            if ((int64_t)k * heavy_tiles <= pol.max_units) { c_heavy = k; break; }
        // (generic backward of wide nets: more whole tiles than workgroups is fine - one chunk each, taken by stride)
        if (c_heavy == 0 && pol.big_tiles_only && heavy_tiles > 0) c_heavy = 1;
        int grid = 0, c_light = 1;
        double est_us = 1e30;
        if (c_heavy > 0) {
            int64_t kc = (p.n + c_heavy - 1) / c_heavy;
            kc = (kc + WGK - 1) / WGK * WGK;
            c_heavy = (int)((p.n + kc - 1) / kc);
            grid = c_heavy * heavy_tiles;
            if (grid > pol.max_units) grid = pol.max_units;
            bool light_own = false;  // room for the cheap units as workgroups of their own?
            if (light_tiles > 0) {
                light_own = grid + light_tiles <= pol.max_units;
                c_light = light_own ? (pol.max_units - grid) / light_tiles : grid / light_tiles;
                c_light = c_light < 1 ? 1 : (c_light > p.chunks ? p.chunks : c_light);
            }
            // a step of a full tile (32 rows) takes ~2.1 us at one workgroup per CU (measured); + launch, prologue and
            // epilogue of every unit
            const double heavy_us = (double)cmax / 16.0 * (double)kc / 32.0 * 2.1 + 6.0;
            double light_us = 0.0;
            for (int e = 0; e < nj; ++e)
                if (cost[e] != cmax)
                    light_us += tiles_of[e] * c_light * ((double)cost[e] / 16.0 * (double)p.n / c_light / 32.0 * 2.1 + 4.0);
            est_us = light_own ? heavy_us : heavy_us + light_us / grid;
            if (light_own) grid += light_tiles * c_light;  // an upper bound; the exact count follows below
        }
        // ---- stream-K for the costliest jobs: equal runs of k-steps across tile boundaries instead of whole chunks ----
        // (24 tiles on 64 workgroups is 2.67 workgroups per tile: cut in whole chunks that is 2 per tile = 48 busy
        // workgroups with 43 steps each; as one axis of 24 x 85 steps it is 64 workgroups with 32 steps each)
        const int sk_steps = (int)((p.n + WGK - 1) / WGK);
        int sk_q = 0, sk_grid = 0, sk_cmax = 0;
        bool sk_on_tiles = false;
        if (c_heavy > 0 && heavy_tiles > 0 && heavy_tiles <= pol.max_units && sk_steps >= 8) {
            sk_grid = pol.max_units;
            const int64_t total_steps = (int64_t)heavy_tiles * sk_steps;
            sk_q = (int)((total_steps + sk_grid - 1) / sk_grid);
            if (sk_q < 4) sk_q = 4;
            sk_grid = (int)((total_steps + sk_q - 1) / sk_q);
            for (int t = 0; t < heavy_tiles; ++t) {  // pieces of tile t = workgroups whose run touches it
                const int c = (int)(((int64_t)(t + 1) * sk_steps - 1) / sk_q - ((int64_t)t * sk_steps) / sk_q + 1);
                sk_cmax = sk_cmax > c ? sk_cmax : c;
            }
            const double sk_us = (double)cmax / 16.0 * sk_q * 2.0 + 2 * 4.0 + 2.0;
            double light_us = 0.0;
            // (the thin jobs go to the tile workgroups of the carrying launch where it has them: cut for their count)
            const bool on_tiles = pol.tile_wgs > 0 && light_tiles > 0;
                        // (0.6 / 0.8 / 1.0 / 1.2 units per tile workgroup: 1.874 / 1.862 / 1.858 / 1.903 ms per config2_train step)
            int cl = light_tiles > 0 ? (on_tiles ? pol.tile_wgs : sk_grid) / light_tiles : 1;
            const int cl_max = on_tiles ? p.slab_chunks : p.chunks;
            cl = cl < 1 ? 1 : (cl > cl_max ? cl_max : cl);
            for (int e = 0; e < nj; ++e)
                if (cost[e] != cmax)
                    light_us += tiles_of[e] * cl * ((double)cost[e] / 16.0 * (double)p.n / cl / 32.0 * 2.1 + 4.0);
            const double sk_est = on_tiles ? sk_us : sk_us + light_us / sk_grid;
            sk_on_tiles = on_tiles;
            if (sk_q > sk_steps || sk_cmax > p.chunks || sk_est >= est_us) {
                sk_q = 0;  // not better than whole chunks (or the slabs were not planned for that many pieces)
                sk_on_tiles = false;
            } else {
                est_us = sk_est;
                c_light = cl;
                grid = sk_grid;
            }
        }
        if (grid == 0 || est_us > pol.budget_us) wide = false;
        int sk_jobs = 0;
        for (int q = 0; q < nj && wide; ++q) {
            const int e = order[q];
            wg.job[q] = gg.job[e];
            wg.lda[q] = gg.lda[e], wg.ldb[q] = gg.ldb[e];
            wg.M[q] = gg.M[e], wg.N[q] = gg.N[e];
            wg.gx[q] = (gg.N[e] + WGN - 1) / WGN;
            if (sk_q > 0 && cost[e] == cmax) {  // stream-K job: unit_base counts its tiles, chunks = slabs to reduce
                cj[e] = sk_cmax;
                wg.chunks[q] = sk_cmax;
                wg.kchunk[q] = (int32_t)((int64_t)sk_q * WGK);
                max_kchunk = max_kchunk > (int64_t)sk_q * WGK ? max_kchunk : (int64_t)sk_q * WGK;
                wg.unit_base[q] = units;
                units += tiles_of[e];
                sk_jobs = q + 1;
                continue;
            }
            int64_t c = cost[e] == cmax ? c_heavy : c_light;
            int64_t kc = (p.n + c - 1) / c;
            kc = (kc + WGK - 1) / WGK * WGK;
            c = (p.n + kc - 1) / kc;
            cj[e] = (int)c;
            wg.chunks[q] = (int32_t)c;
            wg.kchunk[q] = (int32_t)kc;
            max_kchunk = max_kchunk > kc ? max_kchunk : kc;
            wg.unit_base[q] = units;
            units += tiles_of[e] * (int)c;
        }
        if (wide) {
            for (int e = 0; e < nj; ++e) gr.chunks[e] = cj[e];
            // a job in ONE chunk with nothing to add to: its slab is the gradient itself - written in place, nothing for the
            // reduce launch to do for it (every job so: no reduce launch)
            bool all_one = !accumulate && sk_q == 0;
            for (int q = 0; q < nj && !accumulate; ++q) {
                const int e = order[q];
                const bool sk_job = sk_q > 0 && cost[e] == cmax;
                if (cj[e] == 1 && !sk_job) {
                    wg.job[q].C = jobs[e].gw, wg.job[q].aux_out = jobs[e].gb;
                    gr.nw[e] = 0, gr.nb[e] = 0;
                } else {
                    all_one = false;
                }
            }
            if (all_one) L->direct = true;
            if (pol.big_tiles_only && sk_q == 0) {  // XCD blocks for the leading whole-tile jobs (see WideGemmT.xcd_jobs)
                int xj = 0;
                while (xj < nj && cost[order[xj]] == cmax && cj[order[xj]] == 1 && tiles_of[order[xj]] % 8 == 0 &&
                       wg.unit_base[xj] % 8 == 0)
                    ++xj;
                wg.xcd_jobs = (grid % 8 == 0) ? xj : 0;
            }
            wg.unit_base[nj] = units;
            wg.K = p.n, wg.njobs = nj;
            if (sk_q > 0) {
                wg.sk_q = sk_q, wg.sk_steps = sk_steps, wg.sk_tiles = heavy_tiles, wg.sk_jobs = sk_jobs;
                wg.sk_light_base = wg.unit_base[sk_jobs];
                wg.light_on_tiles = sk_on_tiles ? 1 : 0;
                units = grid;
            } else {
                units = grid < units ? grid : units;  // workgroups; units past the grid are picked up by stride
            }
        }
    }
    const bool wide_direct = wide && L->direct;
    L->wide = wide;
    L->units = units;
    if (!reduce_index_space(&gr, nj)) {
        set_error("weight gradients: %d jobs with more than 2^31 reduce items in one launch", nj);
        return GNF_EUNSUPPORTED;
    }
    L->maxred = maxred;
    L->direct = wide_direct;
    L->buf = false;
    if (wide) {
        // buffer path: byte offsets inside a chunk (+ the two tiles fetched past its end) stay below 2^31.  Rows need
        // not be 16-byte aligned: buffer_load_dwordx4 only asks for dword alignment (the attention projections have
        // 170-float rows)
        bool buf = true;
        for (int e = 0; e < nj; ++e)
            buf = buf && (max_kchunk + 4 * WGK) * (jobs[e].lda > jobs[e].ldb ? jobs[e].lda : jobs[e].ldb) * 4 < ((int64_t)1 << 31);
        L->buf = buf;
    } else {
        const int gx = (maxN + TGN - 1) / TGN, gy = (maxM + TGM - 1) / TGM, nz = nj * p.chunks;
        int groups = nz >= 32 ? 1 : (32 + nz - 1) / nz;   // enough bands that every XCD gets several
        if (groups > gy) groups = gy;
        const int rpg = (gy + groups - 1) / groups;
        groups = (gy + rpg - 1) / rpg;
        const int nv = nz * groups;
        L->blocks = 8u * (unsigned)((nv + 7) / 8) * (unsigned)(gx * rpg);
        L->gx = gx, L->gy = gy, L->nz = nz, L->groups = groups, L->rpg = rpg;
        if (p.chunks == 1 && !accumulate) {  // one chunk: its "slab" IS the gradient - written in place, no reduce pass
            for (int e = 0; e < nj; ++e) gg.job[e].C = jobs[e].gw, gg.job[e].aux_out = jobs[e].gb;
            L->direct = true;
        }
        bool gbuf = true;
        for (int e = 0; e < nj; ++e)
            gbuf = gbuf && gemm_buf_ok(OPND_MC, OPND_MC, p.kchunk, jobs[e].lda, p.kchunk, jobs[e].ldb, p.n);
        L->gbuf = gbuf;
    }
    return GNF_OK;
}

static int run_weight_gemms(const DwLaunch& L, hipStream_t st) {
    if (L.wide) {
        GNF_ONCE_PER_DEVICE(
            GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dw_wide<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_dw_wide<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
        if (L.buf)
            hipLaunchKernelGGL(k_gemm_dw_wide<true>, dim3((unsigned)L.units), dim3(kWideThreads), L.lds, st, L.wg);
        else
            hipLaunchKernelGGL(k_gemm_dw_wide<false>, dim3((unsigned)L.units), dim3(kWideThreads), L.lds, st, L.wg);
        GNF_LAUNCH_CHECK("k_gemm_dw_wide");
    } else {
        if (L.gbuf)
            hipLaunchKernelGGL(k_gemm_dw_grouped<true>, dim3(L.blocks), dim3(kGemmThreads), 0, st, L.gg, L.gx, L.gy, L.nz,
                               L.groups, L.rpg);
        else
            hipLaunchKernelGGL(k_gemm_dw_grouped<false>, dim3(L.blocks), dim3(kGemmThreads), 0, st, L.gg, L.gx, L.gy, L.nz,
                               L.groups, L.rpg);
        GNF_LAUNCH_CHECK("k_gemm_dw_grouped");
    }
    return GNF_OK;
}

static int run_weight_reduce(const DwLaunch& L, hipStream_t st) {
    if (L.direct) return GNF_OK;
    int64_t blocks = ((L.maxred + 3) / 4 * L.nj + 255) / 256;  // about one 16-byte quad per thread
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(k_reduce_grouped, dim3((unsigned)blocks), dim3(256), 0, st, L.gr, (int)L.nj);
    GNF_LAUNCH_CHECK("k_reduce_grouped");
    return GNF_OK;
}

static int launch_weight_grads(const BwdPlan& p, const DwPolicy& pol, const WGJob* jobs, int nj, bool accumulate,
                               float* ws, hipStream_t st) {
    DwLaunch L;
    int rc = plan_weight_grads(p, pol, jobs, nj, accumulate, ws, 0, &L);
    if (rc) return rc;
    rc = run_weight_gemms(L, st);
    if (rc) return rc;
    return run_weight_reduce(L, st);
}

// ---- one launch per half-step of the backward walk (small batches) ------------------------------------------------
// A batch of up to kMergedMaxTiles 16-node tiles leaves CUs idle under the fused backward kernel (config-2 batch: 170
// tiles on 256 CUs).  Until round 2 the weight-gradient GEMMs of a half-step ran on a second stream into exactly those
// CUs; every fork / join event between the two queues cost ~6 us of command-processor latency on the critical path
// (profiles/r2u_train_timeline.txt).  Here ONE launch carries, software-pipelined,
//   workgroups [0, n_bwd)        the backward kernel of half-step k            (half_bwd_body, one 16-node tile each)
//   workgroups [n_bwd, n_bwd+n_dw) the dW GEMMs of half-step k-1                 (dw_wide_body; its operands are complete)
//   all workgroups behind n_bwd  the fixed-order slab reduce of half-step k-2  (reduce_jobs_strided; slab sets alternate)
// so the walk needs no second stream, no events, and the kernel boundary is the only synchronisation.  Every
// workgroup asks for the backward kernel's LDS footprint (> 80 KB), i.e. one workgroup per CU.
template <int MT, bool STASHED>
__global__ __launch_bounds__(kBwdThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_half_bwd_dw(const BwdArgs a, const WideGemmS g, const GroupedReduceS r, const int n_bwd, const int n_dw, const int r_nj) {
    const int bid = (int)blockIdx.x;
    int vb = -1, vg = 1, it0 = 0, it1 = 0x7fffffff;  // this workgroup's place in the weight-gradient work (vb < 0: none)
    if (bid < n_bwd) {
        half_bwd_body<MT, STASHED>(a, bid, n_bwd);
        // the tiles finish before the dW workgroups do (with the MLP-row stash after about 60 % of their time): the slab
        // reduce is theirs
        // (recomputing tiles are the longer side: there the reduce stays with the dW workgroups - 2.18 vs 2.29 ms per step)
        if (STASHED && r_nj > 0)
            reduce_jobs_strided(r, r_nj, (int64_t)bid * kBwdThreads + threadIdx.x, (int64_t)n_bwd * kBwdThreads);
        if (!STASHED || !g.light_on_tiles) return;
        __syncthreads();  // (the tile's last rows have left the LDS buffers the operand stages overlay)
        // (units are dealt from the LAST tile workgroup down: where there are fewer units than workgroups, the ones left
        // without are the first ones - whose threads carry the second round of the strided reduce above)
        vb = n_bwd - 1 - bid, vg = n_bwd, it0 = 2;
    } else {
        const int v = bid - n_bwd;
        if (v < n_dw) {
            vb = v, vg = n_dw;
            if (STASHED && g.light_on_tiles && n_bwd > 0) it1 = 2;
        }
    }
    if (vb >= 0) dw_wide_body<true>(g, vb, vg, it0, it1);
    if (bid >= n_bwd && r_nj > 0 && (!STASHED || n_bwd == 0))
        reduce_jobs_strided(r, r_nj, (int64_t)(bid - n_bwd) * kBwdThreads + threadIdx.x,
                            (int64_t)((int)gridDim.x - n_bwd) * kBwdThreads);
}
static_assert(kBwdThreads == kWideThreads, "the merged launch runs both bodies with one workgroup size");
static_assert(sizeof(BwdArgs) + sizeof(WideGemmS) + sizeof(GroupedReduceS) + 16 <= 4096, "kernel arguments exceed 4 KB");

static void narrow_wide(const WideGemm& w, WideGemmS* o) {
    memset(o, 0, sizeof(*o));
    for (int q = 0; q < kMergedGroup && q < w.njobs; ++q) {
        o->job[q] = w.job[q];
        o->lda[q] = w.lda[q], o->ldb[q] = w.ldb[q];
        o->M[q] = w.M[q], o->N[q] = w.N[q];
        o->unit_base[q] = w.unit_base[q];
        o->gx[q] = w.gx[q], o->chunks[q] = w.chunks[q], o->kchunk[q] = w.kchunk[q];
    }
    o->unit_base[w.njobs] = w.unit_base[w.njobs];
    o->K = w.K, o->njobs = w.njobs;
    o->sk_q = w.sk_q, o->sk_steps = w.sk_steps, o->sk_tiles = w.sk_tiles, o->sk_jobs = w.sk_jobs;
    o->sk_light_base = w.sk_light_base;
    o->xcd_jobs = 0;
    o->light_on_tiles = w.light_on_tiles;
}
static void narrow_reduce(const GroupedReduce& r, int nj, GroupedReduceS* o) {
    memset(o, 0, sizeof(*o));
    for (int q = 0; q < kMergedGroup && q < nj; ++q) {
        o->job[q] = r.job[q];
        o->nw[q] = r.nw[q], o->nb[q] = r.nb[q], o->chunks[q] = r.chunks[q];
    }
    o->accumulate = r.accumulate;
    (void)reduce_index_space(o, nj < kMergedGroup ? nj : kMergedGroup);  // (a subset of a plan that fitted)
}

// bwd: the half-step to walk (NULL: none - the tail of the pipeline); dw: GEMMs to run beside it (NULL: none);
// red: a finished GEMM launch whose slabs are due (NULL: none)
static int launch_half_bwd_dw(const BwdArgs* bwd, int64_t bwd_tiles, size_t bwd_lds, const DwLaunch* dw, const DwLaunch* red,
                              bool stashed, hipStream_t st) {
    static const BwdArgs kNoBwd = {};
    WideGemmS g;
    GroupedReduceS r;
    memset(&g, 0, sizeof(g));
    memset(&r, 0, sizeof(r));
    int n_dw = 0, r_nj = 0;
    if (dw) narrow_wide(dw->wg, &g), n_dw = dw->units;
    if (red && !red->direct) narrow_reduce(red->gr, red->nj, &r), r_nj = red->nj;
    const int n_bwd = bwd ? (int)bwd_tiles : 0;
    int grid = n_bwd + n_dw;
    if (r_nj > 0 && grid == n_bwd) grid += 64;  // reduce only: some workgroups to carry it
    if (grid == 0) return GNF_OK;
    GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_bwd_dw<1, false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_bwd_dw<1, true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)));
    size_t lds = bwd_lds > kWideLdsMin ? bwd_lds : kWideLdsMin;
    if (lds <= 80 * 1024) lds = 80 * 1024 + 256;  // one workgroup per CU whatever the layer widths: a backward tile never shares its CU
    if (stashed)
        hipLaunchKernelGGL((k_half_bwd_dw<1, true>), dim3((unsigned)grid), dim3(kBwdThreads), lds, st, bwd ? *bwd : kNoBwd, g, r, n_bwd, n_dw, r_nj);
    else
        hipLaunchKernelGGL((k_half_bwd_dw<1, false>), dim3((unsigned)grid), dim3(kBwdThreads), lds, st, bwd ? *bwd : kNoBwd, g, r, n_bwd, n_dw, r_nj);
    GNF_LAUNCH_CHECK("k_half_bwd_dw");
    return GNF_OK;
}

// the backward half of the merged-launch / stash conditions (the forward half: fused_stash_shape)
static bool merged_walk_ok(const GnfFlow* flow, int64_t n, int64_t* tiles_out, size_t* lds_out) {
    if (opt(OPT_BWD_GENERIC) || opt(OPT_DW_GROUPED)) return false;
    if (flow->s_nets[0].attn && 2 * (flow->s_nets[0].num_layers + 4) > kMergedGroup) return false;
    const int n_nets = flow->weight_sharing ? 2 : 2 * flow->num_timesteps;
    for (int q = 0; q < n_nets; ++q)
        if (!fused_bwd_supported(&flow->s_nets[q], &flow->t_nets[q])) return false;
    int64_t tiles;
    size_t lds;
    fused_bwd_launch_shape(&flow->s_nets[0], n, &tiles, &lds);
    if (tiles_out) *tiles_out = tiles;
    if (lds_out) *lds_out = lds;
    return tiles == (n + 15) / 16 && tiles <= kMergedMaxTiles;
}

// The stash's second mode: nets whose forward runs the LAYERED path (too wide for the fused kernels - the data driver's
// 2048 x 3 MLPs - or unpacked) and whose backward is the generic GEMM path.  There the backward walk recomputed every
// hidden layer of both nets per half-step (a quarter of a training step's flops at that width); with the stash the forward's
// layer outputs and s, t go straight into the half-step's slot (same layout, no act' ballots used) and the recompute loop of
// mlp_backward_generic is skipped.  Memory for time, like the fused mode: 2 T slots of (in0 + 2 (K - 1) lmax + 2 H) n floats
// (1.8 GB for the data driver's defaults on a 2 718-node batch), bounded at 48 GB.
bool layered_stash_mode(const GnfFlow* flow, int64_t n, int32_t H) {
    if (n <= 0 || !flow->s_nets || !flow->t_nets || flow->num_timesteps < 1) return false;
    const int n_nets = flow->weight_sharing ? 2 : 2 * flow->num_timesteps;
    for (int q = 0; q < n_nets; ++q) {
        const GnfMlp *s = &flow->s_nets[q], *t = &flow->t_nets[q];
        if (s->num_layers != t->num_layers || s->num_layers < 2) return false;
        for (int j = 0; j <= s->num_layers; ++j)
            if (s->dims[j] != t->dims[j]) return false;
        const bool fwd_fused = s->packed && t->packed && fused_fits_lds(s);   // (fused_supported's rule)
        if (fwd_fused || fused_bwd_supported(s, t) || (s->attn && s->attn->layer_norm)) return false;
    }
    const size_t bytes = (size_t)2 * flow->num_timesteps * mlp_stash_layout(&flow->s_nets[0], n, H).slot * sizeof(float);
    return bytes <= ((size_t)48 << 30);
}

bool mlp_stash_supported(const GnfFlow* flow, int64_t n, int32_t H) {
    if (layered_stash_mode(flow, n, H)) return true;
    if (opt(OPT_BWD_GENERIC) || n <= 0 || !flow->s_nets || !flow->t_nets) return false;
    const int n_nets = flow->weight_sharing ? 2 : 2 * flow->num_timesteps;
    for (int q = 0; q < n_nets; ++q)
        if (!fused_stash_shape(&flow->s_nets[q], &flow->t_nets[q], n) || !fused_bwd_supported(&flow->s_nets[q], &flow->t_nets[q]))
            return false;
    int64_t tiles;
    size_t lds;
    fused_bwd_launch_shape(&flow->s_nets[0], n, &tiles, &lds);
    return tiles == (n + 15) / 16 && tiles <= big_cu_count();  // 16-node tiles in both directions (one per CU at most)
}

// job list of one half-step: the K layers of both nets, then the four attention matrices of both nets
static int weight_grad_jobs(const BwdPlan& p, const BwdOperands& o, const GnfMlp* const* nets,
                            const GnfMlp* const* grads, WGJob* jobs) {
    const int K = p.K;
    int nj = 0;
    for (int q = 0; q < 2; ++q)
        for (int j = 0; j < K; ++j) {
            const int e = q * K + j;
            jobs[nj++] = WGJob{o.hin[e], o.ldh[e], o.dPs[e], o.lddp[e], nets[q]->dims[j], nets[q]->dims[j + 1],
                               const_cast<float*>(grads[q]->W[j]), const_cast<float*>(grads[q]->b[j])};
        }
    if (nets[0]->attn)
        for (int q = 0; q < 2; ++q) {
            const GnfAttn* ga = grads[q]->attn;
            const int nq = p.nh * p.kq, off = nets[q]->attn->concat ? p.H : 0;
            jobs[nj++] = WGJob{o.xc, p.H, o.dqkv[q], p.P, p.H, nq, const_cast<float*>(ga->Wq), nullptr};
            jobs[nj++] = WGJob{o.xc, p.H, o.dqkv[q] + nq, p.P, p.H, nq, const_cast<float*>(ga->Wk), nullptr};
            jobs[nj++] = WGJob{o.xc, p.H, o.dqkv[q] + 2 * nq, p.P, p.H, p.vd, const_cast<float*>(ga->Wv), nullptr};
            jobs[nj++] = WGJob{o.agg[q], p.NV, o.dh0[q] + off, p.in0, p.NV, p.C, const_cast<float*>(ga->Wo), nullptr};
        }
    return nj;
}

static int launch_aggregate_bwd(const BwdPlan& p, const GnfCsr* csr_t, const GnfGnnSpec& gnn, const float* invdeg,
                                const float* dh0s, const float* dh0t, float* g_cond, int64_t ldg, hipStream_t st) {
    const int H = p.H;
    const bool vec4 = (H % 4 == 0) && (p.in0 % 4 == 0) && (ldg % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(dh0s) | reinterpret_cast<uintptr_t>(dh0t) |
                        reinterpret_cast<uintptr_t>(g_cond)) % 16 == 0);
    const int per_row = vec4 ? H / 4 : H;
    int G = 1;
    while (G < per_row && G < 64) G <<= 1;
    const int64_t blocks = (p.n * G + 255) / 256;
    const float* w = gnn.agg == GNF_AGG_MEAN ? invdeg : nullptr;
    const int concat = gnn.combine == GNF_COMBINE_CONCAT ? 1 : 0;
    if (vec4)
        hipLaunchKernelGGL(k_aggregate_bwd<4>, dim3((unsigned)blocks), dim3(256), 0, st, csr_t->rowptr, csr_t->col, w,
                           p.n, dh0s, dh0t, p.in0, H, concat, gnn.epsilon, g_cond, ldg, G);
    else
        hipLaunchKernelGGL(k_aggregate_bwd<1>, dim3((unsigned)blocks), dim3(256), 0, st, csr_t->rowptr, csr_t->col, w,
                           p.n, dh0s, dh0t, p.in0, H, concat, gnn.epsilon, g_cond, ldg, G);
    GNF_LAUNCH_CHECK("k_aggregate_bwd");
    return GNF_OK;
}

// Generic (any layer width) recompute + coupling + dP chain of one half-step out of GEMM building blocks:
// o.h0[q] holds the layer-0 inputs on entry; on exit o.hin / o.dPs / o.gst / o.dh0 are filled, y and g updated.
// snt.LayerNorm backwards (gnn.py:550-552), y = xhat gamma + beta with xhat = (u - mean) rstd over the W features of a row:
//   g_u = rstd (gamma g_y - mean_f(gamma g_y) - xhat mean_f(gamma g_y xhat)),  d gamma = sum_r g_y xhat,  d beta = sum_r g_y
// One wave per row (lanes along the features); g_y [N, W] is overwritten with g_u; a workgroup owns kLnBwdRows rows and
// leaves its share of d gamma / d beta in part[block][net][2][W] (fixed order: per wave over its rows, then wave 0..3).
struct LnBwdJob {
    const float* u;      // un-normalised rows [N, W]
    float* g;            // in: dL/dy, out: dL/du
    const float* gamma;
};
struct LnBwdArgs {
    LnBwdJob job[2];
    float* part;
    int64_t n;
    int32_t W;
};
__global__ __launch_bounds__(256) void k_layer_norm_bwd(const LnBwdArgs a) {
    extern __shared__ float ln_acc[];  // [4 waves][2][W]
    const LnBwdJob j = a.job[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = a.W;
    const float inv_w = 1.f / (float)W;
    float* accg = ln_acc + (size_t)wave * 2 * W;
    float* accb = accg + W;
    for (int f = lane; f < W; f += 64) accg[f] = 0.f, accb[f] = 0.f;
    const int64_t row0 = (int64_t)blockIdx.x * kLnBwdRows;
    for (int rl = wave; rl < kLnBwdRows; rl += 4) {
        const int64_t r = row0 + rl;
        if (r >= a.n) break;
        const float* u = j.u + r * W;
        float* g = j.g + r * W;
        float sum = 0.f;
        for (int f = lane; f < W; f += 64) sum += u[f];
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum * inv_w;
        float sq = 0.f;
        for (int f = lane; f < W; f += 64) {
            const float d = u[f] - mean;
            sq = fmaf(d, d, sq);
        }
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = 1.f / sqrtf(sq * inv_w + GNF_LN_EPS);
        float sa = 0.f, sb = 0.f;
        for (int f = lane; f < W; f += 64) {
            const float xh = (u[f] - mean) * rstd, gy = g[f], gg = j.gamma[f] * gy;
            sa += gg;
            sb = fmaf(gg, xh, sb);
            accg[f] = fmaf(gy, xh, accg[f]);
            accb[f] += gy;
        }
        for (int off = 32; off > 0; off >>= 1) {
            sa += __shfl_xor(sa, off, 64);
            sb += __shfl_xor(sb, off, 64);
        }
        sa *= inv_w, sb *= inv_w;
        for (int f = lane; f < W; f += 64) {
            const float xh = (u[f] - mean) * rstd;
            g[f] = rstd * (j.gamma[f] * g[f] - sa - xh * sb);
        }
    }
    __syncthreads();
    float* out = a.part + ((size_t)blockIdx.x * 2 + blockIdx.y) * 2 * W;
    for (int i = threadIdx.x; i < 2 * W; i += 256)
        out[i] = ((ln_acc[i] + ln_acc[2 * W + i]) + ln_acc[4 * W + i]) + ln_acc[6 * W + i];
}

// d gamma / d beta of both nets: fixed-order sum of the workgroup partials
struct LnGradArgs {
    float* gamma[2];
    float* beta[2];
    const float* part;
    int32_t nblk, W, accumulate;
};
__global__ __launch_bounds__(256) void k_layer_norm_grad(const LnGradArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // [net][2][W]
    if (i >= 4 * a.W) return;
    const int net = i / (2 * a.W), k = (i - net * 2 * a.W) / a.W, f = i - (net * 2 + k) * a.W;
    float s = 0.f;
    for (int b = 0; b < a.nblk; ++b) s += a.part[((size_t)b * 2 + net) * 2 * a.W + (size_t)k * a.W + f];
    float* dst = (k ? a.beta[net] : a.gamma[net]) + f;
    *dst = a.accumulate ? *dst + s : s;
}

// have_rows: o.hin / o.stb already hold every layer input and s, t (the forward's stash, layered_stash_mode): no recompute
static int mlp_backward_generic(const BwdPlan& p, const BwdOperands& o, const GnfGnnSpec& gnn, const GnfMlp* const* nets,
                                const GnfMlp* const* grads, bool acc, const float* x_cond, float* y_upd, int64_t ld,
                                float* g_upd, int64_t ldg, float* ws, hipStream_t st, bool have_rows = false) {
    const int64_t n = p.n;
    const int K = p.K, H = p.H;
    int rc;
    float* const sk[2] = {ws + p.splitk, ws + p.splitk + p.splitk_each};
    for (int j = have_rows ? K : 0; j < K; ++j) {  // recompute the two MLPs, keeping every layer output
        const int I = nets[0]->dims[j], O = nets[0]->dims[j + 1];
        const bool last = j == K - 1;
        GemmJob jobs[2];
        for (int q = 0; q < 2; ++q)
            jobs[q] = GemmJob{o.hin[q * K + j], nets[q]->W[j], last ? o.stb[q] : o.hin[q * K + j + 1], nets[q]->b[j], nullptr};
        GemmShape sh;
        memset(&sh, 0, sizeof(sh));
        sh.lda = o.ldh[j];
        sh.ldb = O;
        sh.ldc = last ? H : p.lmax;
        sh.M = n, sh.K = I, sh.N = O, sh.chunks = 1, sh.kchunk = TGK;
        sh.act = gnn.activation, sh.alpha = gnn.alpha, sh.apply_act = last ? 0 : 1;
        {   // wide layers of packed nets: the large-batch kernel's inner loop, exactly as the forward pass ran them (run_mlps)
            const float* xin[2] = {jobs[0].A, jobs[1].A};
            float* yq[2] = {jobs[0].C, jobs[1].C};
            rc = launch_linear_big(nets, 2, j, xin, sh.lda, yq, sh.ldc, n, gnn.activation, gnn.alpha, last ? 0 : 1, st);
            if (rc == 1) rc = launch_linear_short(nets, 2, j, xin, sh.lda, yq, sh.ldc, n, gnn.activation, gnn.alpha, last ? 0 : 1, st);
            if (rc == GNF_OK) continue;
            if (rc != 1) return rc;
        }
        rc = launch_gemm<OPND_KC, OPND_MC, EPI_BIAS_ACT>(jobs, 2, sh, st, sk, p.splitk_each);
        if (rc) return rc;
    }
    const bool lnorm = nets[0]->attn && nets[0]->attn->layer_norm;
    {
        const bool res = nets[0]->attn && nets[0]->attn->residual;
        const float* sv[2] = {o.stb[0], o.stb[1]};
        if (lnorm) {  // s, t = LayerNorm(MLP(h0) (+ x_cond)): normalised rows to the side, u = MLP(h0) (+ x_cond) kept in place
            float* lny[2] = {ws + p.lny, ws + p.lny + al64((size_t)n * H)};
            LnArgs a;
            memset(&a, 0, sizeof(a));
            for (int q = 0; q < 2; ++q) {
                a.job[q] = LnJob{o.stb[q], lny[q], o.stb[q], nets[q]->attn->ln_gamma, nets[q]->attn->ln_beta};
                sv[q] = lny[q];
            }
            a.ldin = a.ldy = H;
            a.xres = res ? x_cond : nullptr;
            a.ldx = ld;
            a.n = n;
            a.W = H;
            rc = launch_layer_norm(a, 2, st);
            if (rc) return rc;
        }
        int64_t blocks = (n * H + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_coupling_bwd, dim3((unsigned)blocks), dim3(256), 0, st, sv[0], sv[1], y_upd, ld,
                           g_upd, ldg, o.gst[0], o.gst[1], n, H, res && !lnorm ? x_cond : nullptr, ld);
        GNF_LAUNCH_CHECK("k_coupling_bwd");
    }
    if (lnorm) {  // g_s, g_t: dL/d(normalised rows) -> dL/du in place; d gamma, d beta
        const int nblk = (int)((n + kLnBwdRows - 1) / kLnBwdRows);
        LnBwdArgs a;
        memset(&a, 0, sizeof(a));
        for (int q = 0; q < 2; ++q) a.job[q] = LnBwdJob{o.stb[q], o.gst[q], nets[q]->attn->ln_gamma};
        a.part = ws + p.lnpart;
        a.n = n;
        a.W = H;
        hipLaunchKernelGGL(k_layer_norm_bwd, dim3((unsigned)nblk, 2), dim3(256), (size_t)8 * H * sizeof(float), st, a);
        GNF_LAUNCH_CHECK("k_layer_norm_bwd");
        LnGradArgs ga;
        memset(&ga, 0, sizeof(ga));
        for (int q = 0; q < 2; ++q) {
            ga.gamma[q] = const_cast<float*>(grads[q]->attn->ln_gamma);
            ga.beta[q] = const_cast<float*>(grads[q]->attn->ln_beta);
        }
        ga.part = ws + p.lnpart;
        ga.nblk = nblk, ga.W = H, ga.accumulate = acc ? 1 : 0;
        hipLaunchKernelGGL(k_layer_norm_grad, dim3((unsigned)((4 * H + 255) / 256)), dim3(256), 0, st, ga);
        GNF_LAUNCH_CHECK("k_layer_norm_grad");
    }
    for (int j = K - 1; j >= 0; --j) {  // dP_{j-1} = (dP_j W_j^T) * act'(h_j)   [nodes, O] x [O, I]
        const int I = nets[0]->dims[j], O = nets[0]->dims[j + 1];
        GemmJob jobs[2];
        for (int q = 0; q < 2; ++q)
            jobs[q] = GemmJob{o.dPs[q * K + j], nets[q]->W[j], j == 0 ? o.dh0[q] : o.dPs[q * K + j - 1],
                              j == 0 ? nullptr : o.hin[q * K + j], nullptr};
        GemmShape sh;
        memset(&sh, 0, sizeof(sh));
        sh.lda = o.lddp[j];
        sh.ldb = O;
        sh.ldc = j == 0 ? p.in0 : p.lmax;
        sh.ldaux = p.lmax;
        sh.M = n, sh.K = O, sh.N = I, sh.chunks = 1, sh.kchunk = TGK;
        sh.act = gnn.activation, sh.alpha = gnn.alpha;
        {   // dX of a wide layer: k_linear_big on the transposed fragments (the packed copy's W^T half)
            const float* dy[2] = {jobs[0].A, jobs[1].A};
            float* dx[2] = {jobs[0].C, jobs[1].C};
            const float* hm[2] = {jobs[0].aux, jobs[1].aux};
            rc = launch_linear_big_dx(nets, 2, j, dy, sh.lda, dx, sh.ldc, j == 0 ? nullptr : hm, sh.ldaux, n, gnn.activation, gnn.alpha, st);
            if (rc == GNF_OK) continue;
            if (rc != 1) return rc;
        }
        rc = launch_gemm<OPND_KC, OPND_KC, EPI_MASK>(jobs, 2, sh, st, sk, p.splitk_each);
        if (rc) return rc;
    }
    return GNF_OK;
}

}  // namespace gnf

using namespace gnf;

extern "C" {

size_t gnf_backward_workspace_bytes(int64_t n_nodes, int32_t D, const GnfFlow* flow) {
    if (n_nodes < 0 || D < 2 || (D & 1) || !flow || !flow->s_nets) return 0;
    return plan_backward(n_nodes, D, &flow->s_nets[0], kBwdSetsDefault, flow->weight_sharing ? 2 : 2 * flow->num_timesteps).total * sizeof(float);
}

int gnf_grevnet_backward_f32(const GnfCsr* csr, const GnfCsr* csr_t, const GnfFlow* flow, const GnfFlow* grad,
                             float* z, int64_t ld, int32_t D, void* ws, size_t ws_bytes, gnf_stream_t stream,
                             gnf_stream_t aux_stream) {
    int rc = validate_flow_call(csr, flow, ld, D, "gnf_grevnet_backward_f32");
    if (rc) return rc;
    if (!csr_t || csr_t->n_nodes != csr->n_nodes || csr_t->n_edges != csr->n_edges ||
        (csr->n_nodes > 0 && (!csr_t->rowptr || (csr->n_edges > 0 && !csr_t->col)))) {
        set_error("gnf_grevnet_backward_f32: csr_t must be the by-sender CSR of the same batch");
        return GNF_EINVAL;
    }
    if (!grad || !grad->s_nets || !grad->t_nets || grad->num_timesteps != flow->num_timesteps ||
        grad->weight_sharing != flow->weight_sharing) {
        set_error("gnf_grevnet_backward_f32: grad must mirror flow (same T, weight_sharing)");
        return GNF_EINVAL;
    }
    const int T = flow->num_timesteps;
    const int n_nets = flow->weight_sharing ? 2 : 2 * T;
    for (int q = 0; q < n_nets; ++q) {
        const GnfMlp* pairs[2][2] = {{&flow->s_nets[q], &grad->s_nets[q]}, {&flow->t_nets[q], &grad->t_nets[q]}};
        for (auto& pr : pairs) {
            if (pr[0]->attn && (!pr[1]->attn || !pr[1]->attn->Wq || !pr[1]->attn->Wk || !pr[1]->attn->Wv ||
                                !pr[1]->attn->Wo)) {
                set_error("gnf_grevnet_backward_f32: grad net %d needs a GnfAttn with Wq / Wk / Wv / Wo gradient buffers", q);
                return GNF_EINVAL;
            }
            if (pr[0]->attn && pr[0]->attn->layer_norm && (!pr[1]->attn->ln_gamma || !pr[1]->attn->ln_beta)) {
                set_error("gnf_grevnet_backward_f32: grad net %d needs ln_gamma / ln_beta gradient buffers (layer_norm)", q);
                return GNF_EINVAL;
            }
            if (pr[1]->num_layers != pr[0]->num_layers ||
                memcmp(pr[1]->dims, pr[0]->dims, sizeof(int32_t) * (pr[0]->num_layers + 1))) {
                set_error("gnf_grevnet_backward_f32: grad net %d has other layer widths than the flow's", q);
                return GNF_ESHAPE;
            }
            for (int j = 0; j < pr[0]->num_layers; ++j)
                if (!pr[1]->W[j] || !pr[1]->b[j]) {
                    set_error("gnf_grevnet_backward_f32: grad net %d layer %d has null buffers", q, j);
                    return GNF_EINVAL;
                }
        }
    }
    if (flow->bns) {
        if (!grad->bns) {
            set_error("gnf_grevnet_backward_f32: the flow has batch-norm bijectors, grad->bns is NULL");
            return GNF_EINVAL;
        }
        for (int q = 0; q < 2 * T; ++q) {
            rc = validate_bn(&flow->bns[q], GNF_FORWARD, "gnf_grevnet_backward_f32", q);
            if (rc) return rc;
            if (!flow->bns[q].batch_mean || !flow->bns[q].batch_variance || !grad->bns[q].gamma || !grad->bns[q].beta) {
                set_error("gnf_grevnet_backward_f32: batch-norm %d needs the batch moments of the forward pass "
                          "(batch_mean / batch_variance) and gradient buffers (grad->bns[].gamma / beta)", q);
                return GNF_EINVAL;
            }
        }
    }
    const int64_t n = csr->n_nodes;
    hipStream_t st = (hipStream_t)stream;
    if (n_nets == 0) return GNF_OK;
    const int n_nets_each = flow->weight_sharing ? 2 : 2 * flow->num_timesteps;
    const BwdPlan p = plan_backward(n, D, &flow->s_nets[0], kBwdSetsDefault, n_nets_each);
    if (n > 0 && (!z || !ws || ws_bytes < p.total * sizeof(float))) {
        set_error("gnf_grevnet_backward_f32: workspace %zu < %zu bytes (or null z/ws)", ws_bytes,
                  p.total * sizeof(float));
        return GNF_EWORKSPACE;
    }
    const int H = D / 2;
    const size_t stash_slot = attn_stash_slot_floats(flow, n);
    if (flow->attn_stash && stash_slot > 0 && flow->attn_stash_bytes < (size_t)2 * T * stash_slot * sizeof(float)) {
        set_error("gnf_grevnet_backward_f32: attn_stash %zu < %zu bytes", flow->attn_stash_bytes,
                  (size_t)2 * T * stash_slot * sizeof(float));
        return GNF_EWORKSPACE;
    }
    if (n == 0) {  // an empty batch has zero gradient
        for (int q = 0; q < n_nets; ++q)
            for (int kind = 0; kind < 2; ++kind) {
                const GnfMlp* gm = kind ? &grad->t_nets[q] : &grad->s_nets[q];
                for (int j = 0; j < gm->num_layers; ++j) {
                    GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->W[j]), 0, sizeof(float) * gm->dims[j] * gm->dims[j + 1], st));
                    GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->b[j]), 0, sizeof(float) * gm->dims[j + 1], st));
                }
                if (gm->attn) {
                    const GnfAttn* fa = (kind ? &flow->t_nets[q] : &flow->s_nets[q])->attn;
                    const size_t nq = (size_t)fa->num_heads * fa->kq_dim;
                    GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->attn->Wq), 0, sizeof(float) * H * nq, st));
                    GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->attn->Wk), 0, sizeof(float) * H * nq, st));
                    GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->attn->Wv), 0, sizeof(float) * H * fa->v_dim, st));
                    GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->attn->Wo), 0,
                                               sizeof(float) * fa->num_heads * fa->v_dim * fa->out_dim, st));
                    if (fa->layer_norm) {
                        GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->attn->ln_gamma), 0, sizeof(float) * H, st));
                        GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(gm->attn->ln_beta), 0, sizeof(float) * H, st));
                    }
                }
            }
        if (flow->bns)
            for (int q = 0; q < 2 * T; ++q) {
                GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(grad->bns[q].gamma), 0, sizeof(float) * H, st));
                GNF_HIP_TRY(hipMemsetAsync(const_cast<float*>(grad->bns[q].beta), 0, sizeof(float) * H, st));
            }
        return GNF_OK;
    }
    float* wsf = (float*)ws;
    float* g = wsf + p.g;
    {
        int64_t blocks = (n * D + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        const bool mean = flow->gnn.agg == GNF_AGG_MEAN;
        hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)blocks), dim3(256), 0, st, z, ld, g, (int64_t)D, n, D,
                           mean ? csr->rowptr : nullptr, mean ? wsf + p.invdeg : nullptr);
        GNF_LAUNCH_CHECK("k_copy_rows");
    }
    // fork / join events of the weight-gradient stream: flag-only events owned by THIS call (created on the current
    // device, destroyed when the call returns - hipEventDestroy defers the release until the recorded work has
    // completed), so concurrent callers, other devices and stream capture never share one.  ev[0]: "this half-step's
    // operands are ready" (recorded on the main stream, waited for by the auxiliary one); ev[1 + set]: "the dW launch
    // that read operand set `set` is done" - needed before a set is refilled (flows with more half-steps than sets) and,
    // for the last launch, as the final join.
    hipStream_t aux = (hipStream_t)aux_stream;
    if (aux == st) aux = nullptr;
    // small message-passing batches: one launch per half-step carries the backward kernel, the previous half-step's dW
    // GEMMs and the reduce of the one before (k_half_bwd_dw) - no second stream
    bool merged = false;
    int64_t m_tiles = 0;
    size_t m_lds = 0;
    if (p.slab_sets == 2) merged = merged_walk_ok(flow, n, &m_tiles, &m_lds);
    // the training forward left every half-step's MLP rows in GnfFlow.mlp_stash: no recompute (ABI v8)
    const bool mstashed = flow->mlp_stash != nullptr && mlp_stash_supported(flow, n, D / 2);
    const MlpStashLayout msl = mstashed ? mlp_stash_layout(&flow->s_nets[0], n, D / 2) : MlpStashLayout{};
    const bool layered = mstashed && layered_stash_mode(flow, n, D / 2);
    if (mstashed && flow->mlp_stash_bytes < (size_t)2 * T * msl.slot * sizeof(float)) {
        set_error("gnf_grevnet_backward_f32: mlp_stash %zu < %zu bytes", flow->mlp_stash_bytes,
                  (size_t)2 * T * msl.slot * sizeof(float));
        return GNF_EWORKSPACE;
    }
    if (merged) aux = nullptr;
    // attention nets: Wo of every net as transposed fragments, once per call (k_pack_wot) - the tile
    // kernel then forms dagg = dnew Wo^T itself (its last table row) and the GEMM launch in front of the edge kernels
    // goes (7.7 us per half-step on the config-2 batch).
    bool wot_packed = false;
    if (flow->s_nets[0].attn) {
        for (int k0 = 0; k0 < n_nets_each; k0 += 32) {  // (64 pointers per array in the kernel's arguments: 32 nets of each kind a launch)
            const int cnt = n_nets_each - k0 < 32 ? n_nets_each - k0 : 32;
            PackWot pw;
            memset(&pw, 0, sizeof(pw));
            for (int k = 0; k < cnt; ++k) {
                for (int q = 0; q < 2; ++q) {
                    const GnfAttn* at = q ? flow->t_nets[k0 + k].attn : flow->s_nets[k0 + k].attn;
                    const int slot = q * cnt + k, ix = q * n_nets_each + k0 + k;
                    pw.wo[slot] = at->Wo, pw.wq[slot] = at->Wq, pw.wk[slot] = at->Wk, pw.wv[slot] = at->Wv;
                    pw.out[slot] = wsf + p.wot + (size_t)ix * p.wot_each;
                    pw.out_ct[slot] = wsf + p.wct + (size_t)ix * p.wct_each;
                }
            }
            pw.NV = p.NV, pw.C = p.C, pw.NVp = (p.NV + 15) & ~15, pw.Cp = (p.C + 15) & ~15;
            pw.H = p.H, pw.nq = p.nh * p.kq, pw.vd = p.vd, pw.Pp = (p.P + 15) & ~15, pw.Hp = (p.H + 15) & ~15;
            const int pack_elems = pw.NVp * pw.Cp > pw.Pp * pw.Hp ? pw.NVp * pw.Cp : pw.Pp * pw.Hp;
            hipLaunchKernelGGL(k_pack_wot, dim3((unsigned)((pack_elems + 255) / 256), (unsigned)(2 * cnt)), dim3(256), 0, st, pw);
            GNF_LAUNCH_CHECK("k_pack_wot");
        }
        wot_packed = true;
    }
    const bool reuse_sets = 2 * T > p.n_sets;
    // The events live in a per-host-thread, per-device cache (created at a thread's first call on a device, released
    // when the thread ends): no create / destroy per training step, nothing shared between threads or devices, and no
    // event destroyed while work recorded on it is still pending or being captured.
    struct EventCache {
        hipEvent_t ev[64][kBwdMaxSets + 2] = {};
        ~EventCache() {
            for (auto& dev_ev : ev)
                for (hipEvent_t e : dev_ev)
                    if (e) (void)hipEventDestroy(e);
        }
    };
    static thread_local EventCache tl_events;
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    hipEvent_t* const call_ev = tl_events.ev[cur_dev & 63];
    if (aux)
        for (int q = 0; q < kBwdMaxSets + 2; ++q)
            if (!call_ev[q]) GNF_HIP_TRY(hipEventCreateWithFlags(&call_ev[q], hipEventDisableTiming));
    // whatever path leaves this function after the auxiliary stream was given work, the caller's stream is ordered
    // behind it (an early error return used to leave the fork unjoined: fatal inside a stream capture)
    struct AuxJoin {
        hipStream_t aux, st;
        hipEvent_t ev;
        bool armed = false;
        ~AuxJoin() {
            if (armed && aux && ev && hipEventRecord(ev, aux) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
        }
    } aux_join{aux, st, aux ? call_ev[kBwdMaxSets + 1] : nullptr};
    const hipEvent_t ev_ready = call_ev[0];
    hipEvent_t ev_done[kBwdMaxSets] = {};
    hipEvent_t ev_last = nullptr;
    int step = 0;
    bool used[2] = {false, false};  // weight sharing: the T uses of a net accumulate
    // the attention front-end backwards: dagg = dnew Wo^T ([nodes, C] x [C, heads*v]; Wo is [heads*v, C]: rows = output
    // columns), then the edge kernels; dL/dx_cond accumulates into g_cond
    auto attention_backward = [&](const GnfMlp* const* nets_, const BwdOperands& o_, float* g_cond, const AttnBnFold* bnf,
                                  const float* x_cond_, bool have_dagg = false, const float* const* wct_ = nullptr) -> int {
        const GnfAttn* at[2] = {nets_[0]->attn, nets_[1]->attn};
        const int off = at[0]->concat ? D / 2 : 0;
        if (!have_dagg) {  // (else: the backward tile kernel's last table row has written it)
            GemmJob jobs[2];
            for (int q = 0; q < 2; ++q) jobs[q] = GemmJob{o_.dh0[q] + off, at[q]->Wo, o_.dagg[q], nullptr, nullptr};
            GemmShape sh;
            memset(&sh, 0, sizeof(sh));
            sh.lda = p.in0, sh.ldb = p.C, sh.ldc = p.NV;
            sh.M = n, sh.K = p.C, sh.N = p.NV, sh.chunks = 1, sh.kchunk = TGK;
            const int rc_ = launch_gemm<OPND_KC, OPND_KC, EPI_MASK>(jobs, 2, sh, st);
            if (rc_) return rc_;
        }
        return launch_attn_backward(at, n, D / 2, p.in0, csr->rowptr, csr->col, csr_t->rowptr, csr_t->col, o_.qkv, o_.dh0, o_.gst,
                                    o_.dqkv, o_.agg, o_.dagg, o_.stats, g_cond, D, st, csr->n_edges, bnf, x_cond_, ld, o_.xc, wct_);
    };
    int32_t bn_pre = 0;  // batch-norm backward moments left by the attention backward's last kernel (partial rows)
    DwLaunch pend[2];
    bool pend_ok[2] = {false, false};
    bool have_fold = false;  // the previous half-step left its dL/dh0 rows for this one's prologue to scatter
    // the previous half-step's batch-norm bijector, left for this half-step's tile kernel to undo where it reads the
    // half it updates (BwdArgs.bn_part): merged walk, partial sums left by the attention backward's last kernel, no
    // cross-rank moments
    struct PendingBn {
        const GnfBatchNorm* bn = nullptr;
        const GnfBatchNorm* gbn = nullptr;
        int nparts = 0;
        int co = 0;
    } pend_bn;
    auto flush_bn = [&]() -> int {
        if (!pend_bn.bn) return GNF_OK;
        const int rc_ = launch_bn_backward(flow, pend_bn.bn, pend_bn.gbn, z + pend_bn.co, ld, g + pend_bn.co, D, n, H,
                                           reinterpret_cast<double*>(wsf + p.bnpart), st, pend_bn.nparts);
        pend_bn.bn = nullptr;
        return rc_;
    };
    const float* fold_dh[2] = {nullptr, nullptr};
    for (int i = T - 1; i >= 0; --i)
        for (int half = 1; half >= 0; --half) {
            const GnfMlp* nets[2] = {pick_net(flow, flow->s_nets, half, i), pick_net(flow, flow->t_nets, half, i)};
            const GnfMlp* grads[2] = {pick_net(grad, grad->s_nets, half, i), pick_net(grad, grad->t_nets, half, i)};
            const bool acc = flow->weight_sharing && used[half];
            used[half] = true;
            const int co = half == 0 ? 0 : H, uo = half == 0 ? H : 0;
            const bool no_fused = opt(OPT_BWD_GENERIC) != 0;  // (shape forcing for the parity tests)
            const bool attn = nets[0]->attn != nullptr;
            const bool fused = !no_fused && fused_bwd_supported(nets[0], nets[1]);
            const int set = step % p.n_sets;
            BwdOperands o = bwd_operands(p, wsf, set, attn);
            // attention front-end left behind by the forward pass (GnfFlow.attn_stash): q | k | v and the layer-0
            // inputs of both nets are read from the half-step's slot instead of being recomputed
            const bool stashed = attn && stash_slot > 0 && flow->attn_stash != nullptr;
            if (stashed) {
                float* slot = flow->attn_stash + (size_t)(2 * i + half) * stash_slot;
                for (int q = 0; q < 2; ++q) {
                    o.qkv[q] = slot + (size_t)q * n * p.P;
                    o.h0[q] = slot + 2 * (size_t)n * p.P + (size_t)q * n * p.in0;
                    o.hin[q * p.K] = o.h0[q];
                    // attended values (A operand of dWo) and softmax statistics of the forward pass
                    o.agg[q] = slot + 2 * (size_t)n * (p.P + p.in0) + (size_t)q * n * p.NV;
                    o.stats[q] = slot + 2 * (size_t)n * (p.P + p.in0 + p.NV) + (size_t)q * n * 3 * p.nh;
                }
            }
            float* x_cond = z + co;
            // this set's previous reader (the dW GEMMs of n_sets half-steps ago) must be done
            if (aux && reuse_sets && ev_done[set]) GNF_HIP_TRY(hipStreamWaitEvent(st, ev_done[set], 0));
            // ---- layer-0 inputs -------------------------------------------------------------------------
            if (attn) {   // recompute the attention front-end of both nets (q | k | v kept for the way back)
                const GnfAttn* at[2] = {nets[0]->attn, nets[1]->attn};
                if (!stashed) {
                    rc = launch_attn_front(csr->rowptr, csr->col, n, x_cond, ld, H, at, 2, p.in0, o.qkv[0], o.h0, st,
                                           csr->n_edges, true, nullptr, o.agg, o.stats);
                    if (rc) return rc;
                }
                // (the conditioning half as the dW GEMMs of Wq / Wk / Wv read it, o.xc, is copied by the attention
                // backward's last kernel: launch_attn_backward)
            } else if (!fused && !(mstashed && layered)) {   // (layered stash: the forward's layer-0 rows are in the slot)
                rc = launch_aggregate(csr->rowptr, csr->col, n, x_cond, ld, H, flow->gnn.agg == GNF_AGG_MEAN,
                                      flow->gnn.combine == GNF_COMBINE_CONCAT ? 1 : 0, flow->gnn.epsilon, o.h0[0], p.in0, st);
                if (rc) return rc;
            }
            if (merged) {
                BwdArgs ba;
                int mt;
                int64_t tiles;
                size_t lds;
                // the previous half-step's message-passing backward rides in this launch's prologue: it scatters into
                // the gradient of the half this half-step updates (with a batch-norm bijector in between, its backward
                // needs that gradient complete first: the scatter keeps its own launch)
                if (mstashed) {  // layer inputs of both nets come from the half-step's slot of the stash
                    float* slot = flow->mlp_stash + (size_t)(2 * i + half) * msl.slot;
                    for (int q = 0; q < 2; ++q) {
                        if (!attn) {  // (attention nets: the layer-0 inputs are the front-end's, stashed or recomputed above)
                            o.h0[q] = slot + msl.h0;
                            o.hin[q * p.K] = o.h0[q];
                        }
                        for (int j = 1; j < p.K; ++j) o.hin[q * p.K + j] = slot + msl.act + ((size_t)q * (p.K - 1) + (j - 1)) * msl.act_each;
                    }
                }
                const float* h0c[2] = {o.h0[0], o.h0[1]};
                BwdFold bf;
                const bool folded = have_fold;
                if (folded) {
                    bf.rowptr_t = csr_t->rowptr, bf.col_t = csr_t->col;
                    bf.invdeg = flow->gnn.agg == GNF_AGG_MEAN ? wsf + p.invdeg : nullptr;
                    bf.dh_prev[0] = fold_dh[0], bf.dh_prev[1] = fold_dh[1];
                }
                rc = build_bwd_args(csr->rowptr, csr->col, n, flow->gnn, nets[0], nets[1], x_cond, z + uo, ld, g + uo, D, H,
                                    o.h0[0], attn ? h0c : nullptr, o.hin, p.lmax, o.dPs, p.lmax, o.gst, o.dh0, &ba, &mt, &tiles, &lds,
                                    folded ? &bf : nullptr);
                if (rc) return rc;
                if (pend_bn.bn) {  // (pend_bn.co is this half-step's updated half by construction)
                    ba.bn_part = reinterpret_cast<const double*>(wsf + p.bnpart);
                    ba.bn_nparts = pend_bn.nparts;
                    ba.bn_gamma = pend_bn.bn->gamma, ba.bn_beta = pend_bn.bn->beta;
                    ba.bn_mean = pend_bn.bn->batch_mean, ba.bn_var = pend_bn.bn->batch_variance;
                    ba.bn_eps = pend_bn.bn->epsilon;
                    ba.bn_dgamma = const_cast<float*>(pend_bn.gbn->gamma), ba.bn_dbeta = const_cast<float*>(pend_bn.gbn->beta);
                    pend_bn.bn = nullptr;
                }
                bool have_dagg = false;
                const float* wct[2] = {nullptr, nullptr};
                if (attn && wot_packed) {  // dagg = dnew Wo^T as the tile kernel's last row instead of a GEMM launch
                    const int ni = flow->weight_sharing ? half : half * T + i;
                    const float* wot[2] = {wsf + p.wot + (size_t)ni * p.wot_each, wsf + p.wot + (size_t)(n_nets_each + ni) * p.wot_each};
                    have_dagg = bwd_args_add_dagg_row(&ba, wot, o.dagg, p.C, p.NV, nets[0]->attn->concat ? H : 0);
                    wct[0] = wsf + p.wct + (size_t)ni * p.wct_each;
                    wct[1] = wsf + p.wct + (size_t)(n_nets_each + ni) * p.wct_each;
                }
                if (mstashed) {
                    const float* slot = flow->mlp_stash + (size_t)(2 * i + half) * msl.slot;
                    for (int q = 0; q < 2; ++q) ba.st_in[q] = slot + msl.st + (size_t)q * msl.st_each;
                    ba.mask_in = reinterpret_cast<const unsigned long long*>(slot + msl.mask);
                }
                const int prev = (step + 1) & 1, cur = step & 1;   // pend[prev]: half-step k-1, pend[cur]: k-2
                rc = launch_half_bwd_dw(&ba, tiles, lds, step >= 1 && pend_ok[prev] ? &pend[prev] : nullptr,
                                        step >= 2 && pend_ok[cur] ? &pend[cur] : nullptr, mstashed, st);
                if (rc) return rc;
                // this half-step's dW GEMMs ride in the next launch (the last one's get the whole chip)
                WGJob jobs[kMaxGroup];
                const int nj = weight_grad_jobs(p, o, nets, grads, jobs);
                const bool last = step == 2 * T - 1;
                const int cus = big_cu_count();
                int room = last ? cus : cus - (int)tiles;   // CUs the backward tiles of the NEXT launch leave
                if (const int64_t force = opt(OPT_DW_WIDE_UNITS)) room = force < room ? (int)force : room;  // (shape forcing for the parity tests)
                DwPolicy pol{room, lds, 1e30};
                if (mstashed && !last && !opt(OPT_DW_THIN_ON_DW)) pol.tile_wgs = (int)tiles;  // (the launch that carries this plan walks the next half-step)
                rc = plan_weight_grads(p, pol, jobs, nj, acc, wsf, cur, &pend[cur]);
                if (rc) return rc;
                pend_ok[cur] = pend[cur].wide && pend[cur].buf && nj <= kMergedGroup;
                if (!pend_ok[cur]) {  // (a plan the merged launch cannot carry: run it here and now)
                    rc = run_weight_gemms(pend[cur], st);
                    if (rc) return rc;
                    rc = run_weight_reduce(pend[cur], st);
                    if (rc) return rc;
                }
                have_fold = !attn && !flow->bns && !last;
                if (attn) {
                    const GnfBatchNorm* bq = flow->bns ? &flow->bns[half * T + i] : nullptr;
                    const AttnBnFold bnf{z + co, ld, bq ? bq->gamma : nullptr, bq ? bq->beta : nullptr,
                                         reinterpret_cast<double*>(wsf + p.bnpart), &bn_pre};
                    bn_pre = 0;
                    rc = attention_backward(nets, o, g + co, bq ? &bnf : nullptr, x_cond, have_dagg, wct[0] ? wct : nullptr);
                    if (rc) return rc;
                } else if (have_fold) {
                    fold_dh[0] = o.dh0[0], fold_dh[1] = o.dh0[1];
                } else {
                    rc = launch_aggregate_bwd(p, csr_t, flow->gnn, wsf + p.invdeg, o.dh0[0], o.dh0[1], g + co, D, st);
                    if (rc) return rc;
                }
                ++step;
                if (flow->bns) {
                    const bool more = !(i == 0 && half == 0);
                    if (more && attn && bn_pre > 0 && !flow->bn_allreduce && H <= 128 && !folded) {
                        pend_bn.bn = &flow->bns[half * T + i], pend_bn.gbn = &grad->bns[half * T + i];
                        pend_bn.nparts = bn_pre, pend_bn.co = co;
                    } else {
                        rc = launch_bn_backward(flow, &flow->bns[half * T + i], &grad->bns[half * T + i], z + co, ld, g + co, D, n, H,
                                                reinterpret_cast<double*>(wsf + p.bnpart), st, attn ? bn_pre : 0);
                        if (rc) return rc;
                    }
                }
                continue;
            }
            // ---- recompute + coupling + dP chain -------------------------------------------------------------
            bool nm_have_dagg = false;
            const float* nm_wct[2] = {nullptr, nullptr};
            if (attn && wot_packed) {
                const int ni = flow->weight_sharing ? half : half * T + i;
                nm_wct[0] = wsf + p.wct + (size_t)ni * p.wct_each;
                nm_wct[1] = wsf + p.wct + (size_t)(n_nets_each + ni) * p.wct_each;
            }
            if (fused && mstashed) {  // (attention nets, or the auxiliary-stream scheme by option: the stash without the merged launch)
                float* slot = flow->mlp_stash + (size_t)(2 * i + half) * msl.slot;
                for (int q = 0; q < 2; ++q)
                    for (int j = 1; j < p.K; ++j) o.hin[q * p.K + j] = slot + msl.act + ((size_t)q * (p.K - 1) + (j - 1)) * msl.act_each;
                if (!attn) o.h0[0] = o.h0[1] = o.hin[0] = o.hin[p.K] = slot + msl.h0;
                const float* h0c[2] = {o.h0[0], o.h0[1]};
                BwdArgs ba;
                int mt;
                int64_t tiles;
                size_t lds;
                rc = build_bwd_args(csr->rowptr, csr->col, n, flow->gnn, nets[0], nets[1], x_cond, z + uo, ld, g + uo, D, H,
                                    o.h0[0], attn ? h0c : nullptr, o.hin, p.lmax, o.dPs, p.lmax, o.gst, o.dh0, &ba, &mt, &tiles, &lds);
                if (rc) return rc;
                for (int q = 0; q < 2; ++q) ba.st_in[q] = slot + msl.st + (size_t)q * msl.st_each;
                ba.mask_in = reinterpret_cast<const unsigned long long*>(slot + msl.mask);
                if (attn && wot_packed) {  // (as on the merged walk)
                    const int ni = flow->weight_sharing ? half : half * T + i;
                    const float* wot[2] = {wsf + p.wot + (size_t)ni * p.wot_each, wsf + p.wot + (size_t)(n_nets_each + ni) * p.wot_each};
                    nm_have_dagg = bwd_args_add_dagg_row(&ba, wot, o.dagg, p.C, p.NV, nets[0]->attn->concat ? H : 0);
                }
                rc = launch_half_bwd_fused_stashed(ba, mt, tiles, lds, st);
            } else if (fused) {
                const float* h0c[2] = {o.h0[0], o.h0[1]};
                rc = launch_half_bwd_fused(csr->rowptr, csr->col, n, flow->gnn, nets[0], nets[1], x_cond, z + uo, ld,
                                           g + uo, D, H, o.h0[0], attn ? h0c : nullptr, o.hin, p.lmax, o.dPs, p.lmax,
                                           o.gst, o.dh0, st);
            } else {
                const bool rows = mstashed && layered;
                if (rows) {  // the forward's layered path left every layer input and s, t in the half-step's slot
                    float* slot = flow->mlp_stash + (size_t)(2 * i + half) * msl.slot;
                    for (int q = 0; q < 2; ++q) {
                        if (!attn) o.h0[q] = slot + msl.h0, o.hin[q * p.K] = o.h0[q];
                        for (int j = 1; j < p.K; ++j) o.hin[q * p.K + j] = slot + msl.act + ((size_t)q * (p.K - 1) + (j - 1)) * msl.act_each;
                        o.stb[q] = slot + msl.st + (size_t)q * msl.st_each;
                    }
                }
                rc = mlp_backward_generic(p, o, flow->gnn, nets, grads, acc, x_cond, z + uo, ld, g + uo, D, wsf, st, rows);
            }
            if (rc) return rc;
            // message passing: the weight gradients only read what the fused kernel has just written, so their stream
            // forks HERE, before the scatter of dL/dx_cond (with an attention front-end they also read its backward
            // pass and fork after it)
            const bool early_fork = aux && !attn;
            if (early_fork) {
                GNF_HIP_TRY(hipEventRecord(ev_ready, st));
                GNF_HIP_TRY(hipStreamWaitEvent(aux, ev_ready, 0));
            }
            // ---- dL/dx_cond: on the critical path (the next half-step's coupling reads g), so it goes first -----
            if (attn) {
                const GnfBatchNorm* bq = flow->bns ? &flow->bns[half * T + i] : nullptr;
                const AttnBnFold bnf{z + co, ld, bq ? bq->gamma : nullptr, bq ? bq->beta : nullptr,
                                     reinterpret_cast<double*>(wsf + p.bnpart), &bn_pre};
                bn_pre = 0;
                rc = attention_backward(nets, o, g + co, bq ? &bnf : nullptr, x_cond, nm_have_dagg, nm_wct[0] ? nm_wct : nullptr);
            } else {
                rc = launch_aggregate_bwd(p, csr_t, flow->gnn, wsf + p.invdeg, o.dh0[0], o.dh0[1], g + co, D, st);
            }
            if (rc) return rc;
            // ---- weight gradients fork off behind it ------------------------------------------------------------
            {
                hipStream_t wst = st;
                if (aux) {
                    if (!early_fork) {
                        GNF_HIP_TRY(hipEventRecord(ev_ready, st));
                        GNF_HIP_TRY(hipStreamWaitEvent(aux, ev_ready, 0));
                    }
                    wst = aux;
                }
                WGJob jobs[kMaxGroup];
                const int nj = weight_grad_jobs(p, o, nets, grads, jobs);
                int64_t bwd_tiles = 0;
                size_t bwd_lds = 0;
                // (attention nets: the edge kernels of the next half-step want every CU; the grouped kernel's short
                // workgroups share with them better - measured 7.8 vs 8.4 ms per step)
                if (fused && !attn) fused_bwd_launch_shape(nets[0], n, &bwd_tiles, &bwd_lds);
                rc = launch_weight_grads(p, dw_policy(nets[0], bwd_tiles, bwd_lds), jobs, nj, acc, wsf, wst);
                if (rc) return rc;
                if (aux && (reuse_sets || step == 2 * T - 1)) {
                    hipEvent_t e = call_ev[1 + (reuse_sets ? set : 0)];
                    aux_join.armed = true;
                    GNF_HIP_TRY(hipEventRecord(e, aux));
                    ev_done[set] = e;
                    ev_last = e;
                }
            }
            ++step;
            if (rc) return rc;
            if (flow->bns) {  // the bijector sat in front of this half-step (gnn.py:310-313, 325-328)
                rc = launch_bn_backward(flow, &flow->bns[half * T + i], &grad->bns[half * T + i], z + co, ld, g + co, D, n, H,
                                        reinterpret_cast<double*>(wsf + p.bnpart), st, attn ? bn_pre : 0);
                if (rc) return rc;
            }
        }
    rc = flush_bn();  // (never pending here: the last half-step of the walk runs its bijector's own launch)
    if (rc) return rc;
    if (merged) {  // drain the pipeline: dW of the last half-step (+ the reduce before it), then its own reduce
        const int last = (step + 1) & 1, before = step & 1;
        rc = launch_half_bwd_dw(nullptr, 0, m_lds, pend_ok[last] ? &pend[last] : nullptr,
                                step >= 2 && pend_ok[before] ? &pend[before] : nullptr, false, st);
        if (rc) return rc;
        if (pend_ok[last]) {
            rc = run_weight_reduce(pend[last], st);
            if (rc) return rc;
        }
    }
    // join: the auxiliary stream runs its launches in order, so the last one's event covers them all
    if (aux && ev_last) GNF_HIP_TRY(hipStreamWaitEvent(st, ev_last, 0));
    return GNF_OK;
}

}  // extern "C"
