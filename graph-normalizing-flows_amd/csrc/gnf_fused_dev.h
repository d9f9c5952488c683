// Device code shared by the fused forward (gnf_fused.hip) and fused backward (gnf_fused_bwd.hip) half-step
// kernels: the per-wave MLP chunk loop on the exact-fp32 matrix cores with weights streamed from L2 in
// packed fragment order.  See gnf_fused.hip for the design notes.
#pragma once
#include <type_traits>
#include "gnf_common.h"

namespace gnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));


static constexpr int kFusedPF = 2;  // k-groups of weights in flight ahead of the one being multiplied

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- kernel arguments of the fused forward half-step kernels (gnf_fused.hip, gnf_fused_big.hip) ----
struct FusedArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* x_cond;
    float* x_upd;
    const float* x_upd_src;              // old value of the updated half (= x_upd, or the source buffer of an out-of-place first step)
    float* cond_copy;                    // NULL, or where this tile's rows of the conditioning half are copied to (out-of-place first step)
    double* partials;
    double* sq_partials;                 // NETS = 2, forward: also sum(x_upd_new^2) per workgroup (the Gaussian term of the
                                         // flow's last two half-steps, whose outputs are z), or NULL
    double* bn_part;                     // NETS = 2, forward: [tile][H][2] column sums / sums of squares of the updated rows
                                         // (batch moments of the bijector in front of the next half-step), or NULL
    float* st_out[2];                    // NETS = 1: global [N, H] scratch for s (0) and t (1)
    const float* h0[2];                  // precomputed layer-0 input per net ([N, in0], attention GNNs) or NULL
    const float* wp[2][GNF_MAX_LAYERS];  // [net][layer] packed weights
    const float* bias[2];                // [net] contiguous padded bias block (bias_tot floats)
    int32_t ipg[GNF_MAX_LAYERS];         // padded input width / 16 of layer j
    int32_t ont[GNF_MAX_LAYERS];         // padded output width / 16 of layer j
    int32_t boff[GNF_MAX_LAYERS];        // offset of layer j's bias in the LDS bias block
    int64_t ld;
    int32_t n_nodes;
    int32_t n_tiles;
    int32_t H;
    int32_t in0;       // true layer-0 input width (H or 2H)
    int32_t K;
    int32_t LS;        // LDS row stride (floats)
    int32_t bias_tot;  // floats of bias per net in LDS
    int32_t mean, concat, act, inverse;
    int32_t residual;  // attention block with residual: s, t += x_cond (gnn.py:547-548)
    // k_half_big: the launch is a sequence of runs of big_seg_n[k] workgroups that own big_seg_sz[k] row tiles of 16 nodes each
    int32_t big_seg_n[6], big_seg_sz[6];
    // ... and big_seg_kind[k]: 0 = those row tiles and nothing else; 1 / 2 = workgroup w of the run also runs the s-net / the
    // t-net of SPLIT row tile big_xg0 + w (a row tile whose two nets go to two workgroups on different CUs: the launch's time
    // then moves in half row tiles per CU, gnf_fused_big.hip).  The kind-1 workgroup leaves the tile's s rows in
    // big_split_s[w] ([16][hp] floats) and sets big_split_flag[w] = big_epoch; the kind-2 workgroup waits for it and couples.
    int32_t big_seg_kind[6];
    int32_t big_xg0;
    int32_t big_epoch;
    int32_t big_epoch_set;  // what the kind-1 workgroup stores: big_epoch (anything else only under the tests' fault injection)
    float* big_split_s;
    int* big_split_flag;
    float eps, alpha;
    // training forward (STASH instance only): every row the backward pass would otherwise recompute goes to the
    // half-step's slot of GnfFlow.mlp_stash - the layer-0 input, each hidden activation of both nets, s and t
    float* stash_h0;                          // [N, in0]
    float* stash_act[2][GNF_MAX_LAYERS];      // [net][j], j = 1 .. K-1: input of layer j, [N, stash_ld]
    float* stash_st[2];                       // s, t [N, H]
    int32_t stash_w[GNF_MAX_LAYERS];          // true output width of layer j
    int32_t stash_ld;
    unsigned long long* stash_mask;           // [tile][net][K-1][4][mld] ballot of "activation > 0" (act' for the way back)
    int32_t stash_mld;
    // attention instance with the batch-norm bijectors applied on load (FrontArgs.bn_part): (scale, shift) [2][H] of the
    // bijector in front of the PREVIOUS half-step - that half-step read its conditioning rows raw and normalised them on
    // the fly; they are the half THIS half-step rewrites, so the coupling applies the pair to the old value first (NULL: none)
    const float* bnu_const;
    // inverse pass: gamma, beta, moving mean, moving variance of the bijector BEHIND the previous half-step of the walk
    // (bn.forward on its conditioning half, gnn.py:356-358,369-371 = the half this half-step rewrites), applied to the
    // old value on load with k_bn_denorm's own expression (NULL: none)
    const float* bnu_inv[4];
    float bnu_inv_eps;
};


// B fragments come through a buffer descriptor: base = this (net, layer)'s packed weights (SGPRs),
// soffset = wave-uniform byte offset of the 1 KiB fragment block, voffset = lane * 16.  No per-load
// 64-bit VALU address arithmetic and a single constant address VGPR.
#define GNF_LOAD_B(RSRC, VOFF, SOFF) \
    __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RSRC, VOFF, SOFF, 0))

// A wave's unit of work: NV (<= 4) column tiles {nt0, nt0+ts, ...} of one layer.  All wave-uniform.
struct WChunk {
    const float* wbase;  // packed weights of the layer
    unsigned wbytes;
    int ipg;  // k-groups (stages) of the layer
    int nt0;  // first column tile
    int nv;   // tiles in this chunk (1..4)
    int layer;
    int boff;  // offset of the layer's bias inside the LDS bias block
    int ont;   // column tiles of the layer
};

static constexpr int kPF = kFusedPF;

// Issue the loads of the first kPF stages of chunk c into b_pre (4 tile slots; slots >= c.nv repeat the
// last valid tile).  Called one round BEFORE the previous chunk ends - and before the prologue for
// the first chunk - so that a layer never starts by waiting a full L2 round trip for its weights.
// A THIN chunk (one column tile, at most kThinStages k-groups: the 256 -> 32 output layer of the reference's MLPs) has
// only 4 MFMAs per stage to hide a weight load behind; its prefetch uses the four tile slots for four consecutive
// STAGES instead of repeating the one tile (stages 0 .. 4 kPF - 1), and mlp_chunk_thin requests the remaining stages at
// once on entry: one exposed round trip per layer instead of one per stage.
static constexpr int kThinStages = 16;
__device__ __forceinline__ bool chunk_is_thin(const WChunk& c) { return c.nv == 1 && c.ipg <= kThinStages; }

__device__ __forceinline__ void prefetch_chunk(const WChunk& c, int ts, int voff, f32x4 (&b_pre)[kPF][4],
                                               bool thin_ok = false) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.wbase), 0, (int)c.wbytes, 0x00020000);
    if (thin_ok && chunk_is_thin(c)) {  // wave-uniform
#pragma unroll
        for (int u = 0; u < kPF; ++u)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int s = 4 * u + b, kn = s < c.ipg ? s : c.ipg - 1;
                b_pre[u][b] = GNF_LOAD_B(rsrc, voff, (kn * c.ont + c.nt0) * 1024);
            }
        return;
    }
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
        const int kn = u < c.ipg ? u : c.ipg - 1;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int tb = b < c.nv ? b : c.nv - 1;
            b_pre[u][b] = GNF_LOAD_B(rsrc, voff, (kn * c.ont + c.nt0 + ts * tb) * 1024);
        }
    }
}

template <int N, class F>
__device__ __forceinline__ void epi_static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
    if constexpr (N > 0) {
        epi_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// What happens to a chunk's outputs besides the write into the LDS activation buffer:
//   EPI_PLAIN  nothing (forward kernel)
//   EPI_EX     backward kernel (ea.mode is wave-uniform):
//     mode 0   forward (recompute) layer: activation as usual, its sign (= act' of this layer) goes into an
//              LDS byte mask, the value also to global memory (input of the dW GEMM)
//     mode 1   backward layer: multiply by act' read from the byte mask, store to LDS and global memory
enum { EPI_PLAIN = 0, EPI_EX = 1 };
struct EpiArgs {
    int mode;
    float* dump;               // global [n_nodes, dld] or NULL
    int64_t dld;
    int width;                 // true (unpadded) output width of the layer
    int row0, n_nodes;
    // act' as one bit per element: for every (M-tile m, accumulator row r, 16-column tile) the 64-lane ballot of
    // "activation > 0" in the accumulator layout (lane = 16 * (row >> 2 & 3) + column).  The backward layer that
    // produces dL/dh_j has exactly the layout the forward layer that produced h_j had, so it tests its own lane's bit.
    unsigned long long* mask;  // LDS [MT * 4][mld] words, written in mode 0, read in mode 1 (NULL: no mask)
    int mld;                   // 16-column tiles per mask row
    float act_slope;           // act' on the negative side (alpha, or 0 for relu)
};

// One-tile chunk with every stage's operands in flight before the first MFMA (see chunk_is_thin).  Same arithmetic and
// the same k order as mlp_chunk<1, 1>: bitwise the same result.  b_pre holds stages 0 .. 4 kPF - 1 (thin packing).
__device__ __forceinline__ void mlp_chunk_thin(const float* __restrict__ in_lds, int LS, const WChunk& c, const WChunk& nx,
                                               int ts, const float* __restrict__ bias_lds, float* __restrict__ out_lds,
                                               float slope, int lane, f32x4 (&b_pre)[kPF][4]) {
    constexpr int NS = kThinStages, NP = 4 * kPF;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int ipg = c.ipg;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.wbase), 0, (int)c.wbytes, 0x00020000);
    const int voff = lane * 16;
    f32x4 bst[NS], ast[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s < NP)
            bst[s] = b_pre[s / 4][s % 4];
        else
            bst[s] = GNF_LOAD_B(rsrc, voff, ((s < ipg ? s : ipg - 1) * c.ont + c.nt0) * 1024);
    }
    const float* arow = in_lds + lrow * LS + 4 * lgrp;
#pragma unroll
    for (int s = 0; s < NS; ++s) ast[s] = *reinterpret_cast<const f32x4*>(arow + 16 * (s < ipg ? s : ipg - 1));
    const float bias = bias_lds[16 * c.nt0 + lrow];
    f32x4 acc = {bias, bias, bias, bias};
    prefetch_chunk(nx, ts, voff, b_pre, true);  // (b_pre was copied above)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < ipg) {  // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ast[s][q], bst[s][q], acc, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = acc[r];
        out_lds[(4 * lgrp + r) * LS + 16 * c.nt0 + lrow] = fmaxf(v, slope * v);
    }
}

template <int MT, int NV, int EPI = EPI_PLAIN>
__device__ __forceinline__ void mlp_chunk(const float* __restrict__ in_lds, int LS, const WChunk& c,
                                          const WChunk& nx, int ts, const float* __restrict__ bias_lds,
                                          float* __restrict__ out_lds, float slope, int lane,
                                          f32x4 (&b_pre)[kPF][4], const EpiArgs& ea = EpiArgs{}, bool thin_ok = false) {
    constexpr int PF = kPF;
    constexpr int R = PF + 1;  // register ring: PF stages in flight + the one being consumed
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int ipg = c.ipg, nt0 = c.nt0;
    f32x4 acc[MT][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) {
        const float bias = bias_lds[16 * (nt0 + ts * b) + lrow];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][b] = f32x4{bias, bias, bias, bias};
    }
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.wbase), 0, (int)c.wbytes, 0x00020000);
    const int voff = lane * 16;
    int wtile[NV];  // byte offset of each tile's fragment block inside a k-group (1 KiB per tile)
#pragma unroll
    for (int b = 0; b < NV; ++b) wtile[b] = (nt0 + ts * b) * 1024;
    const int kstride = c.ont * 1024;  // bytes between consecutive k-groups
    const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) arow[m] = in_lds + (16 * m + lrow) * LS + 4 * lgrp;

    f32x4 a_ring[R][MT], b_ring[R][NV];
    // stage k always lives in ring slot k % R; the B side of the first PF stages was prefetched
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int kn = u < ipg ? u : ipg - 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) a_ring[u][m] = *reinterpret_cast<const f32x4*>(arow[m] + 16 * kn);
#pragma unroll
        for (int b = 0; b < NV; ++b) b_ring[u][b] = b_pre[u][b];
    }
#define GNF_MFMA_STAGE(U)                                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int b = 0; b < NV; ++b)   \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) acc[m][b] =                                 \
            __builtin_amdgcn_mfma_f32_16x16x4f32(a_ring[U][m][q], b_ring[U][b][q], acc[m][b], 0, 0, 0);
    // One round = R stages; a stage = the 4*NV*MT MFMAs of k-group kg with the loads of k-group kg+PF
    // (NV buffer loads, MT LDS reads) issued in their shadow: an MFMA occupies the matrix pipe for 32
    // cycles but the wave's issue slot for ~4, so a load placed after an MFMA costs nothing, while
    // loads bunched in front of the MFMAs leave the pipe idle whenever the partner wave on the SIMD
    // has nothing to issue (the arbiter is oldest-first: the two waves run mostly one after the
    // other, not interleaved).  sched_group_barrier spells the interleave, sched_barrier(0) closes
    // the stage (left alone, hipcc sinks every load of a round to its end and waits vmcnt(0) at the
    // top of the next one); there is NO branch inside a round (it would also force vmcnt(0)).
#define GNF_INTERLEAVE()                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* 1 MFMA */                               \
    __builtin_amdgcn_sched_group_barrier(0x100, MT, 0); /* the LDS reads of the next A fragments */ \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT - 1, 0);                                    \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  /* 1 weight load */                        \
    _Pragma("unroll") for (int b_ = 1; b_ < NV; ++b_) {                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                         \
    }
#define GNF_ROUND(KG0)                                                                             \
    _Pragma("unroll") for (int u = 0; u < R; ++u) {                                                \
        const int kg = (KG0) + u;                                                                  \
        const int kn = (kg + PF < ipg) ? kg + PF : ipg - 1; /* tail re-reads the last stage */     \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) a_ring[(u + PF) % R][m] =                   \
            *reinterpret_cast<const f32x4*>(arow[m] + 16 * kn);                                    \
        _Pragma("unroll") for (int b = 0; b < NV; ++b) b_ring[(u + PF) % R][b] =                   \
            GNF_LOAD_B(rsrc, voff, wtile[b] + kn * kstride);                                       \
        GNF_MFMA_STAGE(u)                                                                          \
        GNF_INTERLEAVE()                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    int kg0 = 0;
    if (ipg >= R) {
        for (; kg0 + 2 * R <= ipg; kg0 += R) {  // every full round but the last
            GNF_ROUND(kg0)
        }
        // the next chunk's first stages go out one round early: they land while the last round's
        // MFMAs, the output write-back and the layer barrier are in progress
        prefetch_chunk(nx, ts, voff, b_pre, thin_ok);
        __builtin_amdgcn_sched_barrier(0);
        GNF_ROUND(kg0)
        kg0 += R;
    } else {
        prefetch_chunk(nx, ts, voff, b_pre, thin_ok);
        __builtin_amdgcn_sched_barrier(0);
    }
    // tail: the last ipg % R stages are already in flight in slots 0 .. rem-1
    const int rem = ipg - kg0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        if (u < rem) {  // wave-uniform
            GNF_MFMA_STAGE(u)
        }
    }
#undef GNF_ROUND
#undef GNF_INTERLEAVE
#undef GNF_MFMA_STAGE
    // accumulator layout: col = lane&15, row = 4*(lane>>4) + r.  slope: 1 on the last (linear) layer,
    // alpha (leaky) or 0 (relu) otherwise: max(v, slope*v) is branch-free for all three.
    // EPI_EX: the chunk's MT * NV * 4 mask words travel through ONE register pair - lane e holds the word of accumulator
    // element e = 4 (NV m + b) + r.  Mode 0 drops each ballot into its lane (v_writelane) and the first lanes store their
    // words at the end; mode 1 has those lanes fetch the words up front and every element reads its word back as a scalar
    // pair, which IS the lane mask of the select.  (One exec-masked 8-byte store, or one broadcast read + 64-bit shift,
    // per element cost the training forward 0.85 us per hidden layer: stamps, CHANGELOG round 6.)
    [[maybe_unused]] int mlo = 0, mhi = 0;
    [[maybe_unused]] int my_word = 0;
    if constexpr (EPI == EPI_EX) {
        const int e = lane < MT * NV * 4 ? lane : 0;
        const int em = e / (4 * NV), eb = (e >> 2) % NV, er = e & 3;
        my_word = (4 * em + er) * ea.mld + nt0 + ts * eb;
        if (ea.mode != 0) {
            unsigned long long w = ~0ull;
            if (ea.mask != nullptr && lane < MT * NV * 4) w = ea.mask[my_word];
            mlo = (int)(unsigned)w, mhi = (int)(unsigned)(w >> 32);
        }
    }
    if constexpr (EPI == EPI_PLAIN) {  // (the inference kernels' epilogue, as it has always been)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int b = 0; b < NV; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[m][b][r];
                    const int rl = 16 * m + 4 * lgrp + r, col = 16 * (nt0 + ts * b) + lrow;
                    out_lds[rl * LS + col] = fmaxf(v, slope * v);
                }
    } else
    epi_static_for<MT * NV * 4>([&](auto e_c) {
                constexpr int e = decltype(e_c)::value;
                constexpr int m = e / (4 * NV), b = (e >> 2) % NV, r = e & 3;
                const float v = acc[m][b][r];
                const int rl = 16 * m + 4 * lgrp + r, col = 16 * (nt0 + ts * b) + lrow;
                {
                    float o;
                    if (ea.mode == 0) {
                        o = fmaxf(v, slope * v);
                        const unsigned long long bal = __ballot(o > 0.f);
                        // (gfx940+: a VALU that reads an SGPR another VALU has just written needs two wait states; the compiler
                        // inserts them for its own instructions, not in front of inline assembly)
                        asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                                     : "+v"(mlo), "+v"(mhi)
                                     : "s"((int)(unsigned)bal), "s"((int)(unsigned)(bal >> 32)), "n"(e));
                    } else {
                        const unsigned long long keep = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, e) << 32) |
                                                        (unsigned)__builtin_amdgcn_readlane(mlo, e);
                        const float vs = v * ea.act_slope;
                        asm volatile("s_nop 1\n\tv_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(o) : "v"(vs), "v"(v), "s"(keep));
                    }
                    out_lds[rl * LS + col] = o;
                    if (ea.dump && ea.row0 + rl < ea.n_nodes && col < ea.width)
                        ea.dump[(int64_t)(ea.row0 + rl) * ea.dld + col] = o;
                }
            });
    if constexpr (EPI == EPI_EX) {
        if (ea.mode == 0 && ea.mask != nullptr && lane < MT * NV * 4)
            ea.mask[my_word] = ((unsigned long long)(unsigned)mhi << 32) | (unsigned)mlo;
    }
}



// A tile's rows of one LDS activation buffer ([TM][LS], first `width` columns) to global memory [n_nodes, dld]: coalesced
// 16-byte-per-lane copy where the widths allow (element-wise stores from the accumulator layout, 64-byte segments, made
// the backward kernel store-bound on large batches).  All NTHR threads take part.
// NT: non-temporal stores - rows nobody reads before the launch ends (dW operands, the stash): left as ordinary dirty
// lines they sit in the XCD L2s until the kernel-end write-back and the launch ends that much later (23 MB per
// launch: 6 us of the training forward).
template <int TM, int NTHR, bool NT = false>
__device__ __forceinline__ void tile_dump(const float* __restrict__ src, int LS, float* __restrict__ dump, int64_t dld,
                                          int width, int row0, int n_nodes, int tid) {
    if (((width | (int)dld) & 3) == 0 && (reinterpret_cast<uintptr_t>(dump) & 15) == 0) {
        // thread = (row, lane of the row): NTHR / TM lanes walk a row 16 bytes each - no division by the run-time width and
        // ONE 64-bit row address per thread (a flat index i -> (i / w4, i % w4) per element cost ~40 instructions each)
        static_assert(NTHR % TM == 0, "whole threads per row");
        constexpr int TPR = NTHR / TM;
        const int w4 = width >> 2;
        const int rl = tid / TPR, c0 = tid - rl * TPR;
        const int r = row0 + rl;
        if (r < n_nodes) {
            const float* sp = src + rl * LS;
            float* dp = dump + (int64_t)r * dld;
            for (int c4 = c0; c4 < w4; c4 += TPR) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sp + 4 * c4);
                f32x4* d = reinterpret_cast<f32x4*>(dp + 4 * c4);
                if (NT)
                    __builtin_nontemporal_store(v, d);
                else
                    *d = v;
            }
        }
    } else {
        for (int i = tid; i < TM * width; i += NTHR) {
            const int rl = i / width, c = i - rl * width;
            const int r = row0 + rl;
            if (r < n_nodes) {
                if (NT)
                    __builtin_nontemporal_store(src[rl * LS + c], dump + (int64_t)r * dld + c);
                else
                    dump[(int64_t)r * dld + c] = src[rl * LS + c];
            }
        }
    }
}

// ---- A: aggregate + combine of one node tile into the layer-0 input (shared by the forward and backward
// kernels; same arithmetic and order as gnn.py:103-104,117-118,123 | 108-109) ----------------------------
// The tile's CSR slice is staged in LDS first (the caller has put rowptr[row0 .. row0+TM] into s_rowptr;
// the contiguous col segment follows here, coalesced), so that the neighbour-row gathers are independent
// loads issued 8 at a time instead of a rowptr -> col -> x chain of dependent global round trips per
// thread.  Writes buf0 (and buf1 when non-NULL) [TM][LS] and, when h0_out != NULL, the true [n, in0] rows.
struct TileAgg {
    const int32_t* col;
    const float* x_cond;
    int64_t ld;
    int n_nodes, row0, H, in0, in0p, mean, concat;
    float eps;
    float* cond_copy = nullptr;  // NULL, or [n, H] destination (leading dimension ld) for the tile's own x_cond rows
};

// The gather itself.  s_rowptr[0 .. TM] and - when the segment is short enough (seg_len <= COLCAP) - s_col[0 .. seg_len)
// must already be in LDS and visible (barrier behind the staging stores).
template <int TM, int NTHR, int COLCAP>
__device__ __forceinline__ void tile_gather(const TileAgg& t, const int* __restrict__ s_rowptr, const int* __restrict__ s_col,
                                            float* __restrict__ buf0, float* __restrict__ buf1, int LS,
                                            float* __restrict__ h0_out, int tid) {
    const int seg_beg = s_rowptr[0];
    const int seg_len = s_rowptr[TM] - seg_beg;
    const bool staged = seg_len <= COLCAP;  // workgroup-uniform
    // sum of x_cond[nbr, f] over the incoming edges [beg, end) of one node, in edge order
    auto gather = [&](int beg, int end, const float* xf, auto colat) -> float {
        float s = 0.f;
        int e = beg;
        for (; e + 8 <= end; e += 8) {  // 8 independent row reads in flight
            int ci[8];
            float vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ci[q] = colat(e + q);
#pragma unroll
            for (int q = 0; q < 8; ++q) vv[q] = xf[(int64_t)ci[q] * t.ld];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += vv[q];
        }
        if (e < end) {  // up to 7 left: still issued together (indices clamped, adds predicated)
            int ci[7];
            float vv[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) ci[q] = colat(e + q < end ? e + q : end - 1);
#pragma unroll
            for (int q = 0; q < 7; ++q) vv[q] = xf[(int64_t)ci[q] * t.ld];
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (e + q < end) s += vv[q];
        }
        return s;
    };
    const int H = t.H, in0p = t.in0p;
    for (int idx = tid; idx < TM * in0p; idx += NTHR) {
        const int rl = idx / in0p, c = idx - rl * in0p;
        const int r = t.row0 + rl;
        float v = 0.f;
        if (r < t.n_nodes && c < t.in0) {
            const int f = c < H ? c : c - H;
            if (t.concat && c < H) {
                v = t.x_cond[(int64_t)r * t.ld + f];
                if (t.cond_copy) t.cond_copy[(int64_t)r * t.ld + f] = v;
            } else {
                const int beg = s_rowptr[rl], end = s_rowptr[rl + 1];
                const float* xf = t.x_cond + f;
                float s = staged ? gather(beg, end, xf, [&](int e) { return s_col[e - seg_beg]; })
                                 : gather(beg, end, xf, [&](int e) { return t.col[e]; });
                if (t.mean) {
                    const int cnt = end - beg;
                    s = s / (float)(cnt > 1 ? cnt : 1);
                }
                if (t.concat) {
                    v = s;
                } else {
                    const float xc = t.x_cond[(int64_t)r * t.ld + f];
                    if (t.cond_copy) t.cond_copy[(int64_t)r * t.ld + f] = xc;
                    v = t.eps * xc + s;
                }
            }
            if (h0_out) h0_out[(int64_t)r * t.in0 + c] = v;
        }
        buf0[rl * LS + c] = v;
        if (buf1) buf1[rl * LS + c] = v;
    }
}

// staging of the col segment (after the caller has put rowptr[row0 .. row0+TM] into s_rowptr and synchronised) + gather
template <int TM, int NTHR, int COLCAP>
__device__ __forceinline__ void tile_aggregate(const TileAgg& t, const int* __restrict__ s_rowptr, int* __restrict__ s_col,
                                               float* __restrict__ buf0, float* __restrict__ buf1, int LS,
                                               float* __restrict__ h0_out, int tid) {
    const int seg_beg = s_rowptr[0];
    const int seg_len = s_rowptr[TM] - seg_beg;
    if (seg_len <= COLCAP)
        for (int i = tid; i < seg_len; i += NTHR) s_col[i] = t.col[seg_beg + i];
    __syncthreads();
    tile_gather<TM, NTHR, COLCAP>(t, s_rowptr, s_col, buf0, buf1, LS, h0_out, tid);
}

}  // namespace gnf
