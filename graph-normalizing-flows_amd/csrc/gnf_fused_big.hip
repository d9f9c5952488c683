// Fused coupling half-step, large-batch form (gfx950 / MI355X): the kernel for batches with several node tiles per CU.
//
// Same arithmetic as k_half_fused (gnf_fused.hip; gnn.py:103-126, 159-180, 320-338 | 356-372): same packed weights, same
// MFMA fragment mapping and k order, so s, t and the updated rows are BITWISE what the 16- / 32-row shapes produce.  What
// changes is how the work is laid out, to answer what bounds those shapes once the chip is full (every
// workgroup re-streams both nets' packed weights from L2 for its own 16 or 32 rows; one 133 KB workgroup per CU leaves
// nothing to run in the shadow of its prologue, layer barriers, thin layers and epilogue; and a launch's time moves in
// quanta of "one tile per CU"):
//
//   * a workgroup is 4 waves (one per SIMD) and owns 1 .. 4 row tiles of 16 nodes; it runs the s-net, then the t-net,
//     over them.  In a 256-wide layer a wave owns column tiles {w, w+4, w+8, w+12} x every row tile: 16 accumulators,
//     4 weight loads + 4 LDS reads per 64 MFMAs - the weight stream per row is half the 32-row both-nets shape's;
//   * activations live in ONE [64][260] LDS buffer, updated in place: a wave's column tiles of a layer fit its
//     accumulators, so a layer is "all waves read - barrier - all waves write - barrier" (layers wider than 256 would
//     need a second pass per wave: such nets stay on k_half_fused).  66.7 KB per workgroup = TWO workgroups per CU, each
//     filling the other's barriers / prologue / thin layers / epilogue with its MFMAs; 2 waves per SIMD = 256 VGPRs each,
//     no spills (a first 8-wave / 128-register form spilled 70-90 registers per layer: 880 MB of scratch traffic per launch);
//   * how many row tiles a workgroup gets is a RUN-TIME value (big_plan): a batch of up to one pass of the chip is dealt
//     out evenly over 2 x CUs workgroups so that the launch ends everywhere at once instead of rounding up to whole
//     64-row tiles per CU; larger batches run whole double rounds of 4-tile workgroups and a closing round dealt out the
//     same way.  The MFMAs of a k-group run row tile by row
//     tile; 4- and 3-tile workgroups have an instance each without row-tile branches, workgroups of 1 or 2 row tiles
//     take one with a twice-as-deep weight ring in the same registers;
//   * s stays in the accumulator registers while the t-net runs; then s | t go side by side into the (now free)
//     activation buffer and ONE compact loop does the coupling update x*exp(s)+t | (x-t)*exp(-s) with 16-byte row
//     accesses and the fp64 partials of sum(s), sum(x_new^2);
//   * the layer-0 input rows (eps*x + agg | [x || agg]) come from the standalone aggregation kernel (k_aggregate,
//     launched in front by launch_half_fused): thousands of light waves hide the neighbour-row latency there, two
//     256-register waves per SIMD cannot (an in-kernel gather cost a 64-row workgroup 20 k cycles at H = 32 and
//     70-170 k at H = 128).  Attention GNNs hand their per-net layer-0 rows over the same way;
//   * thin layers (1 or 2 column tiles: the 256 -> 32 output layer) split the ROW tiles over the waves so that all four
//     SIMDs work on them, with every k-group's weights requested up front;
//   * the kernel's code is kept small on purpose (loops that run once per tile are NOT unrolled, one copy of expf):
//     two CUs share one 64 KB instruction cache, and a 33 KB unrolled gather once made everything 2x slower.
#include "gnf_common.h"
#include "gnf_fused_dev.h"

namespace gnf {

static constexpr int kBigThreads = 256;  // 4 waves: one per SIMD and workgroup, two workgroups per CU
static constexpr int kBigWaves = 4;
static constexpr int kBigMT = 4;         // row tiles (of 16 nodes) a workgroup holds at most
static constexpr int kBigTM = 16 * kBigMT;
static constexpr int kBigLS = 260;       // LDS row stride (floats): 256 + 4, rows 16-byte aligned and spread over banks
static constexpr int kBigMaxW = 256;     // widest padded layer input / hidden width
static constexpr int kBigMaxH = 128;     // widest padded output (s stays in 8 * 4 accumulator registers of its wave)
static constexpr int kBigRBW = 2;            // weight k-groups in the register ring of the widest shapes (one less in flight; 3 measured equal)
static constexpr int kBigRB2 = 2 * kBigRBW;  // the same for workgroups of 1 or 2 row tiles (their k-groups are half as long)

static inline int pad16(int v) { return (v + 15) & ~15; }


// a wave's share of one layer (all wave-uniform)
struct BChunk {
    const float* wbase;  // packed weights of the layer
    const float* bias;   // this layer's padded bias row (LDS)
    unsigned wbytes;
    int ipg;   // k-groups
    int ont;   // column tiles of the layer
    int col0;  // first column tile (the others: col0 + 4, + 8, + 12)
    int nv;    // column tiles of this wave: 0 .. 4
    int m0;    // first row tile
    int mw;    // row tiles of this wave: 0 .. 4
    int thin;  // the layer has 1 or 2 column tiles: the row tiles are split over the waves (big_chunk_thin)
    int active;
};

#define GNF_BIG_LOAD_B(RSRC, VOFF, SOFF) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RSRC, VOFF, SOFF, 0))

// stage 0 of a chunk's weight stream: requested while the previous layer is still multiplying
__device__ __forceinline__ void big_prefetch(const BChunk& c, int lane, f32x4 (&b_nx)[4]) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.wbase), 0, (int)c.wbytes, 0x00020000);
    const int voff = lane * 16;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int ct = c.col0 + (b < c.nv ? kBigWaves * b : 0);
        b_nx[b] = GNF_BIG_LOAD_B(rsrc, voff, ct * 1024);
    }
}

// Accumulator layout.  The MFMAs take the WEIGHT fragment as their first operand and the activation fragment as the
// second (k_half_fused has them the other way round): the product is the transposed tile, lane l holds
// out[row = l & 15][16 ct + 4 (l >> 4) + r], r = 0 .. 3 - four consecutive COLUMNS of one row, so a layer's write-back is
// one 16-byte LDS store per (row tile, column tile) instead of four 4-byte ones.  Same registers, same four products per
// MFMA in the same k order: bitwise the same sums (tests/test_big_shape_gpu.py compares with the 16-row shape).
//
// What happens to a chunk's accumulators (row tiles m < c.mw): a hidden layer ends with "barrier (every wave has read the
// layer's input) - activation, write in place - barrier"; the s-net's last layer leaves s in registers; the t-net's last
// layer puts s (columns [0, hp)) and t (columns [hp, 2 hp)) of the tile's rows into the activation buffer, behind a
// barrier (every wave has read the last hidden rows) - the kernel's coupling loop takes them from there.
template <int NV, int MW>
__device__ __forceinline__ void big_layer_end(float* __restrict__ act, const BChunk& c, int lane, const f32x4 (&acc)[MW][NV],
                                              bool last, float slope, f32x4 (&s_keep)[kBigMT][2], bool couple, int hp) {
    const int lrow = lane & 15, lgrp = lane >> 4;
    float* const base = act + (16 * c.m0 + lrow) * kBigLS + 16 * c.col0 + 4 * lgrp;
    if (!last) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MW; ++m)
            if (m < c.mw) {  // wave-uniform
#pragma unroll
                for (int b = 0; b < NV; ++b) {
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc[m][b][r], slope * acc[m][b][r]);
                    *reinterpret_cast<f32x4*>(base + 16 * m * kBigLS + 16 * kBigWaves * b) = o;
                }
            }
        __syncthreads();
    } else if constexpr (NV <= 2) {  // (the coupling half has at most 8 column tiles: two per wave)
        if (!couple) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int b = 0; b < NV; ++b) s_keep[m][b] = acc[m][b];
        } else {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MW; ++m)
                if (m < c.mw) {
#pragma unroll
                    for (int b = 0; b < NV; ++b) {
                        float* p = base + 16 * m * kBigLS + 16 * kBigWaves * b;
                        *reinterpret_cast<f32x4*>(p) = s_keep[m][b];
                        *reinterpret_cast<f32x4*>(p + hp) = acc[m][b];
                    }
                }
        }
    }
}

// One layer's share of a wave in a layer with 3 or more column tiles:
//     acc[m][b] = bias + sum_k act[16 m .. +16][k] * W[k][16 (col0 + 4 b) .. +16],   m < c.mw (run time), b < NV.
// B ring: RB slots (RB - 1 k-groups in flight).  A fragments: ONE slot per row tile, refilled in place - the MFMAs of a
// k-group run row tile by row tile and a[m]'s next k-group is requested right behind the last MFMA that reads the current
// one, most of a stage before it is needed.  An accumulator sees its k-groups and the four MFMAs inside a k-group in
// k_half_fused's order: bitwise the same sums.  The issue order is pinned the same way (sched_group_barrier): loads sit
// behind MFMAs, never bunched in front of them.  Row tiles >= c.mw are skipped (wave-uniform branches between the row
// tiles' MFMA blocks).
// MW = 4 | 3 | 2: the row tiles the instance holds accumulators for.  A workgroup of 1 or 2 row tiles takes the MW = 2 instance,
// whose B ring is twice as deep in the same registers: its k-groups are half as long, and the weight fragments have to be
// requested the same TIME ahead (a lone 1-tile workgroup with one k-group in flight ran 925 cycles per 512-cycle k-group:
// the L2 round trip; a 2-tile workgroup beside a 4-tile one took as long as its partner).
// FULL: every row tile of the instance is live (c.mw == MW) - no branches between the row tiles' MFMA blocks.
template <int NV, int RB, int MW, bool FULL>
__device__ __forceinline__ void big_chunk(float* __restrict__ act, const BChunk& c, const BChunk& nx, bool have_nx,
                                          int lane, f32x4 (&b_nx)[4], bool last, float slope,
                                          f32x4 (&s_keep)[kBigMT][2], bool couple, int hp) {
    constexpr int PF = RB - 1;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int ipg = c.ipg, mw = FULL ? MW : c.mw;
    f32x4 acc[MW][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b)
#pragma unroll
        for (int m = 0; m < MW; ++m) acc[m][b] = *reinterpret_cast<const f32x4*>(c.bias + 16 * (c.col0 + kBigWaves * b) + 4 * lgrp);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.wbase), 0, (int)c.wbytes, 0x00020000);
    const int voff = lane * 16;
    int wtile[NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) wtile[b] = (c.col0 + kBigWaves * b) * 1024;
    const int kstride = c.ont * 1024;
    const float* arow = act + lrow * kBigLS + 4 * lgrp;  // (m0 = 0: every row tile of the workgroup)

    f32x4 a_frag[MW], b_ring[RB][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) b_ring[0][b] = b_nx[b];
#pragma unroll
    for (int u = 1; u < RB - 1; ++u) {  // (slot RB - 1 is requested by the first stage below)
        const int kn = u < ipg ? u : ipg - 1;
#pragma unroll
        for (int b = 0; b < NV; ++b) b_ring[u][b] = GNF_BIG_LOAD_B(rsrc, voff, wtile[b] + kn * kstride);
    }
#pragma unroll
    for (int m = 0; m < MW; ++m)
        if (FULL || m < mw) a_frag[m] = *reinterpret_cast<const f32x4*>(arow + 16 * m * kBigLS);

    // the MFMAs of one row tile with B slot SB, then the refill of its A fragment with k-group KA
#define GNF_BIG_MBLOCK(M, SB, KA)                                                                       \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int b = 0; b < NV; ++b)        \
        acc[M][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_ring[SB][b][q], a_frag[M][q], acc[M][b], 0, 0, 0); \
    a_frag[M] = *reinterpret_cast<const f32x4*>(arow + 16 * (M) * kBigLS + 16 * (KA));
    // stage kg: B slot kg % RB; the k-group RB - 1 ahead goes into the slot stage kg - 1 just left (LOADS: not in the tail)
#define GNF_BIG_STAGE(SB, KA, KB, LOADS)                                                                \
    if (LOADS) {                                                                                        \
        _Pragma("unroll") for (int b = 0; b < NV; ++b) b_ring[((SB) + PF) % RB][b] =                    \
            GNF_BIG_LOAD_B(rsrc, voff, wtile[b] + (KB) * kstride);                                      \
    }                                                                                                   \
    GNF_BIG_MBLOCK(0, SB, KA)                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                  \
    if (LOADS) __builtin_amdgcn_sched_group_barrier(0x020, NV, 0);                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NV - 1, 0);                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                  \
    if (!FULL) __builtin_amdgcn_sched_barrier(0);                                                       \
    _Pragma("unroll") for (int m_ = 1; m_ < MW; ++m_) {                                                 \
        if (FULL || m_ < mw) { /* wave-uniform */                                                       \
            GNF_BIG_MBLOCK(m_, SB, KA)                                                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NV, 0);                                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
            if (!FULL) __builtin_amdgcn_sched_barrier(0);                                               \
        }                                                                                               \
    }                                                                                                   \
    if (FULL) __builtin_amdgcn_sched_barrier(0);
#define GNF_BIG_ROUND(KG0)                                                                              \
    _Pragma("unroll") for (int u = 0; u < RB; ++u) {                                                    \
        const int kg = (KG0) + u;                                                                       \
        const int ka = kg + 1 < ipg ? kg + 1 : ipg - 1;                                                 \
        const int kb = kg + PF < ipg ? kg + PF : ipg - 1;                                               \
        GNF_BIG_STAGE(u, ka, kb, true)                                                                  \
    }
    // Wave priority.  The SIMD's arbiter serves the older wave first and a wave that streams MFMAs always has one ready:
    // everything runs at priority 3 except the MFMA stream of a wide chunk, so that the co-resident workgroup's
    // latency-bound phases take the few issue slots they need at once and the stream fills the rest.
    __builtin_amdgcn_s_setprio(0);
    // (one copy of the round: the next layer's first k-group is requested in front of the LAST round - it lands while
    // that round's MFMAs, the write-back and the layer barriers are in progress - behind a scalar branch at the round
    // boundary, where the issue order is pinned anyway)
    int kg0 = 0;
    bool pre_done = false;
    for (; kg0 + RB <= ipg; kg0 += RB) {
        if (kg0 + 2 * RB > ipg) {
            if (have_nx) big_prefetch(nx, lane, b_nx);  // (b_nx was consumed above)
            pre_done = true;
            __builtin_amdgcn_sched_barrier(0);
        }
        GNF_BIG_ROUND(kg0)
    }
    if (!pre_done) {
        if (have_nx) big_prefetch(nx, lane, b_nx);
        __builtin_amdgcn_sched_barrier(0);
    }
    // tail: stages kg0 .. ipg-1 sit in B slots 0 .. rem-1 (requested by the last round, or by the chunk's entry when
    // there was no round)
    const int rem = ipg - kg0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        if (u < rem) {  // wave-uniform
            const int ka = kg0 + u + 1 < ipg ? kg0 + u + 1 : ipg - 1;
            GNF_BIG_STAGE(u, ka, 0, false)
        }
    }
#undef GNF_BIG_ROUND
#undef GNF_BIG_STAGE
#undef GNF_BIG_MBLOCK
    __builtin_amdgcn_s_setprio(3);
    big_layer_end<NV, MW>(act, c, lane, acc, last, slope, s_keep, couple, hp);
}

// Thin layers (1 or 2 column tiles: 4 or 8 MFMAs per k-group cannot hide a weight load): the row tiles are split over
// the waves (<= 2 each), every k-group's fragment is requested before the first MFMA (<= 16 k-groups = 64 registers),
// the A fragments alternate between two slots.  Same k order as big_chunk: bitwise the same sums.
__device__ __forceinline__ void big_chunk_thin(float* __restrict__ act, const BChunk& c, const BChunk& nx, bool have_nx,
                                               int lane, f32x4 (&b_nx)[4], bool last, float slope,
                                               f32x4 (&s_keep)[kBigMT][2], bool couple, int hp) {
    constexpr int NS = kBigMaxW / 16, MW = 2;
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int ipg = c.ipg, mw = c.mw;
    f32x4 acc[MW][1];
#pragma unroll
    for (int m = 0; m < MW; ++m) acc[m][0] = *reinterpret_cast<const f32x4*>(c.bias + 16 * c.col0 + 4 * lgrp);
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.wbase), 0, (int)c.wbytes, 0x00020000);
    const int voff = lane * 16, wtile = c.col0 * 1024, kstride = c.ont * 1024;
    const float* arow = act + (16 * c.m0 + lrow) * kBigLS + 4 * lgrp;
    const int m1 = mw > 1 ? 1 : 0;  // (a wave with one row tile reads it twice and drops the second result)
    f32x4 bst[NS], a_ring[2][MW];
    bst[0] = b_nx[0];
#pragma unroll
    for (int u = 1; u < NS; ++u) bst[u] = GNF_BIG_LOAD_B(rsrc, voff, wtile + (u < ipg ? u : ipg - 1) * kstride);
    a_ring[0][0] = *reinterpret_cast<const f32x4*>(arow);
    a_ring[0][1] = *reinterpret_cast<const f32x4*>(arow + 16 * m1 * kBigLS);
    if (have_nx) big_prefetch(nx, lane, b_nx);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        if (u < ipg) {  // wave-uniform
            const int ka = u + 1 < ipg ? u + 1 : ipg - 1;
            a_ring[(u + 1) % 2][0] = *reinterpret_cast<const f32x4*>(arow + 16 * ka);
            a_ring[(u + 1) % 2][1] = *reinterpret_cast<const f32x4*>(arow + 16 * m1 * kBigLS + 16 * ka);
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bst[u][q], a_ring[u % 2][m][q], acc[m][0], 0, 0, 0);
        }
    }
    big_layer_end<1, MW>(act, c, lane, acc, last, slope, s_keep, couple, hp);
}

// SPLIT: the launch has split row tiles (big_seg_kind != 0 somewhere); the plain instance carries none of that code
template <bool SPLIT>
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_half_big(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    double* red = reinterpret_cast<double*>(act + kBigTM * kBigLS);  // [4] sum(s) | [4] sum(x_new^2)
    int* tab = reinterpret_cast<int*>(red + 16);                     // [K][8]: ipg, ont, boff, -, wp[0] lo/hi, wp[1] lo/hi
    float* bias_lds = reinterpret_cast<float*>(tab + GNF_MAX_LAYERS * 8);  // [2][bias_tot]: every layer's padded bias row, both nets

    // workgroup = block (dispatch order = row order: nothing here gathers, so which XCD a row tile lands on does not
    // matter); its row tiles from the launch's run table (big_plan)
    const int wg = blockIdx.x;
    int mt = 1, row0 = 0, kind = 0, wrun = 0;
    {
        int w = wg, g0 = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int nk = a.big_seg_n[k], sk = a.big_seg_sz[k];
            if (w >= 0 && w < nk) {
                mt = sk;
                kind = SPLIT ? a.big_seg_kind[k] : 0;
                wrun = w;
                g0 += w * sk;
                w = -1;
            } else if (w >= 0) {
                w -= nk;
                g0 += nk * sk;
            }
        }
        row0 = 16 * g0;
    }
    const int rows = a.n_nodes - row0 < 16 * mt ? a.n_nodes - row0 : 16 * mt;   // live rows (> 0)
    // split row tile (kind 1: this workgroup runs its s-net; kind 2: its t-net and the coupling): LDS row tile `mt`, right
    // behind the workgroup's own row tiles (which are whole 16-row tiles whenever a launch has split tiles)
    const int xrow0 = 16 * (a.big_xg0 + wrun);
    const int rows_x = kind == 0 ? 0 : (a.n_nodes - xrow0 < 16 ? a.n_nodes - xrow0 : 16);
    const int mt_s = mt + (kind == 1 ? 1 : 0), mt_t = mt + (kind == 2 ? 1 : 0);
    auto grow = [&](int rl) { return rl < 16 * mt ? row0 + rl : xrow0 + (rl - 16 * mt); };  // LDS row -> node
    auto row_live = [&](int rl) { return rl < 16 * mt ? rl < rows : rl - 16 * mt < rows_x; };
    const int tid = threadIdx.x;
    const int H = a.H, K = a.K;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    __builtin_amdgcn_s_setprio(3);  // (see big_chunk)

    // this wave's share of a layer: column tiles {w, w+4, w+8, w+12} x every row tile when the layer has at least 3 column
    // tiles; a thinner layer spreads its column tiles over the first cg = ont waves of each group and the row tiles over
    // the 4 / cg groups (the 256 -> 32 output layer: 2 column tiles x 2 halves of the rows)
    // mtn: row tiles of this net; mlay: row tiles the split over the wave groups is laid out for - the same for both nets in
    // the LAST layer, whose waves keep s in registers from the s-net's pass to the t-net's (a split tile makes mt_s != mt_t)
    auto assign = [&](BChunk& c, const int mtn, const int mlay) {  // (shifts and masks only: this runs on the scalar unit in front of every layer)
        const int ont = c.ont;
        const int cgl = ont >= 3 ? 2 : (ont <= 1 ? 0 : 1);  // log2 of the waves a group has: 4, 1, 2
        const int gl = 2 - cgl;                             // log2 of the groups: 1, 4, 2
        const int per = (mlay + (1 << gl) - 1) >> gl;       // row tiles per wave group: <= 2 when there are 2 or 4 groups
        const int g = wave >> cgl;
        c.col0 = wave & ((1 << cgl) - 1);
        c.nv = c.col0 < ont ? (ont - c.col0 + kBigWaves - 1) >> 2 : 0;
        c.m0 = g * per;
        const int left = mtn - c.m0;
        c.mw = left < 0 ? 0 : (left < per ? left : per);
        c.thin = cgl < 2;
        c.active = (c.nv > 0 && c.mw > 0) ? 1 : 0;
    };
    auto chunk_from_args = [&](int j, int net) -> BChunk {
        BChunk c;
        c.ipg = a.ipg[j];
        c.ont = a.ont[j];
        c.wbase = a.wp[net][j];
        c.wbytes = (unsigned)c.ipg * (unsigned)c.ont * 1024u;
        c.bias = bias_lds + net * a.bias_tot + a.boff[j];
        assign(c, net ? mt_t : mt_s, j == K - 1 ? mt + (kind ? 1 : 0) : (net ? mt_t : mt_s));
        return c;
    };
    auto chunk_from_tab = [&](int j, int net) -> BChunk {
        const int* row = tab + 8 * j;
        BChunk c;
        c.ipg = __builtin_amdgcn_readfirstlane(row[0]);
        c.ont = __builtin_amdgcn_readfirstlane(row[1]);
        const int boff = __builtin_amdgcn_readfirstlane(row[2]);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(row[4 + 2 * net]);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(row[5 + 2 * net]);
        c.wbase = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
        c.wbytes = (unsigned)c.ipg * (unsigned)c.ont * 1024u;
        c.bias = bias_lds + net * a.bias_tot + boff;
        assign(c, net ? mt_t : mt_s, j == K - 1 ? mt + (kind ? 1 : 0) : (net ? mt_t : mt_s));
        return c;
    };

    // ---- the first chunk's weights start streaming before anything else --------------------------------
    f32x4 b_nx[4];
    BChunk cur = chunk_from_args(0, 0);
    bool have_pre = cur.active;
    if (have_pre) big_prefetch(cur, lane, b_nx);

    for (int i = tid; i < 2 * a.bias_tot; i += kBigThreads)  // (one coalesced copy per net: the bias block is contiguous)
        bias_lds[i] = i < a.bias_tot ? a.bias[0][i] : a.bias[1][i - a.bias_tot];
    if (tid < 8 * K) {  // layer table (one thread per word)
        const int j = tid >> 3, w = tid & 7;
        const unsigned long long p0 = reinterpret_cast<unsigned long long>(a.wp[0][j]);
        const unsigned long long p1 = reinterpret_cast<unsigned long long>(a.wp[1][j]);
        int v = 0;
        switch (w) {
            case 0: v = a.ipg[j]; break;
            case 1: v = a.ont[j]; break;
            case 2: v = a.boff[j]; break;
            case 4: v = (int)(unsigned)p0; break;
            case 5: v = (int)(unsigned)(p0 >> 32); break;
            case 6: v = (int)(unsigned)p1; break;
            case 7: v = (int)(unsigned)(p1 >> 32); break;
            default: break;
        }
        tab[tid] = v;
    }

    // layer-0 input of one net from global rows [n, in0] (the aggregation kernel's or the attention front-end's output):
    // eight 16-byte reads per thread requested before the first is stored (a 64 x 128 tile is exactly eight per thread)
    auto load_h0 = [&](const float* __restrict__ src, const int mtn) {
        const int in0p = a.ipg[0] * 16;
        if ((a.in0 & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const int q4 = in0p >> 2, total = 16 * mtn * q4;
#pragma unroll 1
            for (int base = tid; base < total; base += 8 * kBigThreads) {
                f32x4 v[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int idx = base + g * kBigThreads;
                    const int rl = idx / q4, c = (idx - rl * q4) * 4;
                    const bool live = idx < total && row_live(rl) && c < a.in0;
                    v[g] = live ? *reinterpret_cast<const f32x4*>(src + (int64_t)grow(rl) * a.in0 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const int idx = base + g * kBigThreads;
                    const int rl = idx / q4, c = (idx - rl * q4) * 4;
                    if (idx < total) *reinterpret_cast<f32x4*>(act + rl * kBigLS + c) = v[g];
                }
            }
        } else {
#pragma unroll 1
            for (int idx = tid; idx < 16 * mtn * in0p; idx += kBigThreads) {
                const int rl = idx / in0p, c = idx - rl * in0p;
                act[rl * kBigLS + c] = (row_live(rl) && c < a.in0) ? src[(int64_t)grow(rl) * a.in0 + c] : 0.f;
            }
        }
    };

    // ---- A: the s-net's layer-0 input (see the header); an out-of-place first half-step also copies its conditioning
    // rows on the way ------------------------------------------------------------------------------------------
    load_h0(a.h0[0], mt_s);
    if (a.cond_copy) {  // (a split tile's rows: by the workgroup that couples it)
#pragma unroll 1
        for (int i = tid; i < (rows + (kind == 2 ? rows_x : 0)) * H; i += kBigThreads) {
            const int rl = i / H, f = i - rl * H;
            a.cond_copy[(int64_t)grow(rl) * a.ld + f] = a.x_cond[(int64_t)grow(rl) * a.ld + f];
        }
    }
    __syncthreads();

    // ---- B: s-net, then t-net ------------------------------------------------------------------------------
    f32x4 s_keep[kBigMT][2];
    const int hp = a.ont[K - 1] * 16;  // padded coupling half: t sits at this column offset beside s
    const float slope_hidden = a.act == GNF_ACT_RELU ? 0.f : a.alpha;
    for (int net = 0; net < 2; ++net) {
        for (int j = 0; j < K; ++j) {
            // the layer after this one in the wave's sequence (the t-net's first layer follows the s-net's last)
            const bool last = j == K - 1;
            const bool has_next = !(last && net == 1);
            BChunk nx = cur;
            if (has_next) nx = last ? chunk_from_tab(0, 1) : chunk_from_tab(j + 1, net);
            const bool pre_next = has_next && nx.active;
            if (cur.active) {
                if (!have_pre) big_prefetch(cur, lane, b_nx);
#define GNF_BIG_RUN(NV_, RB_, MW_) \
    big_chunk<NV_, RB_, MW_, false>(act, cur, nx, pre_next, lane, b_nx, last, slope_hidden, s_keep, net == 1, hp)
                if (cur.thin)
                    big_chunk_thin(act, cur, nx, pre_next, lane, b_nx, last, slope_hidden, s_keep, net == 1, hp);
                else if (cur.nv == 4) {
                    if (cur.mw == kBigMT)  // (the shape almost all of a large batch's work runs in: no row-tile branches)
                        big_chunk<4, kBigRBW, kBigMT, true>(act, cur, nx, pre_next, lane, b_nx, last, slope_hidden, s_keep,
                                                                net == 1, hp);
                    else if (cur.mw == 3)  // (its own instance: a 3-tile workgroup in the 4-tile one took a 4-tile workgroup's time)
                        big_chunk<4, kBigRBW, 3, true>(act, cur, nx, pre_next, lane, b_nx, last, slope_hidden, s_keep,
                                                           net == 1, hp);
                    else
                        GNF_BIG_RUN(4, kBigRB2, 2);
                } else if (cur.nv == 3) {  // (the less common widths: one instance each, to keep the code small)
                    GNF_BIG_RUN(3, kBigRBW, kBigMT);
                } else if (cur.nv == 2) {
                    GNF_BIG_RUN(2, 3, kBigMT);
                } else {
                    GNF_BIG_RUN(1, 4, kBigMT);
                }
#undef GNF_BIG_RUN
            } else {
                if (pre_next) big_prefetch(nx, lane, b_nx);
                if (!last) {  // (the two barriers of an in-place layer)
                    __syncthreads();
                    __syncthreads();
                } else if (net == 1) {
                    __syncthreads();  // (the barrier in front of the s | t rows, big_layer_end)
                }
            }
            if (last && net == 0) {
                if (kind == 1 && cur.active) {  // the split tile's s rows: to its partner, through memory the whole device sees
                    float* sx = a.big_split_s + (size_t)wrun * 16 * hp + (lane & 15) * hp + 16 * cur.col0 + 4 * (lane >> 4);
#pragma unroll
                    for (int m = 0; m < kBigMT; ++m)
                        if (m < cur.mw && cur.m0 + m == mt) {
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                if (b < cur.nv) {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        __hip_atomic_store(sx + 16 * kBigWaves * b + r, s_keep[m][b][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                }
                __syncthreads();  // every wave is done with the s-net's last hidden rows
                if (kind == 1 && tid == 0)
                    __hip_atomic_store(a.big_split_flag + wrun, a.big_epoch_set, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                load_h0(a.h0[1], mt_t);
                __syncthreads();
            }
            if (has_next) {
                have_pre = pre_next;
                cur = nx;
            }
        }
    }

    // ---- C: coupling update from the s | t rows in LDS, rows of x coalesced (16 bytes per lane where the widths allow,
    // four requests per thread before the first use); this lane's fp64 shares of sum(s) and sum(x_new^2) ----
    __syncthreads();
    // the partner's s rows never came: the tile's rows of x_upd AND the partial sums turn into NaN instead of the launch
    // hanging - loud in both directions (GRevNet.g returns nodes only, gnn.py:343-373: a sampling pass has no scalar to poison)
    bool lost = false;
    if (kind == 2) {
        // the split tile's s rows from the workgroup that ran its s-net (dispatched before this one, and long done with the
        // s-net by the time this one is through both of its nets): wait for its flag, copy the rows beside t
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(a.big_split_flag + wrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.big_epoch && ++spins < (1 << 20))
                __builtin_amdgcn_s_sleep(32);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            red[15] = spins >= (1 << 20) ? 1.0 : 0.0;
        }
        __syncthreads();
        lost = red[15] != 0.0;
        const float* sx = a.big_split_s + (size_t)wrun * 16 * hp;
#pragma unroll 1
        for (int i = tid; i < 16 * hp; i += kBigThreads) {
            const int r = i / hp, c = i - r * hp;
            const float sv = __hip_atomic_load(sx + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            act[(16 * mt + r) * kBigLS + c] = lost ? __builtin_nanf("") : sv;   // (x exp(NaN) + t = NaN: the rows are written as NaN)
        }
        __syncthreads();
    }
    const int rows_c = rows + (kind == 2 ? rows_x : 0);  // rows this workgroup couples (LDS rows 0 .. rows_c - 1)
    double local = 0.0, local2 = 0.0;
    {
        auto one = [&](float xv, float xr, float sv, float tv, float& xn) {
            const float s_ = sv + xr, t_ = tv + xr;
            xn = a.inverse ? (xv - t_) * expf(-s_) : xv * expf(s_) + t_;
            local += (double)s_;
            local2 += (double)xn * (double)xn;
        };
        const bool v4 = (H & 3) == 0 && (a.ld & 3) == 0 &&
                        ((reinterpret_cast<uintptr_t>(a.x_upd_src) | reinterpret_cast<uintptr_t>(a.x_upd) |
                          reinterpret_cast<uintptr_t>(a.x_cond)) & 15) == 0;
        if (v4) {
            const int q4 = H >> 2, total = rows_c * q4;
#pragma unroll 1
            for (int base = tid; base < total; base += 4 * kBigThreads) {
                f32x4 xv[4], xr[4];
                int off[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int idx = base + g * kBigThreads;
                    const int i = idx < total ? idx : total - 1;
                    const int rl = i / q4, f = (i - rl * q4) * 4;
                    off[g] = (int)(grow(rl) * a.ld) + f;  // < 2^31: choose_big checked n_nodes * ld
                    xv[g] = *reinterpret_cast<const f32x4*>(a.x_upd_src + off[g]);
                    xr[g] = a.residual ? *reinterpret_cast<const f32x4*>(a.x_cond + off[g]) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int idx = base + g * kBigThreads;
                    if (idx < total) {
                        const int rl = idx / q4, f = (idx - rl * q4) * 4;
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(act + rl * kBigLS + f);
                        const f32x4 tv = *reinterpret_cast<const f32x4*>(act + rl * kBigLS + hp + f);
                        f32x4 xn;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float o;
                            one(xv[g][k], xr[g][k], sv[k], tv[k], o);
                            xn[k] = o;
                        }
                        *reinterpret_cast<f32x4*>(a.x_upd + off[g]) = xn;
                    }
                }
            }
        } else {
            const int total = rows_c * H;
#pragma unroll 1
            for (int base = tid; base < total; base += 4 * kBigThreads) {  // four independent elements in flight
                float xv[4], xr[4];
                int off[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int idx = base + g * kBigThreads;
                    const int i = idx < total ? idx : total - 1;
                    const int rl = i / H, f = i - rl * H;
                    off[g] = (int)(grow(rl) * a.ld) + f;
                    xv[g] = a.x_upd_src[off[g]];
                    xr[g] = a.residual ? a.x_cond[off[g]] : 0.f;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int idx = base + g * kBigThreads;
                    if (idx < total) {
                        const int rl = idx / H, f = idx - rl * H;
                        float xn;
                        one(xv[g], xr[g], act[rl * kBigLS + f], act[rl * kBigLS + hp + f], xn);
                        a.x_upd[off[g]] = xn;
                    }
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        local += __shfl_down(local, off, 64);
        local2 += __shfl_down(local2, off, 64);
    }
    if (lane == 0) {
        red[wave] = local;
        red[kBigWaves + wave] = local2;
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0, tot2 = 0.0;
        for (int w = 0; w < kBigWaves; ++w) {
            tot += red[w];
            tot2 += red[kBigWaves + w];
        }
        if (lost) tot = tot2 = __builtin_nan("");
        a.partials[wg] = tot;
        if (a.sq_partials) a.sq_partials[wg] = tot2;
    }
}


// ------------------------------------------------------------------------------------------------
static size_t big_lds_bytes(int bias_tot) {
    return (size_t)kBigTM * kBigLS * sizeof(float) + 16 * sizeof(double) + (size_t)(GNF_MAX_LAYERS * 8) * sizeof(int) +
           (size_t)2 * bias_tot * sizeof(float);
}

// layer widths the in-place form holds: every layer input <= 256 (one LDS row), every hidden width <= 256 (a wave's
// four column tiles), the coupling half <= 128 (s stays in two column tiles' accumulators per wave)
bool big_supported(const GnfMlp* s, int32_t H) {
    if (!s->packed) return false;
    const int K = s->num_layers;
    for (int j = 0; j < K; ++j)
        if (pad16(s->dims[j]) > kBigMaxW) return false;
    if (pad16(s->dims[K]) > kBigMaxH || pad16(H) > kBigMaxH) return false;
    return true;
}

// How the batch's 16-row granules are dealt out (cus = CUs of the device, cap = row tiles per workgroup at most: 4,
// developer override 1 .. 4), as runs of (count, row tiles) in dispatch order:
//   * up to one pass of the chip (g <= cap * 2 * cus granules): an even deal over 2 * cus workgroups, sizes differing by
//     one row tile, the larger ones first - every CU slot gets one workgroup and the launch ends everywhere at once
//     (20 k nodes at the config-4 widths: 149 us against 177 for the 32-row both-nets shape, 214 for whole 64-row tiles);
//   * more than that: whole double rounds of cap-tile workgroups and a closing round (below).  The dispatcher hands a
//     new workgroup to whichever slot frees first, and the SIMD arbiter serves the OLDER of a CU's two workgroups first:
//     the older one runs at its own pace, the younger fills its gaps and becomes the older one in turn - pairs drift out
//     of phase by themselves.  Measured and dropped (tools/ab_shapes.sh, CHANGELOG.md): an even deal over ALL rounds
//     (642 vs 555 us on config 4 while 3-tile workgroups still ran the 4-tile instance) and an opening of cap-tile +
//     half-tile workgroups (the half-size one, served second, takes as long as its partner).
int big_plan(int64_t n_nodes, int cus, int cap, int32_t* seg_n, int32_t* seg_sz, int32_t* seg_kind, int32_t* xg0) {
    for (int k = 0; k < 6; ++k) seg_n[k] = 0, seg_sz[k] = 1;
    if (seg_kind)
        for (int k = 0; k < 6; ++k) seg_kind[k] = 0;
    if (xg0) *xg0 = 0;
    const int64_t g = (n_nodes + 15) / 16;
    // The batch's LAST `tiles` row tiles, dealt evenly over w workgroups (sizes differing by one row tile, the larger ones
    // first in dispatch order), as runs k, k + 1, ...  w = 2 cus: a pair per CU (the first cus workgroups are the CUs' first
    // slots); w = cus: one workgroup per CU.
    // Split row tiles (seg_kind != NULL: the caller has scratch and flags for them).  The deal leaves a top layer of e row
    // tiles - one more for e of the CUs, and the launch ends when those CUs do: 8 row tiles against the 7.2 a CU gets on
    // average on the config-5 batch (1 842 row tiles on 256 CUs).  When that layer covers at most half of the CUs, each of
    // its tiles goes to TWO workgroups on different CUs - one runs its s-net, the other its t-net and the coupling - and
    // the longest CU carries half a row tile more than the shortest instead of a whole one.  The split tiles are the
    // batch's LAST e row tiles; a workgroup holds its own row tiles plus the split one (base + 1 <= cap LDS slots).
    auto deal = [&](int k, int64_t tiles, int64_t w) {
        const int base = (int)(tiles / w);
        const int64_t rem = tiles % w;
        const int64_t layer = w > cus ? cus : w;                  // workgroups per layer of the deal
        const int64_t e = rem <= layer ? rem : rem - layer;       // the partially filled top layer
        if (seg_kind && xg0 && w >= cus && e > 0 && 2 * e <= cus && base + 1 <= cap && base >= 1) {
            if (rem > layer) seg_n[k] = (int32_t)layer, seg_sz[k] = base + 1, ++k;   // every CU's first workgroup
            seg_n[k] = (int32_t)e, seg_sz[k] = base, seg_kind[k] = 1, ++k;          // + the s-net of split tile xg0 + w
            seg_n[k] = (int32_t)e, seg_sz[k] = base, seg_kind[k] = 2, ++k;          // + its t-net (dispatched after its partner)
            seg_n[k] = (int32_t)((rem > layer ? w - layer : w) - 2 * e), seg_sz[k] = base;
            *xg0 = (int32_t)(g - e);
            return;
        }
        if (rem) seg_n[k] = (int32_t)rem, seg_sz[k] = base + 1, ++k;
        seg_n[k] = (int32_t)(w - rem), seg_sz[k] = base;
    };
    if (g <= (int64_t)cap * 2 * cus) {  // up to one pass of the chip: every slot gets one workgroup
        const int64_t w = g < 2 * (int64_t)cus ? g : 2 * (int64_t)cus;
        deal(0, g, w);
        return (int)w;
    }
    // Larger batches: whole double rounds of cap-tile workgroups (a pair per CU each), then ONE closing round laid out by
    // what is left per CU (t = left / cus row tiles):
    //     t <= 1        : 1-tile workgroups
    //     t <= 4        : ONE workgroup of 1 - 4 row tiles per CU (round 4: a lone 3-tile workgroup ends sooner than a pair
    //                     of a 2-tile and a 1-tile one, which stream the weights twice - config 4: 523.9 -> 514.0 us)
    //     else          : an even deal over 2 cus workgroups: 3 + 2, 3 + 3, 4 + 3 row tiles per CU
    // - each with its top layer split when it covers at most half of the CUs.
    // A closing round of whole cap-tile workgroups handed out one by one would run on a part of the CUs only - for
    // as long as a full pair where two of them share a CU (config 4, 19 row tiles per CU: 23 % of the CUs idle for the
    // last quarter of the launch).  Small workgroups are not efficient themselves (a 1-tile workgroup streams the same
    // 1.7 MB of weights as a 4-tile one, and a CU takes them at 15 - 20 bytes per clock whatever the depth of the register
    // ring: 115 k cycles alone, 180 k beside another, against 315 k / 475 k for 4-tile ones) - they keep every CU busy to
    // the end.
    const int64_t per_round = (int64_t)cap * 2 * cus;
    const int64_t full = cap == kBigMT ? g / per_round : 0, left = g - full * per_round;
    if (full > 0 && left > 0) {
        seg_n[0] = (int32_t)(full * 2 * cus), seg_sz[0] = cap;
        const int64_t w = left <= cus ? left : (left <= (int64_t)cap * cus ? (int64_t)cus : 2 * (int64_t)cus);
        deal(1, left, w);
        return (int)(full * 2 * cus + w);
    }
    seg_n[0] = (int32_t)((g + cap - 1) / cap);
    seg_sz[0] = cap;
    return seg_n[0];
}

int big_cu_count() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<int> cu_cache[64];  // multiProcessorCount per device (0 = not asked yet)
    int c = cu_cache[dev & 63].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cu_cache[dev & 63].store(c, std::memory_order_relaxed);
    }
    return c;
}

int launch_half_big(FusedArgs& a, int64_t n_nodes, int cap, hipStream_t st, int* n_wg_out) {
    const bool split = a.big_split_flag && a.big_split_s && a.big_epoch > 0;
    const int n_wg = big_plan(n_nodes, big_cu_count(), cap, a.big_seg_n, a.big_seg_sz, split ? a.big_seg_kind : nullptr, split ? &a.big_xg0 : nullptr);
    a.n_tiles = n_wg;
    const size_t lds = big_lds_bytes(a.bias_tot);  // <= 66.7 KB + 2 * 8 * 256 * 4: two workgroups per CU
    GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_big<false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024));  // (K = 8 layers of 256: 83.3 KB)
                        GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_big<true>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024)));
    bool any_split = false;
    for (int k = 0; k < 6; ++k) any_split = any_split || (a.big_seg_n[k] > 0 && a.big_seg_kind[k] != 0);
    if (any_split)
        hipLaunchKernelGGL(k_half_big<true>, dim3((unsigned)n_wg), dim3(kBigThreads), lds, st, a);
    else
        hipLaunchKernelGGL(k_half_big<false>, dim3((unsigned)n_wg), dim3(kBigThreads), lds, st, a);
    GNF_LAUNCH_CHECK("k_half_big");
    *n_wg_out = n_wg;
    return GNF_OK;
}

}  // namespace gnf
