// Wide linear layers of the layered path (gfx950 / MI355X): y = act(x W + b) for layers the LDS-resident fused kernels
// cannot hold (the data driver's default MLP, train_grevnet_with_data.py:104-117: 2048 x 3), with the large-batch fused
// kernel's inner loop (gnf_fused_big.hip) instead of the generic GEMM tile:
//
//   * the WEIGHTS come straight from L2 into registers in MFMA fragment order (the packed copy every MLP has for the fused
//     kernels, gnf_fused.hip: Wp[kg][nt][lane][q]) - one coalesced 1 KiB load per wave and column tile, no trip through
//     LDS, no barrier on the weight stream.  The generic tile fetched raw W[k][n] rows into LDS and read its B fragments
//     back with four 4-byte LDS reads per column tile and k-group, behind a barrier per 32-wide k-step;
//   * a workgroup is 4 waves (one per SIMD) and owns 64 rows x 256 columns: wave w holds column tiles {w, w + 4, w + 8, w + 12}
//     x the four row tiles = 16 accumulators, 4 weight loads + 4 LDS reads per 64 MFMAs, accumulators transposed (a lane
//     holds four consecutive columns of a row: 16-byte stores);
//   * the ACTIVATION rows pass through one [64][260] LDS buffer in chunks of 256 columns (16 k-groups), the next chunk in
//     registers while the current one is multiplied; 66.7 KB of LDS = two workgroups per CU, each filling the other's
//     chunk boundaries and epilogue;
//   * every accumulator sees bias, then its k-groups and the four MFMAs inside a k-group in the generic tile's order:
//     bitwise the same sums.
// gnn.py:159-180 (snt.nets.MLP: MatMul + Add, activation between layers).
#include "gnf_common.h"
#include "gnf_fused_dev.h"

namespace gnf {

static constexpr int kLbThreads = 256;
static constexpr int kLbRows = 64;
static constexpr int kLbLS = 276;      // LDS row stride (floats): 256 + one k-group a wave may read past a chunk's end + 4
static constexpr int kLbChunk = 256;   // activation columns per LDS chunk
static constexpr int kLbCols = 256;    // output columns per workgroup

struct LinBigArgs {
    const float* x[2];
    float* y[2];
    const float* wp[2];    // packed weights of the layer (the transposed copy for the backward product)
    const float* bias[2];  // its padded bias row, or NULL (backward: no bias)
    const float* aux[2];   // NULL, or [n, O] rows (leading dimension ldaux) whose sign picks act' for every output element:
    int64_t ldaux;         //   y = (x W^T) * act'(aux) - the backward pass's dP_{j-1} = (dP_j W_j^T) * act'(h_j)
    int64_t ldx, ldy;
    int32_t n, I, O;
    int32_t ipg, ont;      // padded input width / 16, padded output width / 16
    int32_t col_blocks;    // 256-column blocks
    int32_t act, apply_act;
    float alpha;
    // the row tiles (16 rows) of a panel are dealt out over wg_per_panel workgroups: the first wg_rem of them own wg_base + 1
    int32_t wg_per_panel, wg_base, wg_rem;
    // the thin LAST layer behind this one, multiplied out of the accumulators (wp2 != NULL): a workgroup's 256 activated
    // output columns are 16 k-groups of the next layer's reduction, its partial [rows, O2] product goes to slab[net] +
    // cb * slab_stride and the consumer adds the col_blocks slabs and the bias (k_coupling / k_coupling_rows)
    const float* wp2[2];
    float* slab[2];
    int64_t slab_stride;
    int32_t O2, ont2;
    int32_t store_y;       // 0: nobody reads this layer's own output (inference): not written
};

static constexpr int kLinShortRowTiles = 4;  // k_linear_short: a round of workgroups is at most 4 x this many row tiles long
static constexpr int kLbFuseTiles = kLinearBigFusedMaxOut / 16;  // widest fused next layer: 128 columns

#define GNF_LB_LOAD_B(RSRC, VOFF, SOFF) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RSRC, VOFF, SOFF, 0))

// MW: row tiles of the workgroup (2 .. 4; the instance has accumulators and stage registers for exactly those)
template <int MW>
__device__ __forceinline__ void lin_big_body(const LinBigArgs& a, float* __restrict__ act, const int cb, const int net, const int row0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lgrp = lane >> 4;
    const int ct0 = 16 * cb + wave;  // this wave's first column tile; the others + 4, + 8, + 12
    int nv = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) nv += (ct0 + 4 * b < a.ont && 4 * b + wave < 16) ? 1 : 0;
    const float* __restrict__ x = a.x[net];
    const float* __restrict__ bias = a.bias[net];
    const float* __restrict__ aux = a.aux[net];
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp[net]), 0, (int)((unsigned)a.ipg * (unsigned)a.ont * 1024u), 0x00020000);
    const int voff = lane * 16;
    const int kstride = a.ont * 1024;
    int wtile[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wtile[b] = (nv > 0 ? ct0 + (b < nv ? 4 * b : 0) : 0) * 1024;

    f32x4 acc[MW][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        // (a wave without a live column tile - padded O not a multiple of 256 - reads tile 0's bias, not past the row)
        const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + 16 * (nv > 0 ? ct0 + (b < nv ? 4 * b : 0) : 0) + 4 * lgrp)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MW; ++m) acc[m][b] = bv;
    }

    // activation chunk: 16 MW rows x 256 columns = 4 MW float4 per thread (thread t: column 4 (t & 63), rows (t >> 6) + 4 q)
    const bool vec = (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    f32x4 stage[4 * MW];
    auto fetch = [&](int c0) {
        const int c = c0 + 4 * (tid & 63);
#pragma unroll
        for (int q = 0; q < 4 * MW; ++q) {
            const int r = row0 + (tid >> 6) + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < a.n && c < a.I) {
                const float* p = x + (int64_t)r * a.ldx + c;
                if (vec && c + 3 < a.I) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
                    v[0] = p[0];
                    if (c + 1 < a.I) v[1] = p[1];
                    if (c + 2 < a.I) v[2] = p[2];
                    if (c + 3 < a.I) v[3] = p[3];
                }
            }
            stage[q] = v;
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int q = 0; q < 4 * MW; ++q)
            *reinterpret_cast<f32x4*>(act + ((tid >> 6) + 4 * q) * kLbLS + 4 * (tid & 63)) = stage[q];
    };

    // weight ring: two named slots (indexed dynamically the compiler keeps both in one array, cannot tell which loads a
    // k-group waits for and drains the queue - the prefetch included - in front of every k-group)
    f32x4 b0[4], b1[4];
    if (nv > 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) b0[b] = GNF_LB_LOAD_B(rsrc, voff, wtile[b]);
    }
    fetch(0);
    const int n_chunks = (a.ipg * 16 + kLbChunk - 1) / kLbChunk;
    const float* arow = act + lrow * kLbLS + 4 * lgrp;
    // one k-group: request the NEXT one's weights into NXT, multiply with CUR row tile by row tile, each row tile's A
    // fragment refilled for the next k-group right behind the last MFMA that reads it (past a chunk's end that read takes
    // the row's padding: the next chunk's first k-group is read again after the barrier).  Issue order pinned as in
    // gnf_fused_big.hip: loads behind MFMAs, never bunched in front of them.  A k-group past the layer's last reads zeros
    // from both sides (the buffer's range check; the chunk's zero-filled columns): an odd count is rounded up.
#define GNF_LB_MBLOCK(M, CUR, KL)                                                                              \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int b = 0; b < 4; ++b)                \
        acc[M][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(CUR[b][q], af[M][q], acc[M][b], 0, 0, 0);             \
    af[M] = *reinterpret_cast<const f32x4*>(arow + 16 * (M) * kLbLS + 16 * ((KL) + 1));
#define GNF_LB_STEP(CUR, NXT, KL)                                                                              \
    {                                                                                                          \
        _Pragma("unroll") for (int b = 0; b < 4; ++b) NXT[b] = GNF_LB_LOAD_B(rsrc, voff, wtile[b] + (kg0 + (KL) + 1) * kstride); \
        GNF_LB_MBLOCK(0, CUR, KL)                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);                                                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                     \
        _Pragma("unroll") for (int m_ = 1; m_ < MW; ++m_) {                                                     \
            GNF_LB_MBLOCK(m_, CUR, KL)                                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                                                \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    __builtin_amdgcn_s_setprio(3);
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads();  // (the previous chunk has been read)
        put();
        __syncthreads();
        if (ch + 1 < n_chunks) fetch((ch + 1) * kLbChunk);  // in flight behind this chunk's MFMAs
        const int kg0 = ch * (kLbChunk / 16);
        const int kgn = a.ipg - kg0 < kLbChunk / 16 ? a.ipg - kg0 : kLbChunk / 16;
        if (nv > 0) {
            f32x4 af[MW];
#pragma unroll
            for (int m = 0; m < MW; ++m) af[m] = *reinterpret_cast<const f32x4*>(arow + 16 * m * kLbLS);
            __builtin_amdgcn_s_setprio(0);  // (the MFMA stream yields issue slots to the co-resident workgroup's latency-bound phases)
            for (int kl = 0; kl < kgn; kl += 2) {
                GNF_LB_STEP(b0, b1, kl)
                GNF_LB_STEP(b1, b0, kl + 1)
            }
            __builtin_amdgcn_s_setprio(3);
        }
    }
#undef GNF_LB_MBLOCK
#undef GNF_LB_STEP
    // epilogue: lane l holds out[row = 16 m + (l & 15)][16 ct + 4 (l >> 4) + r]
    const float slope = a.act == GNF_ACT_RELU ? 0.f : a.alpha;
    float* __restrict__ y = a.y[net];
    const bool yvec = (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b >= nv) continue;
        const int c = 16 * (ct0 + 4 * b) + 4 * lgrp;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const int r = row0 + 16 * m + lrow;
            f32x4 v = acc[m][b];
            if (aux) {  // act'(pre) read off the stored activation: h > 0 <=> pre > 0 (the generic tile's EPI_MASK)
                if (r >= a.n || c >= a.O) continue;
                const float* pa = aux + (int64_t)r * a.ldaux + c;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c + q < a.O) v[q] = pa[q] > 0.f ? v[q] : v[q] * slope;
            } else if (a.apply_act) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], slope * v[q]);
                acc[m][b] = v;
            }
            if (r >= a.n || c >= a.O || !a.store_y) continue;
            float* p = y + (int64_t)r * a.ldy + c;
            if (yvec && c + 3 < a.O) {
                *reinterpret_cast<f32x4*>(p) = v;
            } else {
                p[0] = v[0];
                if (c + 1 < a.O) p[1] = v[1];
                if (c + 2 < a.O) p[2] = v[2];
                if (c + 3 < a.O) p[3] = v[3];
            }
        }
    }
    if (a.wp2[net] == nullptr) return;
    // The next (last, thin) layer on the accumulators: a lane's four consecutive columns of a row are the B operand of
    // y2^T[16 t + .][row] += W2^T fragment x h^T, k-group = this wave's column tile (padded columns: zero activations x zero
    // weight rows).  The fragments of W2 are the same for every row tile: each is loaded ONCE and used for all of them, four
    // column tiles of the thin layer per pass, the next column tile's fragments in flight behind the MFMAs (round 5 loaded
    // them per row tile: sixteen dependent trips to L2 per workgroup, 100 k cycles where the MFMAs take 14 k - round 6's time
    // stamps).  Row tile by row tile the four waves' partial [16, 64] tiles then meet in LDS (the activation buffer is free
    // now), are added in wave order and written to this column block's slab.  Per output element the same products in the
    // same order as before.
    const __amdgpu_buffer_rsrc_t rsrc2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp2[net]), 0, (int)((unsigned)a.ont * (unsigned)a.ont2 * 1024u), 0x00020000);
    constexpr int TH = 4;
    constexpr int PS = 16 * TH + 4, WS = 16 * MW * PS;   // row stride, floats between two waves' partial tiles
    static_assert(4 * WS <= kLbRows * kLbLS, "the four waves' partial tiles of every row tile fit the activation buffer");
    float* __restrict__ part = act + wave * WS;
    float* __restrict__ slab = a.slab[net] + (int64_t)cb * a.slab_stride;
#pragma unroll 1
    for (int t0 = 0; t0 < a.ont2; t0 += TH) {
        f32x4 y2[MW][TH];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int t = 0; t < TH; ++t) y2[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 wa[TH], wb[TH];
        // (a column tile past the thin layer's, a dead column tile of this wave: offset out of the descriptor's range - zeros)
#define GNF_LB_W2(W, B)                                                                                                    \
    _Pragma("unroll") for (int t = 0; t < TH; ++t) W[t] =                                                                  \
        GNF_LB_LOAD_B(rsrc2, (B) < nv && t0 + t < a.ont2 ? voff : 0x7fffffff, ((ct0 + 4 * (B)) * a.ont2 + t0 + t) * 1024);
#define GNF_LB_MM(W, B)                                                                                                    \
    if ((B) < nv) {                                                                                                        \
        _Pragma("unroll") for (int m = 0; m < MW; ++m) _Pragma("unroll") for (int t = 0; t < TH; ++t)                       \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                   \
                y2[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[t][q], acc[m][B][q], y2[m][t], 0, 0, 0);                  \
    }
        GNF_LB_W2(wa, 0)
        GNF_LB_W2(wb, 1)
        GNF_LB_MM(wa, 0)
        GNF_LB_W2(wa, 2)
        GNF_LB_MM(wb, 1)
        GNF_LB_W2(wb, 3)
        GNF_LB_MM(wa, 2)
        GNF_LB_MM(wb, 3)
#undef GNF_LB_W2
#undef GNF_LB_MM
        // all row tiles' partial tiles of the four waves meet in LDS at once ([wave][16 MW rows][PS]: 69.6 KB of the buffer's
        // 70.6) - two barriers per pass, not two per row tile; a thread then adds one row's 16 columns in wave order
        const int cw = a.O2 - 16 * t0 < 16 * TH ? a.O2 - 16 * t0 : 16 * TH;  // live columns of this pass
        __syncthreads();  // (the last chunk / the previous pass's partial tiles have been read)
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int t = 0; t < TH; ++t) *reinterpret_cast<f32x4*>(part + (16 * m + lrow) * PS + 16 * t + 4 * lgrp) = y2[m][t];
        __syncthreads();
        {
            const int r = tid >> 2, c0 = 16 * (tid & 3);      // row of the group, first of this thread's 16 columns
            const int64_t gr = row0 + r;
            if (r < 16 * MW && gr < a.n && c0 < cw) {
                const float* p = act + r * PS + c0;
                float* o = slab + gr * a.O2 + 16 * t0 + c0;
                const bool v4 = (a.O2 & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(p + 4 * j), w1 = *reinterpret_cast<const f32x4*>(p + WS + 4 * j);
                    const f32x4 w2 = *reinterpret_cast<const f32x4*>(p + 2 * WS + 4 * j), w3 = *reinterpret_cast<const f32x4*>(p + 3 * WS + 4 * j);
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = ((w0[q] + w1[q]) + w2[q]) + w3[q];
                    const int c = c0 + 4 * j;
                    if (v4 && c + 3 < cw) {
                        *reinterpret_cast<f32x4*>(o + 4 * j) = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (c + q < cw) o[4 * j + q] = v[q];
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(kLbThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_linear_big(const LinBigArgs a) {
    __shared__ __attribute__((aligned(16))) float act[kLbRows * kLbLS];
    // block -> (panel = column block x net, workgroup of the panel), XCD-aware: block b runs on XCD b % 8, and a 256-column
    // weight panel (2 MB at 2048 inputs) has to stay in that XCD's 4 MB L2 for every workgroup that uses it - so each XCD gets
    // a CONTIGUOUS range of the (panel-major) order and works its panels off one after the other.
    const int64_t nwg = gridDim.x, bid = blockIdx.x;
    const int64_t xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int64_t L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int panel = (int)(L / a.wg_per_panel), w = (int)(L - (int64_t)panel * a.wg_per_panel);
    const int cb = panel >> 1, net = panel & 1;
    const int mt = a.wg_base + (w < a.wg_rem ? 1 : 0);
    const int rt0 = w < a.wg_rem ? w * (a.wg_base + 1) : a.wg_rem * (a.wg_base + 1) + (w - a.wg_rem) * a.wg_base;
    if (mt >= 4)
        lin_big_body<4>(a, act, cb, net, 16 * rt0);
    else if (mt == 3)
        lin_big_body<3>(a, act, cb, net, 16 * rt0);
    else
        lin_big_body<2>(a, act, cb, net, 16 * rt0);  // (a 1-tile workgroup multiplies a row tile of zeros beside its own)
}

static inline int lb_pad16(int v) { return (v + 15) & ~15; }

// y[q] = act(x[q] W_j + b_j) of nets[q] (nj = 1 or 2 nets with identical shapes), layer j, from the nets' packed weights;
// transposed = true: y[q] = (x[q] W_j^T) * act'(aux[q]) from the transposed packed copy (the backward pass's dX product; aux ==
// NULL: no mask).  1 = not this kernel's case (no packed copy, a narrow output, a launch that would not fill the chip): the
// caller runs the generic tile.
static int launch_linear_big_impl(const GnfMlp* const* nets, int nj, int j, bool transposed, const float* const* x, int64_t ldx,
                                  float* const* y, int64_t ldy, const float* const* aux, int64_t ldaux, int64_t n, int act, float alpha,
                                  int apply_act, hipStream_t st, float* const* slab = nullptr, int32_t* n_slabs = nullptr) {
    const GnfMlp* m = nets[0];
    // I / O: reduction length and output width of THIS product
    const int I = transposed ? m->dims[j + 1] : m->dims[j], O = transposed ? m->dims[j] : m->dims[j + 1];
    // (a short reduction - the 100 -> 2048 first layer - is all prologue and epilogue here: 45 us against the generic tile's 24)
    const bool mine = transposed ? linear_big_bwd_layer(m->dims[j], m->dims[j + 1]) : linear_big_fwd_layer(m->dims[j], m->dims[j + 1]);
    if (nj != 2 || !nets[0]->packed || !nets[1]->packed || !mine || n > (int64_t)INT32_MAX - 64) return 1;
    const int ipg = lb_pad16(I) / 16, ont = lb_pad16(O) / 16;
    const int col_blocks = (ont * 16 + kLbCols - 1) / kLbCols;
    // Row tiles per workgroup.  With whole 64-row workgroups the data driver's batch is 43 x 16 = 688 workgroups on 512
    // slots: 2.7 per CU, some CU takes 3 (0.90 of whatever a pair reaches).  Instead every slot gets the same NUMBER of
    // workgroups (R rounds) and the panels' row tiles are dealt evenly over them, 2 - 4 row tiles each, the larger ones first:
    // 170 row tiles per panel = 42 x 3 + 22 x 2 there, (3 + 3) and (3 + 2) row tiles per slot.
    const int64_t n_rt = (n + 15) / 16, panels = (int64_t)col_blocks * 2, slots = 2 * (int64_t)big_cu_count();
    const int64_t units = n_rt * panels;
    if (units < 2 * slots) return 1;                                   // would not fill the chip with 2-tile workgroups
    const int64_t rounds = (units + 4 * slots - 1) / (4 * slots);
    int64_t wpp = (rounds * slots + panels - 1) / panels;                 // workgroups per panel
    if (wpp * 4 < n_rt) wpp = (n_rt + 3) / 4;
    if (wpp > n_rt) wpp = n_rt;
    const int wg_base = (int)(n_rt / wpp), wg_rem = (int)(n_rt % wpp);
    if (wg_base + (wg_rem ? 1 : 0) > 4 || wg_base < 1) return 1;
    int64_t woff = 0, wtot = 0, boff = 0, btot = 0;
    for (int i = 0; i < m->num_layers; ++i) {
        const int64_t w = (int64_t)lb_pad16(m->dims[i]) * lb_pad16(m->dims[i + 1]);
        if (i < j) woff += w, boff += lb_pad16(m->dims[i + 1]);
        wtot += w;
        btot += lb_pad16(m->dims[i + 1]);
    }
    if ((int64_t)ipg * ont * 1024 >= ((int64_t)1 << 31)) return 1;  // (32-bit buffer offsets)
    LinBigArgs a;
    for (int q = 0; q < 2; ++q) {
        a.x[q] = x[q], a.y[q] = y[q];
        // packed layout (gnf_fused.hip): [Wp_0 .. Wp_{K-1} | bias rows | WpT_0 .. WpT_{K-1}], WpT_j = the fragments of W_j^T
        a.wp[q] = nets[q]->packed + (transposed ? wtot + btot : 0) + woff;
        a.bias[q] = transposed ? nullptr : nets[q]->packed + wtot + boff;
        a.aux[q] = aux ? aux[q] : nullptr;
    }
    a.ldaux = ldaux;
    a.ldx = ldx, a.ldy = ldy;
    a.n = (int32_t)n, a.I = I, a.O = O, a.ipg = ipg, a.ont = ont, a.col_blocks = col_blocks;
    a.act = act, a.apply_act = apply_act, a.alpha = alpha;
    a.wg_per_panel = (int32_t)wpp, a.wg_base = wg_base, a.wg_rem = wg_rem;
    a.wp2[0] = a.wp2[1] = nullptr, a.slab[0] = a.slab[1] = nullptr;
    a.slab_stride = 0, a.O2 = a.ont2 = 0, a.store_y = 1;
    if (slab) {  // layer j + 1 (the last one) out of this layer's accumulators
        if (transposed || !apply_act || j + 2 != m->num_layers || !linear_big_fused_last(m, j + 1)) return 1;
        const int O2 = m->dims[j + 2];
        for (int q = 0; q < 2; ++q) a.wp2[q] = nets[q]->packed + woff + (int64_t)lb_pad16(I) * lb_pad16(O), a.slab[q] = slab[q];
        a.slab_stride = n * O2, a.O2 = O2, a.ont2 = lb_pad16(O2) / 16;
        a.store_y = y[0] != nullptr;
        *n_slabs = col_blocks;
    }
    hipLaunchKernelGGL(k_linear_big, dim3((unsigned)(wpp * panels)), dim3(kLbThreads), 0, st, a);
    GNF_LAUNCH_CHECK("k_linear_big");
    return GNF_OK;
}

int launch_linear_big(const GnfMlp* const* nets, int nj, int j, const float* const* x, int64_t ldx, float* const* y, int64_t ldy,
                      int64_t n, int act, float alpha, int apply_act, hipStream_t st) {
    return launch_linear_big_impl(nets, nj, j, false, x, ldx, y, ldy, nullptr, 0, n, act, alpha, apply_act, st);
}

// Layers j and j + 1 (the thin last layer) of nets[0..2) in one launch: slab[q] receives *n_slabs partial [n, O_{j+1}] products
// (dense, one after the other) whose sum + b_{j+1} is the net's output; y[q] == NULL: layer j's own output is not kept.
// slab[q] has to hold lb_fused_slabs(O_j) * n * O_{j+1} floats.  1 = not this kernel's case.
int launch_linear_big_fused(const GnfMlp* const* nets, int nj, int j, const float* const* x, int64_t ldx, float* const* y, int64_t ldy,
                            float* const* slab, int32_t* n_slabs, int64_t n, int act, float alpha, hipStream_t st) {
    return launch_linear_big_impl(nets, nj, j, false, x, ldx, y, ldy, nullptr, 0, n, act, alpha, 1, st, slab, n_slabs);
}

int linear_big_fused_slabs(int O) { return (lb_pad16(O) + kLbCols - 1) / kLbCols; }

// ---- short reductions into wide layers (the first layer of a wide net: 100 / 164 -> 2048) ----------------------------------
// The other extreme of the same product: the reduction is a handful of k-groups and the OUTPUT is what costs (44 MB per
// half-step on the data driver's batch).  Per 128 x 128 tile the generic kernel spent its time in prologue, barriers and
// epilogue (36 us for 2.2 GFLOP on wide_fc, 48.5 us for 3.7 GFLOP on the data driver's nets); here nothing is staged at all:
//   * a wave keeps the WHOLE reduction of its NB column tiles in registers (KG x NB fragments, loaded once per workgroup),
//   * the activation rows come straight from global memory in operand order - with transposed accumulators the operand of
//     lane (row r, group g) for k-group kg is the 16 bytes x[r][16 kg + 4 g ..], one dwordx4 load, no LDS, no barrier -
//     the next row tile's fragments in flight behind this one's MFMAs,
//   * a lane stores four consecutive columns of its row.
// The four waves of a workgroup share the row loads through L1 and differ in their column tiles.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct LinShortArgs {
    const float* x[2];
    float* y[2];
    const float* wp[2];
    const float* bias[2];
    int64_t ldx, ldy;
    int32_t n, I, O, ipg, ont;
    // a panel's row tiles are dealt over wg_per_panel workgroups, the first wg_rem of them take wg_base + 1 (k_linear_big's deal)
    int32_t wg_per_panel, wg_base, wg_rem;
    int32_t act, apply_act;
    float alpha;
};

template <int KG, int NB>
__global__ __launch_bounds__(kLbThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_linear_short(const LinShortArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, lgrp = lane >> 4;
    // block -> (panel = column block x net, row chunk): panel-major per XCD as in k_linear_big (the panel's fragments stay in one L2)
    const int64_t nwg = gridDim.x, bid = blockIdx.x;
    const int64_t xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int64_t L = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int panel = (int)(L / a.wg_per_panel), chunk = (int)(L - (int64_t)panel * a.wg_per_panel);
    const int cb = panel >> 1, net = panel & 1;
    const int ct0 = 4 * NB * cb + wave;  // column tiles ct0, + 4, ...
    int nv = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) nv += ct0 + 4 * b < a.ont ? 1 : 0;
    if (nv == 0) return;  // (no barrier in this kernel)
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp[net]), 0, (int)((unsigned)a.ipg * (unsigned)a.ont * 1024u), 0x00020000);
    f32x4 w[KG][NB];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int b = 0; b < NB; ++b)  // (k-groups past the layer's: the buffer's range check returns zeros)
            w[kg][b] = GNF_LB_LOAD_B(rw, lane * 16, (kg * a.ont + ct0 + (b < nv ? 4 * b : 0)) * 1024);
    f32x4 bv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bv[b] = *reinterpret_cast<const f32x4*>(a.bias[net] + 16 * (ct0 + (b < nv ? 4 * b : 0)) + 4 * lgrp);
    const float* __restrict__ x = a.x[net];
    float* __restrict__ y = a.y[net];
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((int64_t)a.n * a.ldx * 4), 0x00020000);
    constexpr int kOut = 0x7fffffff;
    const int rt0 = chunk < a.wg_rem ? chunk * (a.wg_base + 1) : a.wg_rem * (a.wg_base + 1) + (chunk - a.wg_rem) * a.wg_base;
    const int rt1 = rt0 + a.wg_base + (chunk < a.wg_rem ? 1 : 0);
    // this lane's fragment of k-group kg starts 64 kg + 16 lgrp bytes into its row; only the layer's LAST k-group can be cut
    // by the layer's width (whole float4s: a lane's fragment is inside or outside), those lanes read nothing
    // (KG >= the layer's k-groups, all of them unrolled without a branch: a group past the layer's reads zeros on both sides)
    const int i4 = 4 * a.I - 16 * lgrp;  // fragment of k-group kg inside the row <=> 64 kg < i4
    auto row_off = [&](int rt) {
        const int r = 16 * rt + lrow;
        return rt < rt1 && r < a.n ? r * (int)a.ldx * 4 + 16 * lgrp : kOut;
    };
    const float slope = !a.apply_act ? 1.f : a.act == GNF_ACT_RELU ? 0.f : a.alpha;  // (max(v, 1 v) = v: no activation)
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)((int64_t)a.n * a.ldy * 4), 0x00020000);
    f32x4 xf[KG];
    {
        const int ro = row_off(rt0);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) xf[kg] = GNF_LB_LOAD_B(rx, 64 * kg < i4 ? ro : kOut, 64 * kg);
        // (NB stores that write nothing: the loop is then entered with the same sequence of outstanding operations it is
        // re-entered with - the wait counts in front of the MFMAs are the minimum over both ways in)
#pragma unroll
        for (int b = 0; b < NB; ++b) __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, ry, kOut, 16 * b, 0);
    }
    for (int rt = rt0; rt < rt1; ++rt) {
        const int rn = row_off(rt + 1);  // the next row tile's fragments are requested behind the MFMAs that read this one's
        f32x4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = bv[b];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kg][b][q], xf[kg][q], acc[b], 0, 0, 0);
            xf[kg] = GNF_LB_LOAD_B(rx, 64 * kg < i4 ? rn : kOut, 64 * kg);
            // (issue order pinned: the refill right behind the MFMAs that read the fragment - left alone the scheduler sinks
            // all refills behind the tile's last MFMA and every row tile waits out a full load latency)
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // Stores through a buffer descriptor, every one of them issued on every path (rows past the batch, dead column tiles:
        // offset out of range, nothing written).  With the stores inside branches the compiler has to assume the path WITHOUT
        // them when it counts what is outstanding behind a fragment load, and its s_waitcnt vmcnt (loads and stores share the
        // in-order counter on gfx9) then waits for the previous row tile's stores to be acknowledged in the middle of the MFMAs.
        const int so = 16 * rt + lrow < a.n ? (16 * rt + lrow) * (int)a.ldy * 4 : kOut;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = 16 * (ct0 + 4 * b) + 4 * lgrp;
            f32x4 v = acc[b];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], slope * v[q]);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (so != kOut && b < nv && c < a.O) ? so + 4 * c : kOut, 0, 0);
        }
    }
}

// y[q] = act(x[q] W_j + b_j) for a short reduction into a wide layer; 1 = not this kernel's case
int launch_linear_short(const GnfMlp* const* nets, int nj, int j, const float* const* x, int64_t ldx, float* const* y, int64_t ldy,
                        int64_t n, int act, float alpha, int apply_act, hipStream_t st) {
    const GnfMlp* m = nets[0];
    const int I = m->dims[j], O = m->dims[j + 1];
    // (operand fragments and output pieces are whole, aligned float4s; 32-bit buffer offsets)
    if (nj != 2 || !nets[0]->packed || !nets[1]->packed || !linear_short_fwd_layer(I, O) || (ldx & 3) || (ldy & 3) || (O & 3) ||
        ((reinterpret_cast<uintptr_t>(x[0]) | reinterpret_cast<uintptr_t>(x[1]) | reinterpret_cast<uintptr_t>(y[0]) |
          reinterpret_cast<uintptr_t>(y[1])) & 15) ||
        n * ldx * 4 >= ((int64_t)1 << 31) || n * ldy * 4 >= ((int64_t)1 << 31))
        return 1;
    const int ipg = lb_pad16(I) / 16, ont = lb_pad16(O) / 16;
    const bool narrow = ipg > 8;              // longer reductions: two column tiles per wave (the fragments have to fit the registers)
    const int cols = narrow ? 128 : 256;
    const int col_blocks = (ont * 16 + cols - 1) / cols;
    const int64_t n_rt = (n + 15) / 16, panels = (int64_t)col_blocks * 2;
    if (n_rt * panels < 2 * (int64_t)big_cu_count()) return 1;
    int64_t woff = 0, wtot = 0, boff = 0;
    for (int i = 0; i < m->num_layers; ++i) {
        const int64_t w = (int64_t)lb_pad16(m->dims[i]) * lb_pad16(m->dims[i + 1]);
        if (i < j) woff += w, boff += lb_pad16(m->dims[i + 1]);
        wtot += w;
    }
    LinShortArgs a;
    for (int q = 0; q < 2; ++q) {
        a.x[q] = x[q], a.y[q] = y[q];
        a.wp[q] = nets[q]->packed + woff;
        a.bias[q] = nets[q]->packed + wtot + boff;
    }
    a.ldx = ldx, a.ldy = ldy;
    a.n = (int32_t)n, a.I = I, a.O = O, a.ipg = ipg, a.ont = ont;
    // every CU slot (two workgroups per CU by registers) gets ONE workgroup per round with its share of the panel's row tiles
    // (fixed chunks of 4 row tiles were 688 workgroups on 512 slots: two rounds for 1.34 rounds of work); rounds grow with the
    // batch so that a workgroup's walk stays short enough to leave the dispatcher something to balance
    const int64_t slots = 2 * (int64_t)big_cu_count();
    const int64_t rounds = (n_rt * panels + kLinShortRowTiles * 4 * slots - 1) / (kLinShortRowTiles * 4 * slots);
    int64_t wpp = (rounds * slots + panels - 1) / panels;
    if (wpp > n_rt) wpp = n_rt;
    a.wg_per_panel = (int32_t)wpp, a.wg_base = (int32_t)(n_rt / wpp), a.wg_rem = (int32_t)(n_rt % wpp);
    a.act = act, a.apply_act = apply_act, a.alpha = alpha;
    const dim3 grid((unsigned)(panels * wpp));
    if (ipg <= 7)
        hipLaunchKernelGGL((k_linear_short<7, 4>), grid, dim3(kLbThreads), 0, st, a);
    else if (ipg == 8)
        hipLaunchKernelGGL((k_linear_short<8, 4>), grid, dim3(kLbThreads), 0, st, a);
    else if (ipg <= 11)
        hipLaunchKernelGGL((k_linear_short<11, 2>), grid, dim3(kLbThreads), 0, st, a);
    else
        hipLaunchKernelGGL((k_linear_short<13, 2>), grid, dim3(kLbThreads), 0, st, a);
    GNF_LAUNCH_CHECK("k_linear_short");
    return GNF_OK;
}

// dX[q] = (dY[q] W_j^T) * act'(h[q]) (h == NULL: no mask) from the transposed fragments every packed MLP carries
int launch_linear_big_dx(const GnfMlp* const* nets, int nj, int j, const float* const* dy, int64_t lddy, float* const* dx, int64_t lddx,
                         const float* const* h, int64_t ldh, int64_t n, int act, float alpha, hipStream_t st) {
    return launch_linear_big_impl(nets, nj, j, true, dy, lddy, dx, lddx, h, ldh, n, act, alpha, 0, st);
}

}  // namespace gnf
