// Fused coupling half-step for gfx950 (MI355X): ONE launch does, for a tile of TM = 16*MT nodes,
//   A  CSR segmented reduce of neighbour rows (coalesced row reads, sum | sum/max(deg,1))   gnn.py:103-104,117-118,151-156
//      + combine  eps*x+agg | [x || agg]  straight into LDS                                 gnn.py:123 | 108-109
//   B  the s-net and/or t-net MLP, all K layers, activations resident in LDS, on the        gnn.py:159-180
//      exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain), weights
//      streamed from L2 in pre-packed fragment order (one buffer_load_dwordx4 per 4*MT MFMAs)
//   C  x_upd <- x_upd*exp(s)+t  |  (x_upd-t)*exp(-s)  and a block-reduced fp64 sum(s)        gnn.py:322-323,337-338 | 359,372
// so the [E,H] edge tensor, the aggregated tensor and every MLP activation of the reference's TF
// graph never touch HBM.
//
// Two workgroup shapes (measured on MI355X, see DESIGN.md "per-CU weight streaming"):
//   NETS = 2  one workgroup = a node tile x BOTH nets: waves 0-3 run the s-net, waves 4-7 the
//             t-net, the coupling update C happens in the same launch.  Used when there are enough
//             node tiles to fill the chip.
//   NETS = 1  one workgroup = a node tile x ONE net (all 8 waves on it); s / t tiles go to a global
//             scratch and the coupling update is the small k_coupling launch.  Half the LDS
//             footprint: keeps layers up to 1024 wide on the fused path.  XCDs 0-3 take the s-net,
//             XCDs 4-7 the t-net, so each XCD L2 holds one net's weights only.
//
// Inside a net, wave w owns output column tiles {w, w+WPN, w+2*WPN, ...} (16 columns each, WPN =
// waves per net) and processes up to four at a time: one ds_read_b128 of the A fragment (16 nodes x
// 16 k) per M-tile feeds 4*NV MFMAs.  K order inside a group of 16 is permuted identically on both
// operands (k = 16*kg + 4*(lane>>4) + q) so that A is one 16-byte LDS read and B one 16-byte global
// read.  Layer outputs go back to LDS in the accumulator layout through a ping-pong pair of
// [TM][LS] buffers per net, LS = widest padded layer + 4 floats.  Biases are staged in LDS once.
//
// Zero padding: every layer width is padded to a multiple of 16 in the packed weights (pad rows,
// pad columns and pad biases are 0), so padded activations are act(0) = 0 and never contribute.
#include <cstring>
#include "gnf_common.h"

namespace gnf {

static constexpr int kFusedThreads = 512;
static constexpr int kLdsLimit = 160 * 1024;
static constexpr int kRowptrPad = 40;   // LDS ints for rowptr[row0 .. row0+TM] (TM <= 32)
static constexpr int kColCap = 2048;    // LDS ints for the tile's col segment (longer segments stay in global memory)

static inline int pad16(int v) { return (v + 15) & ~15; }

// ------------------------------------------------------------------------------------------------
// packed weight layout of one MLP (floats):
//     Wp_0 | Wp_1 | ... | Wp_{K-1} | bias_0 | ... | bias_{K-1} | WpT_0 | ... | WpT_{K-1}
//   (WpT_j: the same fragment order for W_j^T, read by the fused backward kernel gnf_fused_bwd.hip:
//    WpT_j[Op/16][Ip/16][64 lanes][4],  WpT[kg][nt][lane][q] = W[16*nt + (lane&15)][16*kg + 4*(lane>>4) + q])
//   Wp_j[Ip/16][Op/16][64 lanes][4]:  Wp[kg][nt][lane][q] = W[16*kg + 4*(lane>>4) + q][16*nt + (lane&15)]
//   (k-group major: at a given k-group the 8 waves of a workgroup - and every workgroup of the chip, which
//   all walk the stream in step - read ONE contiguous Op/16 KiB region, so the requests spread over all
//   L2 channels; tile-major order put them 16 KiB apart on a quarter of the channels)
//   bias_j[Op]; everything outside [I,O) is 0.  The bias block is contiguous so a workgroup stages it
//   into LDS with one coalesced copy.
// ------------------------------------------------------------------------------------------------
int64_t packed_floats(const GnfMlp* m) {
    int64_t tot = 0;
    for (int j = 0; j < m->num_layers; ++j) {
        const int64_t ip = pad16(m->dims[j]), op = pad16(m->dims[j + 1]);
        tot += 2 * ip * op + op;
    }
    return tot;
}

__global__ __launch_bounds__(256) void k_pack_layer(const float* __restrict__ W,
                                                    const float* __restrict__ b, int I, int O, int Ip,
                                                    int Op, float* __restrict__ wout,
                                                    float* __restrict__ bout, float* __restrict__ wtout) {
    const int64_t nw = (int64_t)Ip * Op;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nw) {
        const int q = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        const int64_t blk = i >> 8;  // kg * (Op/16) + nt
        const int kgs = Ip >> 4;
        const int nts = Op >> 4;
        const int kg = (int)(blk / nts), nt = (int)(blk % nts);
        (void)kgs;
        const int k = 16 * kg + 4 * (lane >> 4) + q;
        const int c = 16 * nt + (lane & 15);
        wout[i] = (k < I && c < O) ? W[(int64_t)k * O + c] : 0.f;
        // transposed copy: k-groups run over the OUTPUT axis, column tiles over the input axis
        const int nts_t = Ip >> 4;
        const int kg_t = (int)(blk / nts_t), nt_t = (int)(blk % nts_t);
        const int ko = 16 * kg_t + 4 * (lane >> 4) + q;
        const int ci = 16 * nt_t + (lane & 15);
        wtout[i] = (ci < I && ko < O) ? W[(int64_t)ci * O + ko] : 0.f;
    } else if (i < nw + Op) {
        const int c = (int)(i - nw);
        bout[c] = c < O ? b[c] : 0.f;
    }
}

static int64_t packed_weight_floats(const GnfMlp* m) {
    int64_t tot = 0;
    for (int j = 0; j < m->num_layers; ++j) tot += (int64_t)pad16(m->dims[j]) * pad16(m->dims[j + 1]);
    return tot;
}

int launch_pack_mlp(const GnfMlp* m, float* packed, hipStream_t st) {
    int64_t woff = 0, boff = packed_weight_floats(m);
    int64_t toff = boff;
    for (int j = 0; j < m->num_layers; ++j) toff += pad16(m->dims[j + 1]);
    for (int j = 0; j < m->num_layers; ++j) {
        const int I = m->dims[j], O = m->dims[j + 1], Ip = pad16(I), Op = pad16(O);
        const int64_t tot = (int64_t)Ip * Op + Op;
        hipLaunchKernelGGL(k_pack_layer, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->W[j],
                           m->b[j], I, O, Ip, Op, packed + woff, packed + boff, packed + toff + woff);
        GNF_LAUNCH_CHECK("k_pack_layer");
        woff += (int64_t)Ip * Op;
        boff += Op;
    }
    return GNF_OK;
}


}  // namespace gnf
#include "gnf_fused_dev.h"
#include "gnf_attn_front_dev.h"
namespace gnf {

// FRONT (MT = 1, NETS = 2, attention GNNs on sparse batches): the attention front-end (gnf_attn_front_dev.h) runs as this
// kernel's prologue and leaves the layer-0 input rows of both nets in the activation buffers - no launch boundary, no
// trip of those rows through global memory.  Its staging area (x rows, q | v of the sender window: 149 KB at the
// reference's head geometry) ALIASES the activation buffers; bias / reduction scratch / layer table sit behind it.
template <int MT, int NETS, bool STASH = false, bool FRONT = false, bool FIXED = false>
__global__ __launch_bounds__(kFusedThreads) void k_half_fused(const FusedArgs a, const FrontArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(!FRONT || (MT == 1 && NETS == 2), "the attention prologue exists for the 16-row both-nets shape");
    constexpr int TM = 16 * MT;
    constexpr int WPN = 8 / NETS;  // waves per net
    const int LS = a.LS;
    // [net][pingpong][TM][LS] | bias [NETS][bias_tot] | reduction scratch
    auto buf = [&](int net_, int pp_) -> float* { return smem + (2 * net_ + pp_) * TM * LS; };
    int base_floats = 2 * NETS * TM * LS;
    if constexpr (FRONT) {
        const int ft = front_lds(FIXED ? front_dims(32, 8, 10, 10, 80) : fa.d).total;
        base_floats = ft > base_floats ? ft : base_floats;
    }
    float* bias_lds = smem + base_floats;
    double* red = reinterpret_cast<double*>(bias_lds + NETS * a.bias_tot + ((NETS * a.bias_tot) & 1));

    // blockIdx -> (tile, net).  Block b is dispatched to XCD b % 8.
    //  NETS = 2: XCD-aware bijective remap, consecutive tiles (neighbouring nodes) share an XCD.
    //  NETS = 1: XCDs 0-3 run the s-net, XCDs 4-7 the t-net (each XCD L2 caches one net's weights).
    int tile, net0;
    {
        const int bid = blockIdx.x;
        if (NETS == 2) {
            const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
            tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
            net0 = 0;
        } else {
            const int xcd = bid & 7;
            net0 = xcd >> 2;
            tile = (bid >> 3) * 4 + (xcd & 3);
            if (tile >= a.n_tiles) return;  // grid is padded to a multiple of 8
        }
    }
    const int row0 = tile * TM;
    const int tid = threadIdx.x;
    const int H = a.H;

    // wave-uniform ids go through readfirstlane so that hipcc keeps them (and everything derived:
    // net, tile offsets, buffer descriptors, branches) in SGPRs instead of waterfalling on VGPRs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nl = wave / WPN;  // local net of this wave
    // column-tile owner index inside the net.  The second net's ownership is rotated by half a
    // turn: waves w and w+4 share a SIMD, so on a layer with fewer tiles than waves (a 32-wide output
    // layer has 2 tiles for 4 waves) the s-net keeps SIMDs 0-1 busy and the t-net SIMDs 2-3, instead
    // of both nets queueing on SIMDs 0-1.
    const int wl = (wave % WPN + nl * (WPN / 2)) % WPN;
    const int net = net0 + nl;
    const int voff = lane * 16;

    // Per-layer descriptors live in a small LDS table: indexing the by-value kernel arguments with a
    // run-time layer index costs a dependent scalar-memory round trip (~1.3k cycles measured) at the
    // start of EVERY layer; an LDS read is ~10x cheaper.  Row j: {ipg, ont, boff, -, wp[net0] lo/hi,
    // wp[net0+1] lo/hi}.  Only the very first chunk (prefetched before the table exists) reads the
    // kernel arguments directly.
    int* tab = reinterpret_cast<int*>(red + 8);
    auto fill_chunk = [&](WChunk& c, int j, int ipg_, int ont_, int boff_, const float* wb, int nt0) {
        c.wbase = wb;
        c.wbytes = (unsigned)ipg_ * (unsigned)ont_ * 1024u;
        c.ipg = ipg_;
        c.ont = ont_;
        c.boff = boff_;
        c.nt0 = nt0;
        const int nv = (ont_ - nt0 + WPN - 1) / WPN;
        c.nv = nv > 4 ? 4 : nv;
        c.layer = j;
    };
    auto chunk_from_args = [&](int j, int nt0) -> WChunk {
        WChunk c;
        fill_chunk(c, j, a.ipg[j], a.ont[j], a.boff[j], a.wp[net][j], nt0);
        return c;
    };
    auto chunk_from_tab = [&](int j, int nt0) -> WChunk {
        const int* row = tab + 8 * j;
        const int ipg_ = __builtin_amdgcn_readfirstlane(row[0]);
        const int ont_ = __builtin_amdgcn_readfirstlane(row[1]);
        const int boff_ = __builtin_amdgcn_readfirstlane(row[2]);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(row[4 + 2 * nl]);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(row[5 + 2 * nl]);
        WChunk c;
        fill_chunk(c, j, ipg_, ont_, boff_,
                   reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo), nt0);
        return c;
    };
    // the chunk after `c` in this wave's sequence (layer == K: none); layers whose column tiles do
    // not reach this wave (e.g. a 32-wide last layer has 2 tiles for 4 waves) are skipped
    auto next_chunk = [&](const WChunk& c) -> WChunk {
        int j = c.layer, nt0 = c.nt0 + 4 * WPN;
        if (nt0 < c.ont) {
            WChunk n = c;
            n.nt0 = nt0;
            const int nv = (c.ont - nt0 + WPN - 1) / WPN;
            n.nv = nv > 4 ? 4 : nv;
            return n;
        }
        for (++j; j < a.K; ++j) {
            const int ont_ = __builtin_amdgcn_readfirstlane(tab[8 * j + 1]);
            if (wl < ont_) return chunk_from_tab(j, wl);
        }
        WChunk n = c;
        n.layer = a.K;
        return n;
    };
    WChunk cur;
    {
        int j0 = 0;
        while (j0 < a.K && wl >= a.ont[j0]) ++j0;
        cur = chunk_from_args(j0 < a.K ? j0 : 0, j0 < a.K ? wl : 0);
        if (j0 >= a.K) cur.layer = a.K;  // (cannot happen: the last layer has >= 1 tile; wave 0 runs it)
    }

    // ---- weights of the first chunk start streaming before anything else -----------------------
    f32x4 b_pre[kPF][4];
    // thin-chunk form (gnf_fused_dev.h) of a one-tile chunk: everywhere in the inference instances; the stash instance
    // only in the LAST layer (a hidden layer's epilogue there also leaves the act' ballots, which the thin form does not do).
    // The prefetch of a chunk packs its registers for the form that will consume it: both sides ask thin_for(layer).
    auto thin_for = [&](int layer) { return MT == 1 && (!STASH || layer == a.K - 1); };
    int* s_rowptr = tab + GNF_MAX_LAYERS * 8;
    int* s_col = s_rowptr + kRowptrPad;
    // STASH only: the act' ballots.  With the attention prologue they sit right behind the activation buffers, inside the
    // front-end's staging area (dead by the time the first layer's epilogue writes them)
    [[maybe_unused]] unsigned long long* fmask =
        FRONT ? reinterpret_cast<unsigned long long*>(smem + 2 * NETS * TM * LS) : reinterpret_cast<unsigned long long*>(s_col + kColCap);
    // FRONT: (scale, shift) of the bijector in front of this half-step [2][Hp], then of the previous one [2][Hp] (where the
    // message-passing prologue has its rowptr / col slices)
    [[maybe_unused]] float* bn_lds = reinterpret_cast<float*>(s_rowptr);
    if constexpr (FRONT) {
        // ---- attention prologue: the layer table first (its words sit behind the front-end's staging area), then the
        // front-end over this tile's 16 receiver rows; the first chunk's weights and the biases are requested between its
        // last barrier and its output projection, and the biases reach LDS behind it ------------------------------------
        for (int j = 0; j < a.K; ++j) {
            if (tid == 0) {
                const unsigned long long p0 = reinterpret_cast<unsigned long long>(a.wp[0][j]);
                const unsigned long long p1 = reinterpret_cast<unsigned long long>(a.wp[1][j]);
                int* row = tab + 8 * j;
                row[0] = a.ipg[j];
                row[1] = a.ont[j];
                row[2] = a.boff[j];
                row[3] = 0;
                row[4] = (int)(unsigned)p0;
                row[5] = (int)(unsigned)(p0 >> 32);
                row[6] = (int)(unsigned)p1;
                row[7] = (int)(unsigned)(p1 >> 32);
            }
        }
        // (the biases are requested before the front-end starts - eight registers per thread through it - so that the slot
        // between its last barrier and its output projection only issues the weight prefetch: with the sixteen 64-bit
        // bias addresses formed there the slot took 1.5 k cycles)
        constexpr int kBiasRegsF = 8;
        const int bias_all_f = NETS * a.bias_tot;
        float breg_f[kBiasRegsF];
#pragma unroll
        for (int q = 0; q < kBiasRegsF; ++q) {
            const int i = tid + q * kFusedThreads;
            const int ic = i < bias_all_f ? i : 0;
            breg_f[q] = ic < a.bias_tot ? a.bias[0][ic] : a.bias[1][ic - a.bias_tot];
        }
        attn_front_tile<true, 10, 10, 4, true, true, FIXED>(fa, smem, row0, buf(0, 0), buf(1, 0), LS,
                                                            [&] { prefetch_chunk(cur, WPN, voff, b_pre, thin_for(cur.layer)); },
                                                            bn_lds);
        if (a.bnu_const) {  // the previous bijector's (scale, shift), for the coupling stage below
            const int HPb = (H + 15) & ~15;
            for (int i = tid; i < 2 * H; i += kFusedThreads) bn_lds[2 * HPb + (i < H ? i : HPb + (i - H))] = a.bnu_const[i];
        } else if (a.bnu_inv[0]) {  // inverse pass: beta | gamma | sqrt(moving variance + eps) | moving mean
            const int HPb = (H + 15) & ~15;
            for (int i = tid; i < H; i += kFusedThreads) {
                bn_lds[i] = a.bnu_inv[1][i];
                bn_lds[HPb + i] = a.bnu_inv[0][i];
                bn_lds[2 * HPb + i] = sqrtf(a.bnu_inv[3][i] + a.bnu_inv_eps);
                bn_lds[3 * HPb + i] = a.bnu_inv[2][i];
            }
        }
#pragma unroll
        for (int q = 0; q < kBiasRegsF; ++q) {
            const int i = tid + q * kFusedThreads;
            if (i < bias_all_f) bias_lds[i] = breg_f[q];
        }
        for (int i = tid + kBiasRegsF * kFusedThreads; i < bias_all_f; i += kFusedThreads)
            bias_lds[i] = i < a.bias_tot ? a.bias[0][i] : a.bias[1][i - a.bias_tot];
        __syncthreads();
    } else {
    prefetch_chunk(cur, WPN, voff, b_pre, thin_for(cur.layer));
    // ---- every independent global read of the prologue is ISSUED before any is consumed: rowptr of
    // the tile, the biases (<= 8 floats per thread in registers), the layer table - one memory round
    // trip instead of three back-to-back ones ----------------------------------------------------------
    int rp_reg = 0;
    if (tid <= TM) {
        const int r = row0 + tid;
        rp_reg = a.rowptr[r < a.n_nodes ? r : a.n_nodes];
    }
    constexpr int kBiasRegs = 8;
    const int bias_all = NETS * a.bias_tot;
    const float* bsrc0 = a.bias[net0];
    const float* bsrc1 = a.bias[NETS == 2 ? 1 : net0];
    float breg[kBiasRegs];
#pragma unroll
    for (int q = 0; q < kBiasRegs; ++q) {
        const int i = tid + q * kFusedThreads;
        const int ic = i < bias_all ? i : 0;  // clamped: the load is unconditional, the store is not
        breg[q] = ic < a.bias_tot ? bsrc0[ic] : bsrc1[ic - a.bias_tot];
    }
    for (int j = 0; j < a.K; ++j) {  // layer table (uniform loop: scalar loads of the kernel arguments)
        if (tid == 0) {
            const unsigned long long p0 = reinterpret_cast<unsigned long long>(a.wp[net0][j]);
            const unsigned long long p1 = reinterpret_cast<unsigned long long>(a.wp[NETS == 2 ? 1 : net0][j]);
            int* row = tab + 8 * j;
            row[0] = a.ipg[j];
            row[1] = a.ont[j];
            row[2] = a.boff[j];
            row[3] = 0;
            row[4] = (int)(unsigned)p0;
            row[5] = (int)(unsigned)(p0 >> 32);
            row[6] = (int)(unsigned)p1;
            row[7] = (int)(unsigned)(p1 >> 32);
        }
    }
    if (tid <= TM) s_rowptr[tid] = rp_reg;
#pragma unroll
    for (int q = 0; q < kBiasRegs; ++q) {
        const int i = tid + q * kFusedThreads;
        if (i < bias_all) bias_lds[i] = breg[q];
    }
    for (int i = tid + kBiasRegs * kFusedThreads; i < bias_all; i += kFusedThreads)  // very wide nets only
        bias_lds[i] = i < a.bias_tot ? bsrc0[i] : bsrc1[i - a.bias_tot];
    // ---- A: aggregate + combine into the layer-0 input of each net (tile_aggregate, gnf_fused_dev.h) ----
    __syncthreads();
    if (a.h0[0] != nullptr) {
        // attention GNNs: the layer-0 input of each net was produced by the attention front-end
        const int in0p = a.ipg[0] * 16;
        const bool v4 = (a.in0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.h0[net0]) | reinterpret_cast<uintptr_t>(a.h0[NETS == 2 ? 1 : net0])) & 15) == 0;
        if (v4) {  // 16 bytes per lane, both nets' rows requested together (LS and in0p are multiples of 4)
            const int q4 = in0p >> 2;
            for (int idx = tid; idx < TM * q4; idx += kFusedThreads) {
                const int rl = idx / q4, c = (idx - rl * q4) * 4;
                const int r = row0 + rl;
                const bool live = r < a.n_nodes && c < a.in0;
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 v0 = live ? *reinterpret_cast<const f32x4*>(a.h0[net0] + (int64_t)r * a.in0 + c) : z4;
                f32x4 v1 = z4;
                if (NETS == 2 && live) v1 = *reinterpret_cast<const f32x4*>(a.h0[1] + (int64_t)r * a.in0 + c);
                *reinterpret_cast<f32x4*>(buf(0, 0) + rl * LS + c) = v0;
                if (NETS == 2) *reinterpret_cast<f32x4*>(buf(1, 0) + rl * LS + c) = v1;
            }
        } else
        for (int idx = tid; idx < TM * in0p; idx += kFusedThreads) {
            const int rl = idx / in0p, c = idx - rl * in0p;
            const int r = row0 + rl;
            const bool live = r < a.n_nodes && c < a.in0;
            buf(0, 0)[rl * LS + c] = live ? a.h0[net0][(int64_t)r * a.in0 + c] : 0.f;
            if (NETS == 2) buf(1, 0)[rl * LS + c] = live ? a.h0[1][(int64_t)r * a.in0 + c] : 0.f;
        }
    } else {
        const TileAgg ta{a.col, a.x_cond, a.ld, a.n_nodes, row0, H, a.in0, a.ipg[0] * 16, a.mean, a.concat, a.eps, a.cond_copy};
        tile_aggregate<TM, kFusedThreads, kColCap>(ta, s_rowptr, s_col, buf(0, 0), NETS == 2 ? buf(1, 0) : nullptr, LS,
                                                   STASH ? a.stash_h0 : nullptr, tid);
    }  // message-passing prologue
    __syncthreads();
    }  // !FRONT

    // ---- B: K layers ------------------------------------------------------------------------------
    int pp = 0;
    for (int j = 0; j < a.K; ++j) {
        const float* in_lds = buf(nl, pp);
        float* out_lds = buf(nl, pp ^ 1);
        const float slope = (j == a.K - 1) ? 1.f : (a.act == GNF_ACT_RELU ? 0.f : a.alpha);
        while (cur.layer == j) {  // wave-uniform
            const WChunk c = cur;
            const WChunk nxt = next_chunk(c);
            const WChunk nx = nxt.layer < a.K ? nxt : c;  // no next chunk: harmless re-load
            const float* bl = bias_lds + nl * a.bias_tot + c.boff;
            constexpr bool kThin = MT == 1;  // the thin-chunk form keeps 32 fragments in registers: one M-tile only
            const bool thin_ok = kThin && thin_for(j), thin_nx = kThin && thin_for(nx.layer);
            // STASH: a hidden layer's epilogue also leaves the ballot of "activation > 0" (act' for the backward pass) in
            // LDS, word [net][layer][4 m + r][column tile] - the recompute rows of the backward kernel write the same words
            constexpr int EPI = STASH ? EPI_EX : EPI_PLAIN;
            EpiArgs ea{};
            if constexpr (STASH) {
                ea.mode = 0;
                ea.mask = j < a.K - 1 ? fmask + (size_t)(nl * (a.K - 1) + j) * (MT * 4) * a.stash_mld : nullptr;
                ea.mld = a.stash_mld;
            }
            if (c.nv >= 4)
                mlp_chunk<MT, 4, EPI>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea, thin_nx);
            else if (c.nv == 3)
                mlp_chunk<MT, 3, EPI>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea, thin_nx);
            else if (c.nv == 2)
                mlp_chunk<MT, 2, EPI>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea, thin_nx);
            else if (kThin && thin_ok && chunk_is_thin(c))
                mlp_chunk_thin(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre);
            else
                mlp_chunk<MT, 1, EPI>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea, thin_nx);
            cur = nxt;
        }
        if constexpr (STASH && NETS == 2) {
            // this layer's INPUT rows of both nets (the previous layer's outputs) go to the stash here, behind the
            // wave's own MFMA work and in front of the layer barrier: the buffer is read-only until the layer after
            // this one writes it, and a wave's copy overlaps the MFMAs of the wave it shares its SIMD with (all waves
            // copying right behind the barrier cost 8 us per launch).  s and t leave through the coupling epilogue.
            if (j >= 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    tile_dump<TM, kFusedThreads, true>(buf(q, pp), LS, a.stash_act[q][j], (int64_t)a.stash_ld, a.stash_w[j - 1], row0,
                                                       a.n_nodes, tid);
            }
        }
        pp ^= 1;
        __syncthreads();
    }

    if constexpr (STASH) {  // (every layer's barrier has passed: the words are complete)
        const int total_words = 2 * (a.K - 1) * (MT * 4) * a.stash_mld;
        unsigned long long* gm = a.stash_mask + (size_t)tile * total_words;
        for (int i = tid; i < total_words; i += kFusedThreads) gm[i] = fmask[i];
    }
    if (NETS == 1) {
        // ---- hand the s (or t) tile to the coupling kernel through the global scratch ------------
        const float* o_lds = buf(0, pp);
        float* dst = a.st_out[net0];
        for (int idx = tid; idx < TM * H; idx += kFusedThreads) {
            const int rl = idx / H, f = idx - rl * H;
            const int r = row0 + rl;
            if (r < a.n_nodes)
                dst[(int64_t)r * H + f] = o_lds[rl * LS + f] + (a.residual ? a.x_cond[(int64_t)r * a.ld + f] : 0.f);
        }
    } else {
        // ---- C: coupling update + block-reduced sum(s) -----------------------------------------
        const float* s_lds = buf(0, pp);
        const float* t_lds = buf(1, pp);
        double local = 0.0, local2 = 0.0;
        float* xn_lds = buf(0, pp ^ 1);  // (free by now) the updated rows, for the column sums below
        for (int idx = tid; idx < TM * H; idx += kFusedThreads) {
            const int rl = idx / H, f = idx - rl * H;
            const int r = row0 + rl;
            if (a.bn_part) xn_lds[rl * LS + f] = 0.f;
            if (r < a.n_nodes) {
                float sv = s_lds[rl * LS + f], tv = t_lds[rl * LS + f];
                [[maybe_unused]] const int HPb = (H + 15) & ~15;
                if (a.residual) {
                    float xr = a.x_cond[(int64_t)r * a.ld + f];
                    if constexpr (FRONT) {
                        if (fa.bn_part) xr = xr * bn_lds[f] + bn_lds[HPb + f];  // (the conditioning rows in memory are raw)
                    }
                    sv += xr;
                    tv += xr;
                }
                float xv = a.x_upd_src[(int64_t)r * a.ld + f];
                if constexpr (FRONT) {
                    if (a.bnu_const) xv = xv * bn_lds[2 * HPb + f] + bn_lds[3 * HPb + f];  // the previous half-step's bijector, deferred
                    else if (a.bnu_inv[0]) xv = (xv - bn_lds[f]) / bn_lds[HPb + f] * bn_lds[2 * HPb + f] + bn_lds[3 * HPb + f];
                }
                const float xn = a.inverse ? (xv - tv) * expf(-sv) : xv * expf(sv) + tv;
                a.x_upd[(int64_t)r * a.ld + f] = xn;
                local += (double)sv;
                local2 += (double)xn * (double)xn;
                if (a.bn_part) xn_lds[rl * LS + f] = xn;
                if constexpr (STASH) {  // (what the coupling used: a residual block's x_cond is already in)
                    a.stash_st[0][(int64_t)r * H + f] = sv;
                    a.stash_st[1][(int64_t)r * H + f] = tv;
                }
            }
        }
        for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
        if (lane == 0) red[wave] = local;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int w = 0; w < kFusedThreads / 64; ++w) tot += red[w];
            a.partials[tile] = tot;
        }
        if (a.bn_part) {  // (workgroup-uniform) column sums over the tile's rows, fixed order
            __syncthreads();
            for (int i = tid; i < 2 * H; i += kFusedThreads) {
                const int f = i >> 1, which = i & 1;
                double acc = 0.0;
                for (int rl = 0; rl < TM; ++rl) {
                    const double v = (double)xn_lds[rl * LS + f];
                    acc += which ? v * v : v;
                }
                a.bn_part[((int64_t)tile * H + f) * 2 + which] = acc;
            }
        }
        if (a.sq_partials) {  // (workgroup-uniform) the same fixed-order reduction for sum(x_new^2)
            for (int off = 32; off > 0; off >>= 1) local2 += __shfl_down(local2, off, 64);
            __syncthreads();
            if (lane == 0) red[wave] = local2;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
                for (int w = 0; w < kFusedThreads / 64; ++w) tot += red[w];
                a.sq_partials[tile] = tot;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
static int max_padded_width(const GnfMlp* m) {
    int w = 16;
    for (int j = 0; j <= m->num_layers; ++j) w = w > pad16(m->dims[j]) ? w : pad16(m->dims[j]);
    return w;
}

static int bias_total(const GnfMlp* m) {
    int t = 0;
    for (int j = 0; j < m->num_layers; ++j) t += pad16(m->dims[j + 1]);
    return t;
}

static size_t fused_lds_bytes(const GnfMlp* m, int MT, int NETS) {
    const int LS = max_padded_width(m) + 4;
    return (size_t)(2 * NETS * 16 * MT * LS + NETS * bias_total(m) + 2) * sizeof(float) + 8 * sizeof(double) +
           (GNF_MAX_LAYERS * 8 + kRowptrPad + kColCap) * sizeof(int);
}

// does any shape of the fused forward kernel hold this MLP's activations in LDS? (else its packed copy is never read)
bool fused_fits_lds(const GnfMlp* m) { return fused_lds_bytes(m, 1, 1) <= (size_t)kLdsLimit; }

static void choose_shape(const HalfStep& hs, int* mt, int* nets);
static int choose_big(const HalfStep& hs);

bool fused_supported(const HalfStep& hs) {
    const GnfMlp *s = hs.s_net, *t = hs.t_net;
    if (!s->packed || !t->packed) return false;
    if (s->num_layers != t->num_layers) return false;
    for (int j = 0; j <= s->num_layers; ++j)
        if (s->dims[j] != t->dims[j]) return false;
    return fused_lds_bytes(s, 1, 1) <= (size_t)kLdsLimit;
}

// may this half-step run out of place (HalfStep.x_upd_src / cond_copy)?  Only the both-nets-per-workgroup kernel
// gathers, copies and updates in one launch; attention nets take their layer-0 input from the front-end kernels.
bool fused_supports_oop(const HalfStep& hs) {
    if (!fused_supported(hs) || hs.s_net->attn) return false;
    if (choose_big(hs)) return true;
    int mt, nets;
    choose_shape(hs, &mt, &nets);
    return nets == 2;
}

// (MT, NETS) choice, from measurements on MI355X at L=256, K=5 (profiles/):
//   (1,2) 16 nodes x both nets : 37.5 us per launch, any tile count <= 256 (one tile per CU)
//   (2,2) 32 nodes x both nets : 64 us per launch = 32 us per 16 nodes -> wins once there is more than
//                                one 16-node tile per CU
//   (*,1) one net per workgroup + k_coupling: never faster at these widths, but its LDS footprint is
//         half, which keeps layers up to 1024 wide on the fused path.
// gnf_set_option("force_shape", <MT><NETS>) (e.g. 21) is a developer override for A/B runs.
static void choose_shape(const HalfStep& hs, int* mt, int* nets) {
    const GnfMlp* s = hs.s_net;
    const int64_t tiles16 = (hs.n_nodes + 15) / 16;
    auto fits = [&](int m, int n) { return fused_lds_bytes(s, m, n) <= (size_t)kLdsLimit; };
    int m = 1, n = 1;
    const int64_t cus = big_cu_count();
    if (tiles16 > cus && fits(2, 2)) {
        m = 2, n = 2;
    } else if (fits(1, 2)) {
        m = 1, n = 2;
    } else if (tiles16 > cus && fits(2, 1)) {
        m = 2, n = 1;
    }
    if (const int64_t force = opt(OPT_FORCE_SHAPE)) {
        const int fm = (int)(force / 10), fn = (int)(force % 10);
        if ((fm == 1 || fm == 2) && (fn == 1 || fn == 2) && fits(fm, fn)) m = fm, n = fn;
    }
    // blocks that end in snt.LayerNorm (gnn.py:550-552): the normalisation needs whole rows of s and t before the
    // coupling, which the one-net-per-workgroup shape already hands over through the global scratch
    if (s->attn && s->attn->layer_norm) n = 1;
    *mt = m;
    *nets = n;
}

// The large-batch form (gnf_fused_big.hip: 4-wave workgroups of up to 4 row tiles, the two nets one after the other,
// activations in place, two workgroups per CU): 0 = not this launch, else the row-tile cap.
// A launch's time moves in quanta - one 32-row tile per CU for the both-nets shape, two 4-tile workgroups per CU here -
// so the choice is by the quanta each needs, in units of the (2,2) shape's pass over 256 tiles (59 us at L = 256, K = 5):
//   (2,2):  ceil(g / (2 C))                                       g = 16-row granules, C = CUs
//   here :  g <= 8 C (even deal, one pass):  1.22 + 0.525 g / (2 C)
//           else: 3.36 per whole double round (8 C granules) + 0.15 (the aggregation launch) + the closing round
//                 (big_plan) by the row tiles left per CU:  <= 1: 0.8 | <= 2: 1.25 | <= 3: 1.7 | <= 4: 2.18 | <= 5: 2.4 |
//                 <= 6: 2.6 | <= 7: 3.0 | 8: another double round
// fitted to tools/ab_shapes.sh / tools/ab_tail.sh on config-4 batches of 3.5 k .. 78 k nodes: e.g. 8.2 k
// nodes 101 vs 117 us, 13 k 127 vs 119, 20 k 149 vs 177, 30 k 221 vs 237, 49 k 326 vs 352, 59 k 422 vs 471, 78 k 520 vs 588.
// gnf_set_option("force_shape", 40 | 30 | 20 | 10) forces it with that cap, any other forced shape keeps it off.
static int choose_big(const HalfStep& hs) {
    const GnfMlp* s = hs.s_net;
    if (!big_supported(s, hs.H) || hs.mlp_stash) return 0;
    if (hs.n_nodes * hs.ld >= (int64_t(1) << 31)) return 0;  // (its coupling loop indexes the rows with 32-bit offsets)
    if (s->attn && s->attn->layer_norm) return 0;  // (whole rows of s and t before the coupling: one-net-per-workgroup shape)
    if (const int64_t force = opt(OPT_FORCE_SHAPE)) {
        if (force == kForceBigLostPartner) return 4;  // (fault injection for the tests: launch_half_fused withholds the hand-over flag)
        return (force % 10 == 0 && force >= 10 && force <= 40) ? (int)(force / 10) : 0;
    }
    const int64_t g = (hs.n_nodes + 15) / 16, c = big_cu_count();
    if (g <= 2 * c) return 0;  // one tile per CU or less: the 16- / 32-row both-nets shapes
    const double old_q = (double)((g + 2 * c - 1) / (2 * c));
    double big_q;
    if (g <= 8 * c) {
        big_q = 1.22 + 0.525 * (double)g / (double)(2 * c);
    } else {
        static const double closing[8] = {0.0, 0.8, 1.25, 1.7, 2.18, 2.4, 2.6, 3.0};
        const int64_t full = g / (8 * c), left = g - full * 8 * c, t = (left + c - 1) / c;  // row tiles left per CU, rounded up
        big_q = 3.36 * (double)full + 0.15 + (t <= 7 ? closing[t] : 3.36);
    }
    return big_q < old_q ? 4 : 0;
}

template <int MT, int NETS, bool STASH = false, bool FRONT = false, bool FIXED = false>
static int launch_shape(const FusedArgs& a, unsigned grid, size_t lds, hipStream_t st, const FrontArgs* fa = nullptr) {
    GNF_ONCE_PER_DEVICE(GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_fused<MT, NETS, STASH, FRONT, FIXED>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit)));
    FrontArgs none;
    memset(&none, 0, sizeof(none));
    hipLaunchKernelGGL((k_half_fused<MT, NETS, STASH, FRONT, FIXED>), dim3(grid), dim3(kFusedThreads), lds, st, a, fa ? *fa : none);
    GNF_LAUNCH_CHECK("k_half_fused");
    return GNF_OK;
}

static bool front_fold_ok(const HalfStep& hs, FrontArgs* fa);
bool fused_bn_on_load_ok(const HalfStep& hs) {
    if (!hs.s_net->attn || hs.H > 128 || !fused_supported(hs)) return false;
    int MT, NETS;
    choose_shape(hs, &MT, &NETS);
    FrontArgs fa;
    memset(&fa, 0, sizeof(fa));
    return MT == 1 && NETS == 2 && !choose_big(hs) && front_fold_ok(hs, &fa);
}

// LDS of the attention instance: the front-end's staging area (or the activation buffers, whichever is larger), then
// bias | reduction scratch | layer table (no rowptr / col slices of its own: the front-end has them in its area)
static size_t fused_front_lds_bytes(const GnfMlp* m, const FrontDims& d) {
    const int LS = max_padded_width(m) + 4;
    const size_t act = (size_t)2 * 2 * 16 * LS, fr = (size_t)front_lds(d).total;
    return ((act > fr ? act : fr) + 2 * (size_t)bias_total(m) + 2) * sizeof(float) + 8 * sizeof(double) + (GNF_MAX_LAYERS * 8) * sizeof(int) +
           (size_t)4 * d.Hp * sizeof(float);  // (scale, shift) of two batch-norm bijectors
}

// May the attention front-end run as the fused kernel's prologue (k_half_fused<1, 2, false, true>)?  The sparse-batch
// front-end with the reference's head geometry (its register-resident instance), widths that need no zero padding in
// the layer-0 rows, and the two areas in 160 KB; the training forward (q | k | v, the attended values and h0 also go to the
// stash) with the drivers' default geometry only.
static bool front_fold_ok(const HalfStep& hs, FrontArgs* fa) {
    const GnfMlp *s = hs.s_net, *t = hs.t_net;
    const GnfAttn *a0 = s->attn, *a1 = t->attn;
    if (!a0 || !a1) return false;
    if (!hs.attn_packed[0] || !hs.attn_packed[1]) return false;
    if (!(hs.n_edges > 0 && hs.n_edges < 24 * hs.n_nodes) || !attn_front_fused_ok(a0, hs.H)) return false;
    if (a1->num_heads != a0->num_heads || a1->kq_dim != a0->kq_dim || a1->v_dim != a0->v_dim || a1->out_dim != a0->out_dim ||
        a1->concat != a0->concat || a1->kq_dim_division != a0->kq_dim_division)
        return false;  // (launch_attn_front reports it)
    if (a0->layer_norm || opt(OPT_ATTN_KERNEL)) return false;
    const FrontDims d = front_dims(hs.H, a0->num_heads, a0->kq_dim, a0->v_dim, a0->out_dim);
    if (!((d.PW >> 4) <= 6 && (d.Hp >> 4) <= 2 && d.kq == 10 && d.vd == 10)) return false;
    const int in0 = s->dims[0];
    if ((d.C & 15) || (a0->concat && (hs.H & 15)) || (in0 & 15) || in0 != (a0->concat ? hs.H : 0) + d.C) return false;
    if (fused_front_lds_bytes(s, d) > (size_t)kLdsLimit) return false;
    if (hs.mlp_stash) {  // (the stash instance exists for the default geometry only; its act' ballots sit behind the activation buffers)
        const MlpStashLayout SL = mlp_stash_layout(s, hs.n_nodes, hs.H);
        const size_t act_bytes = (size_t)2 * 2 * 16 * (max_padded_width(s) + 4) * sizeof(float);
        if (!attn_front_fixed_geometry(d) || act_bytes + (size_t)SL.mask_words * 8 > (size_t)front_lds(d).total * sizeof(float)) return false;
    }
    for (int q = 0; q < 2; ++q) {
        fa->packed[q] = hs.attn_packed[q];
        fa->qkv[q] = nullptr;
        fa->h0[q] = nullptr;
        fa->agg_out[q] = nullptr;
        fa->mz_out[q] = nullptr;
    }
    if (hs.attn_region) {  // training forward: q | k | v, h0, the attended values and the softmax statistics stay in the
                           // half-step's slot of GnfFlow.attn_stash (attn_scratch_floats' layout, as launch_attn_pair)
        const size_t n = (size_t)hs.n_nodes, P = 2 * (size_t)a0->num_heads * a0->kq_dim + a0->v_dim, NV = (size_t)a0->num_heads * a0->v_dim;
        float* r = hs.attn_region;
        fa->qkv[0] = r, fa->qkv[1] = r + n * P;
        fa->h0[0] = r + 2 * n * P, fa->h0[1] = fa->h0[0] + n * in0;
        fa->agg_out[0] = fa->h0[1] + n * in0, fa->agg_out[1] = fa->agg_out[0] + n * NV;
        fa->mz_out[0] = fa->agg_out[1] + n * NV, fa->mz_out[1] = fa->mz_out[0] + n * 3 * a0->num_heads;
    }
    fa->rowptr = hs.rowptr, fa->col = hs.col, fa->x = hs.x_cond, fa->ldx = hs.ld;
    fa->tiles = hs.attn_tiles;
    fa->n_nodes = (int32_t)hs.n_nodes;
    fa->concat = a0->concat ? 1 : 0;
    fa->in0 = in0;
    fa->d = d;
    fa->scale = a0->kq_dim_division ? 1.f / sqrtf((float)a0->kq_dim) : 1.f;
    return true;
}

// One half-step's slot of GnfFlow.mlp_stash (float offsets, every region 256-byte aligned)
MlpStashLayout mlp_stash_layout(const GnfMlp* net, int64_t n, int32_t H) {
    auto al = [](size_t v) { return (v + 63) / 64 * 64; };
    MlpStashLayout L;
    const int K = net->num_layers;
    int lmax = 1;
    for (int j = 1; j < K; ++j) lmax = lmax > net->dims[j] ? lmax : net->dims[j];
    L.ld_act = lmax;
    size_t off = 0;
    L.h0 = off, off += al((size_t)n * net->dims[0]);
    L.act_each = al((size_t)n * lmax);
    L.act = off, off += 2 * (size_t)(K > 1 ? K - 1 : 0) * L.act_each;
    L.st_each = al((size_t)n * H);
    L.st = off, off += 2 * L.st_each;
    L.mld = 1;
    for (int j = 1; j < K; ++j) L.mld = L.mld > pad16(net->dims[j]) / 16 ? L.mld : pad16(net->dims[j]) / 16;
    L.mask_words = 2 * (K > 1 ? K - 1 : 0) * 4 * L.mld;
    L.mask = off, off += al((size_t)((n + 15) / 16) * L.mask_words * 2);
    L.slot = off;
    return L;
}

// the forward half of mlp_stash_supported: the (1,2) shape is what the launch would pick
bool fused_stash_shape(const GnfMlp* s, const GnfMlp* t, int64_t n) {
    if (!s->packed || !t->packed || s->num_layers != t->num_layers) return false;
    if (s->attn && s->attn->layer_norm) return false;  // (one net per workgroup there: s, t leave through the global scratch)
    for (int j = 0; j <= s->num_layers; ++j)
        if (s->dims[j] != t->dims[j]) return false;
    if (opt(OPT_FORCE_SHAPE)) return false;
    return (n + 15) / 16 <= big_cu_count() && fused_lds_bytes(s, 1, 2) <= (size_t)kLdsLimit;
}

int launch_half_fused(const HalfStep& hs, float* scratch, hipStream_t st) {
    const GnfMlp *s = hs.s_net, *t = hs.t_net;
    *hs.n_partials = 0;
    if (hs.n_nodes == 0) return GNF_OK;
    int MT, NETS;
    choose_shape(hs, &MT, &NETS);
    FusedArgs a;
    a.rowptr = hs.rowptr;
    a.col = hs.col;
    a.x_cond = hs.x_cond;
    a.x_upd = hs.x_upd;
    memset(a.big_seg_n, 0, sizeof(a.big_seg_n));
    memset(a.big_seg_sz, 0, sizeof(a.big_seg_sz));
    memset(a.big_seg_kind, 0, sizeof(a.big_seg_kind));
    a.big_xg0 = 0, a.big_epoch = 0, a.big_epoch_set = 0, a.big_split_s = nullptr, a.big_split_flag = nullptr;
    a.x_upd_src = hs.x_upd_src ? hs.x_upd_src : hs.x_upd;
    a.cond_copy = hs.cond_copy;
    a.partials = hs.partials;
    a.sq_partials = NETS == 2 ? hs.sq_partials : nullptr;
    a.bn_part = NETS == 2 ? hs.bn_part : nullptr;
    // NETS = 1 scratch: s [N,H] | t [N,H] at the head of the float scratch
    a.st_out[0] = scratch;
    a.st_out[1] = scratch + hs.n_nodes * hs.H;
    a.h0[0] = a.h0[1] = nullptr;
    a.residual = 0;
    a.stash_h0 = nullptr;
    memset(a.stash_act, 0, sizeof(a.stash_act));
    a.stash_st[0] = a.stash_st[1] = nullptr;
    memset(a.stash_w, 0, sizeof(a.stash_w));
    a.stash_ld = 0;
    a.stash_mask = nullptr;
    a.stash_mld = 0;
    a.bnu_const = nullptr;
    a.bnu_inv[0] = a.bnu_inv[1] = a.bnu_inv[2] = a.bnu_inv[3] = nullptr;
    a.bnu_inv_eps = 0.f;
    FrontArgs fa;
    memset(&fa, 0, sizeof(fa));
    const bool fold = s->attn && MT == 1 && NETS == 2 && !choose_big(hs) && front_fold_ok(hs, &fa);
    if (hs.bnu_inv) {
        if (!fold || hs.direction != GNF_INVERSE) {
            set_error("internal: batch norm on load (inverse) handed to a half-step that does not run the fused attention instance");
            return GNF_EINVAL;
        }
        a.bnu_inv[0] = hs.bnu_inv->gamma, a.bnu_inv[1] = hs.bnu_inv->beta;
        a.bnu_inv[2] = hs.bnu_inv->moving_mean, a.bnu_inv[3] = hs.bnu_inv->moving_variance;
        a.bnu_inv_eps = hs.bnu_inv->epsilon;
    }
    if (hs.bnc) {  // the bijector on load (the caller asked fused_bn_on_load_ok first)
        if (!fold || hs.direction != GNF_FORWARD) {
            set_error("internal: batch norm on load handed to a half-step that does not run the fused attention instance");
            return GNF_EINVAL;
        }
        fa.bn_part = hs.bnc_part, fa.bn_nparts = hs.bnc_nparts;
        fa.bn_gamma = hs.bnc->gamma, fa.bn_beta = hs.bnc->beta, fa.bn_eps = hs.bnc->epsilon;
        fa.bn_mean_out = hs.bnc->batch_mean, fa.bn_var_out = hs.bnc->batch_variance;
        fa.bn_logdet_out = hs.bnc_logdet, fa.bn_const_out = hs.bnc_const;
        a.bnu_const = hs.bnu_const;
    }
    if (s->attn) {
        if (!fold) {
            float* h0_pair[2];
            const int rc0 = launch_attn_pair(hs, scratch, h0_pair, st);
            if (rc0) return rc0;
            a.h0[0] = h0_pair[0];
            a.h0[1] = h0_pair[1];
        }
        a.residual = s->attn->residual ? 1 : 0;
    }
    int64_t off = 0;
    int boff = 0;
    for (int j = 0; j < s->num_layers; ++j) {
        const int ip = pad16(s->dims[j]), op = pad16(s->dims[j + 1]);
        a.wp[0][j] = s->packed + off;
        a.wp[1][j] = t->packed + off;
        a.ipg[j] = ip / 16;
        a.ont[j] = op / 16;
        a.boff[j] = boff;
        boff += op;
        off += (int64_t)ip * op;
    }
    a.bias[0] = s->packed + off;  // contiguous bias block after the weights
    a.bias[1] = t->packed + off;
    a.bias_tot = boff;
    a.ld = hs.ld;
    a.n_nodes = (int32_t)hs.n_nodes;
    a.H = hs.H;
    a.in0 = s->dims[0];
    a.K = s->num_layers;
    a.LS = max_padded_width(s) + 4;
    a.mean = hs.gnn.agg == GNF_AGG_MEAN;
    a.concat = hs.gnn.combine == GNF_COMBINE_CONCAT;
    a.act = hs.gnn.activation;
    a.inverse = hs.direction == GNF_INVERSE;
    a.eps = hs.gnn.epsilon;
    a.alpha = hs.gnn.alpha;

    int rc;
    if (const int big = choose_big(hs)) {
        // bn_part stays unwritten (*n_bn = 0: the bijector in front of the next half-step runs its own moment pass)
        a.sq_partials = hs.sq_partials;
        a.bn_part = nullptr;
        if (!s->attn) {
            // message-passing GNNs: the layer-0 input rows of both nets (eps * x + agg | [x || agg], gnn.py:108-109,123)
            // from the standalone aggregation kernel into the head of the scratch; the large-batch kernel reads them like an
            // attention front-end's output.  Rows of up to 32 edges are summed in edge order with the fused kernels' fma
            // (bitwise their gather); a longer row (an ego hub) is added up as contiguous segments in a fixed order
            // (gnf_layered.hip), which differs from the sequential sum - and from what the fused inverse / backward kernels
            // recompute for that row - in rounding only (the round-trip and gradient pins of tests/test_fullsize_gpu.py,
            // config 5, cover batches whose hub rows cross this boundary)
            rc = launch_aggregate(hs.rowptr, hs.col, hs.n_nodes, hs.x_cond, hs.ld, hs.H, a.mean, a.concat ? 1 : 0, a.eps, scratch,
                                  a.in0, st);
            if (rc) return rc;
            a.h0[0] = a.h0[1] = scratch;
            // split row tiles: flags and s rows behind the layer-0 rows (the caller zeroed the flags: split_epoch > 0)
            int lmax = 1;
            for (int j = 1; j < s->num_layers; ++j) lmax = lmax > s->dims[j] ? lmax : s->dims[j];
            const size_t off = big_split_offset(hs.n_nodes, a.in0);
            const size_t room = (size_t)hs.n_nodes * (size_t)(a.in0 + kLayeredActBufs * lmax + 2 * hs.H);
            if (hs.split_epoch > 0 && big_cu_count() / 2 <= kBigSplitMax && off + big_split_floats() <= room) {
                a.big_split_flag = reinterpret_cast<int*>(scratch + off);
                a.big_split_s = scratch + off + kBigSplitMax;
                a.big_epoch = hs.split_epoch;
                // force_shape = 49: the s-net workgroups of split tiles publish a value the t-net workgroups do not wait for -
                // every split tile's partner is "lost" (the only way to execute that branch: in a healthy launch the flag is
                // there half a launch before anybody looks at it)
                a.big_epoch_set = opt(OPT_FORCE_SHAPE) == kForceBigLostPartner ? -hs.split_epoch : hs.split_epoch;
            }
        }
        int n_wg = 0;
        rc = launch_half_big(a, hs.n_nodes, big, st, &n_wg);
        if (rc) return rc;
        *hs.n_partials = (int32_t)n_wg;
        if (hs.n_sq) *hs.n_sq = a.sq_partials ? (int32_t)n_wg : 0;
        if (hs.n_bn) *hs.n_bn = 0;
        return GNF_OK;
    }
    const int64_t tiles = (hs.n_nodes + 16 * MT - 1) / (16 * MT);
    a.n_tiles = (int32_t)tiles;
    const size_t lds = fused_lds_bytes(s, MT, NETS);
    if (hs.mlp_stash && !(MT == 1 && NETS == 2)) {
        set_error("internal: MLP-row stash on a launch shape other than (1,2) (mlp_stash_supported is false there)");
        return GNF_EINVAL;
    }
    if (hs.mlp_stash) {
        const MlpStashLayout L = mlp_stash_layout(s, hs.n_nodes, hs.H);
        a.stash_h0 = hs.mlp_stash + L.h0;
        for (int q = 0; q < 2; ++q) {
            for (int j = 1; j < s->num_layers; ++j) a.stash_act[q][j] = hs.mlp_stash + L.act + ((size_t)q * (s->num_layers - 1) + (j - 1)) * L.act_each;
            a.stash_st[q] = hs.mlp_stash + L.st + (size_t)q * L.st_each;
        }
        for (int j = 0; j < s->num_layers; ++j) a.stash_w[j] = s->dims[j + 1];
        a.stash_ld = L.ld_act;
        a.stash_mask = reinterpret_cast<unsigned long long*>(hs.mlp_stash + L.mask);
        a.stash_mld = L.mld;
        if (fold)
            rc = launch_shape<1, 2, true, true, true>(a, (unsigned)tiles, fused_front_lds_bytes(s, fa.d), st, &fa);
        else
            rc = launch_shape<1, 2, true>(a, (unsigned)tiles, lds + (size_t)L.mask_words * sizeof(unsigned long long), st);
        if (rc) return rc;
        *hs.n_partials = (int32_t)tiles;
        if (hs.n_sq) *hs.n_sq = a.sq_partials ? (int32_t)tiles : 0;
        if (hs.n_bn) *hs.n_bn = a.bn_part ? (int32_t)tiles : 0;
        return GNF_OK;
    }
    if (NETS == 2) {
        if (fold && attn_front_fixed_geometry(fa.d))
            rc = launch_shape<1, 2, false, true, true>(a, (unsigned)tiles, fused_front_lds_bytes(s, fa.d), st, &fa);
        else if (fold)
            rc = launch_shape<1, 2, false, true>(a, (unsigned)tiles, fused_front_lds_bytes(s, fa.d), st, &fa);
        else
            rc = MT == 2 ? launch_shape<2, 2>(a, (unsigned)tiles, lds, st) : launch_shape<1, 2>(a, (unsigned)tiles, lds, st);
        if (rc) return rc;
        *hs.n_partials = (int32_t)tiles;
        if (hs.n_sq) *hs.n_sq = a.sq_partials ? (int32_t)tiles : 0;
        if (hs.n_bn) *hs.n_bn = a.bn_part ? (int32_t)tiles : 0;
        return GNF_OK;
    }
    if (hs.x_upd_src || hs.cond_copy) {
        set_error("internal: out-of-place half-step on the one-net-per-workgroup shape (fused_supports_oop is false there)");
        return GNF_EINVAL;
    }
    const unsigned grid = (unsigned)(8 * ((tiles + 3) / 4));  // 4 tile slots x 2 nets per group of 8 blocks
    rc = MT == 2 ? launch_shape<2, 1>(a, grid, lds, st) : launch_shape<1, 1>(a, grid, lds, st);
    if (rc) return rc;
    if (s->attn && s->attn->layer_norm) {
        rc = launch_half_layer_norm(hs, a.st_out[0], a.st_out[1], nullptr, st);
        if (rc) return rc;
    }
    return launch_coupling(a.st_out[0], a.st_out[1], hs, nullptr, st);  // residual already added above
}

}  // namespace gnf
