// Device side of the fused backward half-step (see gnf_fused_bwd.hip for the description): the kernel body as a
// function, so that gnf_train.hip can put it into one launch with the weight-gradient GEMMs of the previous half-step.
#pragma once
#include "gnf_fused_dev.h"

namespace gnf {

static constexpr int kBwdThreads = 512;
static constexpr int kBwdLdsLimit = 160 * 1024;
static constexpr int kBwdRowptrPad = 40;
static constexpr int kBwdColCap = 2048;
static constexpr int kRows = 2 * GNF_MAX_LAYERS;

// table row: 0 ipg, 1 ont, 2 boff, 3 true output width, 4 mode (0 recompute, 1 backward), 5 mask slot (-1: none),
//            6 dump row stride, 7 first input column (floats, a multiple of 4), 8-9 packed weights net 0, 10-11 net 1,
//            12-13 dump pointer net 0, 14-15 net 1
// Rows 0 .. K-1: the layers forwards (recompute), K .. 2K-1: backwards; attention GNNs may add row 2K (bwd_args_add_dagg_row):
// dagg = dnew Wo^T, dnew = columns [off, off + C) of dL/dh0 - the first product of the attention backward pass, which
// used to be a GEMM launch of its own between this kernel and the edge kernels.
struct BwdArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* x_cond;
    float* y_upd;
    float* g_upd;
    float* h0_out;
    const float* h0_in[2];  // attention GNNs: the layer-0 input of each net comes from the attention front-end
    float* gst[2];
    const float* bias[2];
    int32_t tab[kRows][16];
    int64_t ld, ldg;
    int32_t n_nodes, n_tiles, H, in0, K, LS, bias_tot, bias_tot2, mld;
    int32_t n_rows;  // rows of the table: 2 K, or 2 K + 1
    int32_t mean, concat, act;
    int32_t residual;  // attention block with residual: s, t = MLP(h0) + x_cond (gnn.py:547-548)
    float eps, alpha;
    // Message-passing backward of the PREVIOUS half-step of the walk, folded into this launch's prologue (NULL rowptr_t:
    // not folded).  That half-step's dL/dh0 rows (both nets) scatter into the gradient of ITS conditioning half, which is
    // this half-step's updated half: g_upd[u, f] += base * dh[u, f] + sum over edges u -> v of dh[v, aggcol + f] * w(v),
    // dh = dh_prev[0] + dh_prev[1] ([N, in0] each), w(v) = invdeg[v] (mean aggregator) or 1 - the arithmetic of
    // k_aggregate_bwd (gnf_train.hip), applied where the coupling stage reads g_upd instead of in a launch of its own.
    const int32_t* rowptr_t;  // CSR by SENDER
    const int32_t* col_t;
    const float* invdeg;
    const float* dh_prev[2];
    int32_t fold_aggcol, fold_concat;
    // STASHED instance (GnfFlow.mlp_stash, filled by the training forward): s, t and the hidden activations of this
    // half-step, read instead of recomputed
    const float* st_in[2];                     // [N, H]
    const unsigned long long* mask_in;         // [tile][net][K-1][4][mld] act' ballot words (NULL: timing ablation)
    // Batch-norm bijector of the PREVIOUS half-step of the walk, undone and differentiated where the coupling stage
    // reads y_upd / g_upd (NULL bn_part: not folded; the arithmetic of k_bn_bwd_apply, gnf_bn_bwd.hip): every workgroup
    // adds up the bn_nparts partial rows (sum G, sum G x^ per feature, fp64) the kernel that finished g left, workgroup 0
    // also writes d gamma / d beta.  Saves that kernel's launch (5.3 us per half-step on the config-2 batch).
    const double* bn_part;
    const float* bn_gamma;
    const float* bn_beta;
    const float* bn_mean;
    const float* bn_var;
    float* bn_dgamma;
    float* bn_dbeta;
    int32_t bn_nparts;
    float bn_eps;
};
struct BwdStash {   // one half-step's rows of GnfFlow.mlp_stash as the backward kernel reads them
    const float* st_in[2];
    const unsigned long long* mask_in;
};
struct BwdFold {
    const int32_t* rowptr_t;
    const int32_t* col_t;
    const float* invdeg;
    const float* dh_prev[2];
};

// bid / nwg: this workgroup's index among the nwg backward workgroups of the launch (the launch may hold other work
// behind them: gnf_train.hip puts the previous half-step's weight-gradient GEMMs on the CUs a small batch leaves idle)
// STASHED: the forward pass left this half-step's rows in GnfFlow.mlp_stash - no aggregation, no recompute layers: the
// act' masks are rebuilt from the stashed activations, the coupling stage reads the stashed s and t, then the K backward
// layers run as usual (about half the matrix work of the recomputing form).
template <int MT, bool STASHED = false>  // 16 * MT nodes per workgroup: MT = 2 halves the weight stream per node on batches with more than
                    // one 16-node tile per CU (measured on the forward kernel: 64 us per 32 nodes vs 37.5 per 16)
__device__ __forceinline__ void half_bwd_body(const BwdArgs& a, const int bid, const int nwg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 16 * MT, WPN = 4;
    const int LS = a.LS;
    auto buf = [&](int net_, int pp_) -> float* { return smem + (2 * net_ + pp_) * TM * LS; };
    float* bias_lds = smem + 4 * TM * LS;
    int* tab = reinterpret_cast<int*>(bias_lds + 2 * a.bias_tot2);
    int* s_rowptr = tab + kRows * 16;
    int* s_col = s_rowptr + kBwdRowptrPad;
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(s_col + kBwdColCap);  // [net][K-1][MT*4][mld]
    const int HP = (a.H + 15) & ~15;
    int* f_rowptr = reinterpret_cast<int*>(masks + (size_t)2 * (a.K > 1 ? a.K - 1 : 0) * (MT * 4) * a.mld);
    int* f_col = f_rowptr + kBwdRowptrPad;
    float* f_own = reinterpret_cast<float*>(f_col + kBwdColCap);  // [TM][HP] base term of the folded scatter
    float* f_acc = f_own + TM * HP;                                // [TM][HP] its neighbour sum
    const bool fold = a.rowptr_t != nullptr;

    int tile;
    {
        const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
        tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    }
    const int row0 = tile * TM;
    const int tid = threadIdx.x;
    const int H = a.H;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nl = wave / WPN;
    const int wl = (wave % WPN + nl * (WPN / 2)) % WPN;  // t-net ownership rotated by half a turn (see gnf_fused.hip)
    const int voff = lane * 16;
    const int R = a.n_rows;  // rows of the layer table

    auto fill_chunk = [&](WChunk& c, int r, int ipg_, int ont_, int boff_, const float* wb, int nt0) {
        c.wbase = wb;
        c.wbytes = (unsigned)ipg_ * (unsigned)ont_ * 1024u;
        c.ipg = ipg_;
        c.ont = ont_;
        c.boff = boff_;
        c.nt0 = nt0;
        const int nv = (ont_ - nt0 + WPN - 1) / WPN;
        c.nv = nv > 4 ? 4 : nv;
        c.layer = r;
    };
    auto ptr_of = [&](const int* row, int slot) -> unsigned long long {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(row[slot]);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(row[slot + 1]);
        return ((unsigned long long)hi << 32) | lo;
    };
    auto chunk_from_tab = [&](int r, int nt0) -> WChunk {
        const int* row = tab + 16 * r;
        WChunk c;
        fill_chunk(c, r, __builtin_amdgcn_readfirstlane(row[0]), __builtin_amdgcn_readfirstlane(row[1]),
                   __builtin_amdgcn_readfirstlane(row[2]), reinterpret_cast<const float*>(ptr_of(row, 8 + 2 * nl)), nt0);
        return c;
    };
    auto next_chunk = [&](const WChunk& c) -> WChunk {
        int r = c.layer, nt0 = c.nt0 + 4 * WPN;
        if (nt0 < c.ont) {
            WChunk n = c;
            n.nt0 = nt0;
            const int nv = (c.ont - nt0 + WPN - 1) / WPN;
            n.nv = nv > 4 ? 4 : nv;
            return n;
        }
        for (++r; r < R; ++r) {
            const int ont_ = __builtin_amdgcn_readfirstlane(tab[16 * r + 1]);
            if (wl < ont_) return chunk_from_tab(r, wl);
        }
        WChunk n = c;
        n.layer = R;
        return n;
    };
    WChunk cur;
    {   // first chunk straight from the kernel arguments (the LDS table does not exist yet)
        int r0 = STASHED ? a.K : 0;
        while (r0 < R && wl >= a.tab[r0][1]) ++r0;
        const int rr = r0 < R ? r0 : 0;
        const unsigned long long wp = ((unsigned long long)(unsigned)a.tab[rr][9 + 2 * nl] << 32) |
                                      (unsigned)a.tab[rr][8 + 2 * nl];
        fill_chunk(cur, rr, a.tab[rr][0], a.tab[rr][1], a.tab[rr][2], reinterpret_cast<const float*>(wp),
                   r0 < R ? wl : 0);
        if (r0 >= R) cur.layer = R;
    }
    f32x4 b_pre[kPF][4];
    prefetch_chunk(cur, WPN, voff, b_pre);

    // ---- prologue loads, all issued before any is consumed ----------------------------------------
    int rp_reg = 0, rp2_reg = 0;
    if (tid <= TM) {
        const int r = row0 + tid;
        if (!STASHED) rp_reg = a.rowptr[r < a.n_nodes ? r : a.n_nodes];
        if (fold) rp2_reg = a.rowptr_t[r < a.n_nodes ? r : a.n_nodes];
    }
    constexpr int kBiasRegs = 8;
    const int bias_all = 2 * a.bias_tot2;
    float breg[kBiasRegs];
#pragma unroll
    for (int q = 0; q < kBiasRegs; ++q) {
        const int i = tid + q * kBwdThreads;
        const int net_ = i >= a.bias_tot2 ? 1 : 0;
        const int k = i - net_ * a.bias_tot2;
        const bool live = !STASHED && i < bias_all && k < a.bias_tot;  // (STASHED: only the zero bias of the backward rows is read)
        const float* src = net_ ? a.bias[1] : a.bias[0];
        breg[q] = live ? src[k] : 0.f;  // the tail of each net's block is the zero bias of the backward rows
    }
    if (tid < R * 16) tab[tid] = a.tab[tid >> 4][tid & 15];
    if (tid <= TM) s_rowptr[tid] = rp_reg;
    if (fold && tid <= TM) f_rowptr[tid] = rp2_reg;
#pragma unroll
    for (int q = 0; q < kBiasRegs; ++q) {
        const int i = tid + q * kBwdThreads;
        if (i < bias_all) bias_lds[i] = breg[q];
    }
    for (int i = tid + kBiasRegs * kBwdThreads; i < bias_all; i += kBwdThreads) {
        const int net_ = i >= a.bias_tot2 ? 1 : 0;
        const int k = i - net_ * a.bias_tot2;
        bias_lds[i] = (!STASHED && k < a.bias_tot) ? (net_ ? a.bias[1] : a.bias[0])[k] : 0.f;
    }
    __syncthreads();
    if constexpr (STASHED) {
        if (fold) {
            const int fb = f_rowptr[0], fl = f_rowptr[TM] - fb;
            if (fl <= kBwdColCap)
                for (int i = tid; i < fl; i += kBwdThreads) f_col[i] = a.col_t[fb + i];
        }
        // act' masks: the training forward kernel left this tile's ballot words (the very words the recompute rows
        // would write: [net][slot][4 m + r][column tile]) next to the activations - one coalesced 8-byte load per thread.
        // (Rebuilding them from the stashed activations - 64 scattered row-segment loads per wave - took 33 us per tile.)
        {
            const int total_words = 2 * (a.K - 1) * (MT * 4) * a.mld;
            const unsigned long long* __restrict__ gm = a.mask_in + (size_t)tile * total_words;
            if (a.mask_in)
                for (int i = tid; i < total_words; i += kBwdThreads) masks[i] = gm[i];
        }
        __syncthreads();  // the transposed CSR slice is in place (the masks are first read after the coupling stage's barrier)
    } else
    // ---- A: aggregate + combine (same arithmetic and order as the forward kernel) ----------------
    if (a.h0_in[0] != nullptr) {
        const int in0p = a.tab[0][0] * 16;
        if ((a.in0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.h0_in[0]) | reinterpret_cast<uintptr_t>(a.h0_in[1])) & 15) == 0) {
            const int q4 = in0p >> 2;  // 16 bytes per lane, both nets' rows requested together
            for (int idx = tid; idx < TM * q4; idx += kBwdThreads) {
                const int rl = idx / q4, c = (idx - rl * q4) * 4;
                const int r = row0 + rl;
                const bool live = r < a.n_nodes && c < a.in0;
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                const f32x4 v0 = live ? *reinterpret_cast<const f32x4*>(a.h0_in[0] + (int64_t)r * a.in0 + c) : z4;
                const f32x4 v1 = live ? *reinterpret_cast<const f32x4*>(a.h0_in[1] + (int64_t)r * a.in0 + c) : z4;
                *reinterpret_cast<f32x4*>(buf(0, 0) + rl * LS + c) = v0;
                *reinterpret_cast<f32x4*>(buf(1, 0) + rl * LS + c) = v1;
            }
        } else
        for (int idx = tid; idx < TM * in0p; idx += kBwdThreads) {
            const int rl = idx / in0p, c = idx - rl * in0p;
            const int r = row0 + rl;
            const bool live = r < a.n_nodes && c < a.in0;
            buf(0, 0)[rl * LS + c] = live ? a.h0_in[0][(int64_t)r * a.in0 + c] : 0.f;
            buf(1, 0)[rl * LS + c] = live ? a.h0_in[1][(int64_t)r * a.in0 + c] : 0.f;
        }
    } else {
        if (fold) {  // the transposed CSR slice rides behind the same barrier as the forward one
            const int fb = f_rowptr[0], fl = f_rowptr[TM] - fb;
            if (fl <= kBwdColCap)
                for (int i = tid; i < fl; i += kBwdThreads) f_col[i] = a.col_t[fb + i];
        }
        const TileAgg ta{a.col, a.x_cond, a.ld, a.n_nodes, row0, H, a.in0, a.tab[0][0] * 16, a.mean, a.concat, a.eps};
        tile_aggregate<TM, kBwdThreads, kBwdColCap>(ta, s_rowptr, s_col, buf(0, 0), buf(1, 0), LS, a.h0_out, tid);
    }
    if (fold) {
        // one (row, feature) per thread, four neighbour rows in flight, adds in edge order (k_aggregate_bwd's order)
        const int fb = f_rowptr[0];
        const bool staged = f_rowptr[TM] - fb <= kBwdColCap;
        const float* __restrict__ d0 = a.dh_prev[0];
        const float* __restrict__ d1 = a.dh_prev[1];
        for (int idx = tid; idx < TM * H; idx += kBwdThreads) {
            const int rl = idx / H, f = idx - rl * H;
            const int u = row0 + rl;
            float own = 0.f, acc = 0.f;
            if (u < a.n_nodes) {
                const int beg = f_rowptr[rl], end = f_rowptr[rl + 1];
                auto colat = [&](int e) { return staged ? f_col[e - fb] : a.col_t[e]; };
                auto ld2 = [&](int v) {
                    const int64_t o = (int64_t)v * a.in0 + a.fold_aggcol + f;
                    return d0[o] + d1[o];
                };
                int e = beg;
                for (; e + 4 <= end; e += 4) {
                    int vi[4];
                    float w[4], vv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) vi[q] = colat(e + q);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w[q] = a.invdeg ? a.invdeg[vi[q]] : 1.f;
                        vv[q] = ld2(vi[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc += vv[q] * w[q];
                }
                for (; e < end; ++e) {
                    const int v = colat(e);
                    acc += ld2(v) * (a.invdeg ? a.invdeg[v] : 1.f);
                }
                const int64_t oo = (int64_t)u * a.in0 + f;
                const float o2 = d0[oo] + d1[oo];
                own = a.fold_concat ? o2 : o2 * a.eps;
            }
            f_own[rl * HP + f] = own;
            f_acc[rl * HP + f] = acc;
        }
    }
    __syncthreads();

    // ---- one table row = one layer of one direction --------------------------------------------
    int pp = 0;
    // A row's outputs (h_{j+1} or dP_{j-1} of both nets) go to global memory for the dW GEMM as one coalesced 16-byte-per-lane
    // copy out of the LDS buffer the NEXT row reads (element-wise stores from the accumulator layout, 64-byte segments, made
    // this kernel store-bound on large batches) - and they go while that next row runs: every wave copies its share behind
    // its own MFMA work and in front of the row's barrier (the buffer is read-only until the row after writes it), where the
    // copy overlaps the MFMAs of the wave it shares its SIMD with.  All waves copying right behind the barrier cost 1.4 us
    // per hidden row (stamps, round 6: 9.4 us against the forward layer's 8.0; the forward kernel's stash copy has had this
    // place since round 4).  `pend`: the row whose outputs have not left yet; flushed behind the last row of a phase.
    int pend = -1;
    auto dump_row = [&](int rr) {
        const int* prow = tab + 16 * rr;
        const int width = __builtin_amdgcn_readfirstlane(prow[3]);
        const int64_t dld = __builtin_amdgcn_readfirstlane(prow[6]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float* dump = reinterpret_cast<float*>(ptr_of(prow, 12 + 2 * q));
            if (dump == nullptr) continue;
            tile_dump<TM, kBwdThreads, true>(buf(q, pp), LS, dump, dld, width, row0, a.n_nodes, tid);
        }
    };
    auto flush_rows = [&]() {
        if (pend >= 0) dump_row(pend);
        pend = -1;
    };
    auto run_row = [&](int r) {
        const int* row = tab + 16 * r;
        const int mode = __builtin_amdgcn_readfirstlane(row[4]);
        const int slot = __builtin_amdgcn_readfirstlane(row[5]);
        const bool last_fwd = (r == a.K - 1);
        const float* in_lds = buf(nl, pp) + __builtin_amdgcn_readfirstlane(row[7]);
        float* out_lds = buf(nl, pp ^ 1);
        const float act_slope = a.act == GNF_ACT_RELU ? 0.f : a.alpha;
        const float slope = (mode == 1 || last_fwd) ? 1.f : act_slope;
        EpiArgs ea;
        ea.mode = mode;
        ea.dump = nullptr;  // the layer's outputs leave through the coalesced copy below, not element by element
        ea.dld = __builtin_amdgcn_readfirstlane(row[6]);
        ea.width = __builtin_amdgcn_readfirstlane(row[3]);
        ea.row0 = row0;
        ea.n_nodes = a.n_nodes;
        ea.mask = slot >= 0 ? masks + ((size_t)(nl * (a.K - 1) + slot)) * (MT * 4) * a.mld : nullptr;
        ea.mld = a.mld;
        ea.act_slope = act_slope;
        while (cur.layer == r) {  // wave-uniform
            const WChunk c = cur;
            const WChunk nxt = next_chunk(c);
            const WChunk nx = nxt.layer < R ? nxt : c;
            const float* bl = bias_lds + nl * a.bias_tot2 + c.boff;
            if (c.nv >= 4)
                mlp_chunk<MT, 4, EPI_EX>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea);
            else if (c.nv == 3)
                mlp_chunk<MT, 3, EPI_EX>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea);
            else if (c.nv == 2)
                mlp_chunk<MT, 2, EPI_EX>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea);
            else
                mlp_chunk<MT, 1, EPI_EX>(in_lds, LS, c, nx, WPN, bl, out_lds, slope, lane, b_pre, ea);
            cur = nxt;
        }
        if (pend >= 0) dump_row(pend);  // (the previous row's outputs = this row's input buffer)
        pend = r;
        pp ^= 1;
        __syncthreads();
    };

    // ---- B: recompute -------------------------------------------------------------------------
    if constexpr (!STASHED) {
        for (int r = 0; r < a.K; ++r) run_row(r);
        flush_rows();
    }

    // ---- the previous half-step's batch-norm bijector: per-feature constants (see BwdArgs.bn_part) -------------------
    float* bnc = f_own;  // [6][H]: m1 = mean(gamma Gy), m2 = mean(gamma Gy xh), gamma, beta, sigma, mu (the folded scatter is off)
    const bool bnf = a.bn_part != nullptr;
    if (bnf) {
        double* red = reinterpret_cast<double*>(buf(0, pp ^ 1));  // [G][H][2] (free until the coupling stage writes g_s there)
        int G = kBwdThreads / H;                                  // (H <= 128: the host checks)
        if (G > TM * LS / (4 * H)) G = TM * LS / (4 * H);         // ... and as many groups as the buffer holds fp64 pairs for (>= 4)
        const int c = tid % H, g = tid / H;
        if (g < G) {
            double s = 0.0, q = 0.0;
            for (int b0 = g; b0 < a.bn_nparts; b0 += 8 * G) {  // eight partial pairs in flight per thread
                double ps[8], pq[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int b = b0 + k * G < a.bn_nparts ? b0 + k * G : g;
                    ps[k] = a.bn_part[((int64_t)b * H + c) * 2 + 0];
                    pq[k] = a.bn_part[((int64_t)b * H + c) * 2 + 1];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (b0 + k * G < a.bn_nparts) s += ps[k], q += pq[k];
            }
            red[(g * H + c) * 2 + 0] = s;
            red[(g * H + c) * 2 + 1] = q;
        }
        __syncthreads();
        if (tid < H) {
            double s = 0.0, q = 0.0;
            for (int g2 = 0; g2 < G; ++g2) s += red[(g2 * H + tid) * 2 + 0], q += red[(g2 * H + tid) * 2 + 1];
            const float gm = a.bn_gamma[tid];
            const double nm = (double)a.n_nodes;
            bnc[0 * H + tid] = (float)(s / nm) * gm;
            bnc[1 * H + tid] = (float)(q / nm) * gm;
            bnc[2 * H + tid] = gm;
            bnc[3 * H + tid] = a.bn_beta[tid];
            bnc[4 * H + tid] = sqrtf(a.bn_var[tid] + a.bn_eps);
            bnc[5 * H + tid] = a.bn_mean[tid];
            if (bid == 0) {
                a.bn_dbeta[tid] = (float)s;
                a.bn_dgamma[tid] = (float)(q - nm / (double)gm);
            }
        }
        __syncthreads();
    }

    // ---- C': coupling, undone and differentiated --------------------------------------------------
    {
        const float* s_lds = buf(0, pp);
        const float* t_lds = buf(1, pp);
        float* gs_lds = buf(0, pp ^ 1);
        float* gt_lds = buf(1, pp ^ 1);
        const int hp = a.tab[a.K][0] * 16;  // padded input width of the first backward row
        for (int idx = tid; idx < TM * hp; idx += kBwdThreads) {
            const int rl = idx / hp, f = idx - rl * hp;
            const int r = row0 + rl;
            float gs = 0.f, gt = 0.f;
            if (r < a.n_nodes && f < H) {
                float sv, tv;
                if constexpr (STASHED) {
                    sv = a.st_in[0][(int64_t)r * H + f], tv = a.st_in[1][(int64_t)r * H + f];
                } else {
                    sv = s_lds[rl * LS + f], tv = t_lds[rl * LS + f];
                }
                if (!STASHED && a.residual) {  // (the stashed s, t are what the coupling used: the residual is in)
                    const float xr = a.x_cond[(int64_t)r * a.ld + f];
                    sv += xr;
                    tv += xr;
                }
                float* py = a.y_upd + (int64_t)r * a.ld + f;
                float* pg = a.g_upd + (int64_t)r * a.ldg + f;
                float yv = *py;
                float gv = *pg;
                if (bnf) {  // k_bn_bwd_apply's arithmetic: the gradient through the bijector and its batch moments, the state rebuilt
                    const float xh = (yv - bnc[3 * H + f]) / bnc[2 * H + f], sig = bnc[4 * H + f];
                    gv = (bnc[2 * H + f] * gv - bnc[0 * H + f] - xh * bnc[1 * H + f] + xh) / sig;
                    yv = xh * sig + bnc[5 * H + f];
                }
                if (fold) gv = gv + f_own[rl * HP + f] + f_acc[rl * HP + f];
                const float d = yv - tv;
                *py = d * expf(-sv);
                *pg = gv * expf(sv);
                gs = gv * d - 1.f;
                gt = gv;
                a.gst[0][(int64_t)r * H + f] = gs;
                a.gst[1][(int64_t)r * H + f] = gt;
            }
            gs_lds[rl * LS + f] = gs;
            gt_lds[rl * LS + f] = gt;
        }
        pp ^= 1;
        __syncthreads();
    }

    // ---- B': the layers backwards -----------------------------------------------------------------
    for (int r = a.K; r < R; ++r) run_row(r);
    flush_rows();
}


// host side (gnf_fused_bwd.hip): fill the kernel arguments of one half-step; *mt / *tiles / *lds = the launch shape
int build_bwd_args(const int32_t* rowptr, const int32_t* col, int64_t n, const GnfGnnSpec& gnn, const GnfMlp* s,
                   const GnfMlp* t, const float* x_cond, float* y_upd, int64_t ld, float* g_upd, int64_t ldg, int32_t H,
                   float* h0_out, const float* const* h0_in, float* const* hin, int64_t ldh, float* const* dP,
                   int64_t lddp, float* const* gst, float* const* dh0, BwdArgs* out, int* mt, int64_t* tiles, size_t* lds,
                   const BwdFold* fold = nullptr);
int launch_half_bwd_fused_stashed(const BwdArgs& a, int mt, int64_t tiles, size_t lds, hipStream_t st);
// appends the row dagg[q] = dL/dh0[q][:, off : off + C] Wo_q^T ([n, NV], row stride NV); wot[q]: Wo_q ([NV, C]) as packed
// transposed fragments (k-groups over C, column tiles over NV: k_pack_layer's wtout layout).  false: no room / widths
// the tile does not hold - the caller keeps its GEMM launch.
bool bwd_args_add_dagg_row(BwdArgs* a, const float* const* wot, float* const* dagg, int C, int NV, int off);

}  // namespace gnf
