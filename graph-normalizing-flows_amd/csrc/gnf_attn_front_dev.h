// Device side of the one-launch attention front-end (gnf_attn_front.hip has the design): the per-tile function, its
// argument / LDS layout structs.  Included by gnf_attn_front.hip (the standalone kernel k_attn_front) and by gnf_fused.hip
// (k_half_fused's attention instance, which runs it as the prologue of the fused half-step kernel).
#pragma once
#include "gnf_attn_dev.h"

namespace gnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

static constexpr int kFrThreads = 512;
static constexpr int kFrRows = 16;         // receiver rows per workgroup (one MFMA M-tile)
static constexpr int kFrWin = 128;         // window nodes projected per pass (eight M-tiles: two per wave of a net)
static constexpr int kFrColCap = 2048;     // ints of LDS for the tile's col slice (longer slices are read from global memory)

// ---- packed weights of one attention block (floats): Wqv | Wk | Wo, each [K/16][N/16][64 lanes][4]:
//      frag(kg, nt, lane, q) = W[16 kg + 4 (lane >> 4) + q][16 nt + (lane & 15)], zero outside the true matrix;
//      Wqv = [Wq | 0 | Wv | 0] with the v block starting at column nqp --------------------------------------------
struct FrontDims {
    int H, nh, kq, vd, C, nq, NV;
    int Hp, nqp, vdp, NVp, Cp, PW;
};
static __host__ __device__ inline FrontDims front_dims(int H, int nh, int kq, int vd, int C) {
    FrontDims d;
    d.H = H, d.nh = nh, d.kq = kq, d.vd = vd, d.C = C, d.nq = nh * kq, d.NV = nh * vd;
    d.Hp = (H + 15) & ~15, d.nqp = (d.nq + 15) & ~15, d.vdp = (vd + 15) & ~15, d.NVp = (d.NV + 15) & ~15, d.Cp = (C + 15) & ~15;
    d.PW = d.nqp + d.vdp;
    return d;
}
bool attn_front_fixed_geometry(const FrontDims& d);  // gnf_attn_front.hip: is this the geometry the FIXED instances are compiled for?
static __host__ __device__ inline size_t front_pack_floats(const FrontDims& d) {
    return (size_t)d.Hp * d.PW + (size_t)d.Hp * d.nqp + (size_t)d.NVp * d.Cp;
}
// ------------------------------------------------------------------------------------------------------------------
struct FrontArgs {
    const float* packed[2];  // per net: Wqv | Wk | Wo fragments
    float* qkv[2];           // NULL, or [N, 2*nq + v] per net (q | k | v of every node)
    float* h0[2];            // [N, in0] per net
    float* agg_out[2];       // NULL, or [N, heads*v] per net: the attended values (kept for the backward pass)
    float* mz_out[2];        // NULL, or [N, 3*heads] per net: softmax running max at [h], denominator at [heads + h]
    const int32_t* rowptr;
    const int32_t* col;
    const float* x;
    int64_t ldx;
    int32_t n_nodes, concat, in0;
    FrontDims d;
    float scale;
    // Batch-norm bijector in front of the half-step, applied where the conditioning rows are read (the fused kernel's
    // attention prologue only; NULL bn_part: none - the rows are normalised already).  Every workgroup adds up the
    // bn_nparts partial rows (sum x, sum x^2 per feature, fp64) of the conditioning half - the arithmetic of k_bn_apply
    // (gnf_bn.hip) - and normalises its own rows and its window's rows as it stages them; the rows in global memory stay
    // RAW (the next half-step's kernel, which rewrites that half, applies bn_const_out first); workgroup 0 also leaves
    // the batch moments, the log-det term and the (scale, shift) pairs.
    const double* bn_part;
    const float* bn_gamma;
    const float* bn_beta;
    float* bn_mean_out;
    float* bn_var_out;
    double* bn_logdet_out;
    float* bn_const_out;  // [2][H]: scale | shift
    int32_t bn_nparts;
    float bn_eps;
    // NULL, or [tiles][2]: the sender window (lo, hi) of every 16-row tile of the batch (k_attn_tiles, once per flow call: the
    // topology is the same for all 2 T half-steps).  With it the tile's col slice and its window's x rows are requested right
    // behind the first round trip (rowptr | own rows | this pair) and arrive behind the k projection, instead of two further
    // dependent round trips (col -> window -> rows): round-5 phase stamps 3.2 k + 1.1 k cycles of the 27.6 k front-end
    const int32_t* tiles;
};

// LDS carve (floats); strides + 4 keep rows 16-byte aligned and spread rows over banks
struct FrontLds {
    int xs_r, xs_e, nets, per_net, k_r, qv_e, agg, ints, total;
    int ldx, ldqv, ldk, ldagg;
};
static __host__ __device__ inline FrontLds front_lds(const FrontDims& d) {
    FrontLds L;
    L.ldx = d.Hp + 4, L.ldqv = d.PW + 4, L.ldk = d.nqp + 4, L.ldagg = d.NVp + 4;
    int o = 0;
    L.xs_r = o, o += kFrRows * L.ldx;
    L.xs_e = o, o += kFrWin * L.ldx;
    int p = 0;
    L.k_r = p, p += kFrRows * L.ldk;
    L.qv_e = p, p += kFrWin * L.ldqv;
    L.agg = p, p += kFrRows * L.ldagg;
    L.per_net = (p + 3) & ~3;
    L.nets = o, o += 2 * L.per_net;
    L.ints = o, o += kFrRows + 4 + 4 + kFrColCap;  // rowptr slice (17, padded to 20) | window header | col slice
    L.total = (o + 3) & ~3;
    return L;
}

// HOIST: the [Wq | Wv] fragments of the wave's projection (PW/16 <= 6 column tiles x Hp/16 <= 2 k-groups: the
// reference's head geometry at H <= 32) stay in registers for the whole kernel instead of being re-read per chunk.
// KQM / VDM: register widths of a thread's k / q / v rows (kq <= KQM, v <= VDM); EU: edges whose rows are in flight.
// EXACT: kq == KQM and v == VDM, both even (the row loads then carry no run-time predicates: with them hipcc puts every
// LDS read of the edge loop into its own basic block behind a scalar branch and waits for each one separately).
// TO_LDS (the fused half-step kernel's attention instance, k_half_fused<1, 2, false, true>): h0 of the two nets goes to
// h0_lds0 / h0_lds1 (LDS rows of stride h0_ls floats: the MLP's layer-0 input buffers, which may ALIAS this function's own
// x / q|v staging area - everything they overlap is dead behind the barrier in front of the output projection), and to
// a.h0 as well when that is not NULL (training forward); `before_out()` runs between that barrier and the output projection (the caller's own prefetches).
// FIXED: the head geometry is the drivers' default at D = 64 (H = 32, 8 heads, kq = v = 10, C = 80: run_grevnet.py:59-80) -
// every width below is a compile-time constant and the loops over k-groups / column tiles lose their run-time predicates
// (a branch around every group of four MFMAs otherwise).
template <bool HOIST, int KQM, int VDM, int EU, bool EXACT, bool TO_LDS, bool FIXED, class Hook>
__device__ __forceinline__ void attn_front_tile(const FrontArgs& a, float* __restrict__ lds, const int row0, float* h0_lds0,
                                                float* h0_lds1, const int h0_ls, Hook&& before_out, float* bn_lds = nullptr) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int net = wave >> 2, wn = wave & 3;  // waves 0-3: s-net, 4-7: t-net
    const int lrow = lane & 15, lgrp = lane >> 4;
    const FrontDims d = FIXED ? front_dims(32, 8, 10, 10, 80) : a.d;
    const int H = d.H, nh = d.nh, kq = d.kq, vd = d.vd, nq = d.nq, NV = d.NV, P = 2 * nq + vd;
    const int Hp = d.Hp, nqp = d.nqp, NVp = d.NVp, Cp = d.Cp, PW = d.PW;
    const FrontLds L = front_lds(d);
    float* xs_r = lds + L.xs_r;
    float* xs_e = lds + L.xs_e;
    float* nb = lds + L.nets + net * L.per_net;
    float* k_r = nb + L.k_r;
    float* qv_e = nb + L.qv_e;
    float* agg = nb + L.agg;
    int* rp_l = reinterpret_cast<int*>(lds + L.ints);
    int* hdr_l = rp_l + kFrRows + 4;  // window lo / hi
    int* col_l = hdr_l + 4;
    const int tn = tid & 255;  // thread index inside the net's half of the workgroup
    const float* wqv_p = a.packed[net];
    const float* wk_p = wqv_p + (size_t)Hp * PW;
    const float* wo_p = wk_p + (size_t)Hp * nqp;
    const int voff = lane * 4;  // this lane's float4 inside a 256-float fragment block
    auto b_frag = [&](const float* packed, int nts, int g, int nt) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(packed + ((size_t)(g * nts + nt) * 64) * 4 + voff);
    };
    // Wk fragments of this wave's first two column tiles of k (wn, wn + 4): requested before anything else
    constexpr int kKNT = 2, kKKG = 2;
    const bool wk_hoist = HOIST && (nqp >> 4) <= 4 * kKNT;
    f32x4 bwk[HOIST ? kKNT : 1][HOIST ? kKKG : 1];
    if (HOIST) {
#pragma unroll
        for (int t = 0; t < kKNT; ++t)
#pragma unroll
            for (int g = 0; g < kKKG; ++g)
                bwk[t][g] = (wk_hoist && wn + 4 * t < (nqp >> 4) && g < (Hp >> 4)) ? b_frag(wk_p, nqp >> 4, g, wn + 4 * t)
                                                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- P0: rowptr slice, own x rows (one batch of independent loads, then the LDS stores) -------------------
    int rp_reg = 0;
    if (tid <= kFrRows) {
        const int r = row0 + tid;
        rp_reg = a.rowptr[r < a.n_nodes ? r : a.n_nodes];
    }
    const bool have_tab = a.tiles != nullptr;  // (workgroup-uniform)
    int tab_reg = 0;
    if (have_tab && tid >= 32 && tid < 34) tab_reg = a.tiles[2 * (row0 / kFrRows) + (tid - 32)];
    {
        constexpr int kB = 4;  // 16 rows x Hp <= 128 floats = 2048 = 512 threads x 4
        float reg[kB];
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int i = tid + u * kFrThreads;
            const int rl = i / Hp, f = i - rl * Hp;
            const bool ok = i < kFrRows * Hp && row0 + rl < a.n_nodes && f < H;
            reg[u] = a.x[ok ? (int64_t)(row0 + rl) * a.ldx + f : 0];  // unconditional load from a clamped address
        }
        // the bijector's per-feature (scale, shift) from the partial sums (see FrontArgs.bn_part), bn_lds = [2][Hp]
        const bool bnf = TO_LDS && bn_lds != nullptr && a.bn_part != nullptr;  // (workgroup-uniform)
        if (bnf) {
            double* red = reinterpret_cast<double*>(lds + L.nets + L.qv_e);  // [G][H][2] (free until the first projection)
            const int G = kFrThreads / H;
            const int c = tid % H, g = tid / H;
            if (g < G) {
                double s_ = 0.0, q_ = 0.0;
                // sixteen partial pairs in flight per thread: the 170 rows of a config-2 batch over 16 thread groups are ONE
                // round trip (eight: two dependent ones, 2 k cycles of this prologue); the additions keep their order
                for (int b0 = g; b0 < a.bn_nparts; b0 += 16 * G) {
                    double ps[16], pq[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int b = b0 + k * G < a.bn_nparts ? b0 + k * G : g;
                        ps[k] = a.bn_part[((int64_t)b * H + c) * 2 + 0];
                        pq[k] = a.bn_part[((int64_t)b * H + c) * 2 + 1];
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (b0 + k * G < a.bn_nparts) s_ += ps[k], q_ += pq[k];
                }
                red[(g * H + c) * 2 + 0] = s_;
                red[(g * H + c) * 2 + 1] = q_;
            }
            __syncthreads();
            double ld_c = 0.0;
            if (tid < H) {
                double s_ = 0.0, q_ = 0.0;
                for (int g2 = 0; g2 < G; ++g2) s_ += red[(g2 * H + tid) * 2 + 0], q_ += red[(g2 * H + tid) * 2 + 1];
                const double nm = (double)a.n_nodes;
                const double mean = s_ / nm;
                double var = q_ / nm - mean * mean;
                if (var < 0.0) var = 0.0;
                const float gm = a.bn_gamma[tid];
                const float sc = gm / sqrtf((float)var + a.bn_eps);
                const float sh = a.bn_beta[tid] - (float)mean * sc;
                bn_lds[tid] = sc;
                bn_lds[Hp + tid] = sh;
                ld_c = log((double)gm) - 0.5 * log(var + (double)a.bn_eps);
                if (row0 == 0) {
                    if (a.bn_mean_out) a.bn_mean_out[tid] = (float)mean;
                    if (a.bn_var_out) a.bn_var_out[tid] = (float)var;
                    a.bn_const_out[tid] = sc;
                    a.bn_const_out[H + tid] = sh;
                }
            }
            if (row0 == 0) {  // log-det term: N * sum_f (log gamma_f - 0.5 log(var_f + eps)), summed in feature order
                __syncthreads();
                if (tid < H) red[tid] = ld_c;
                __syncthreads();
                if (tid == 0) {
                    double tot = 0.0;
                    for (int f = 0; f < H; ++f) tot += red[f];
                    *a.bn_logdet_out = (double)a.n_nodes * tot;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < kB; ++u) {
            const int i = tid + u * kFrThreads;
            const int rl = i / Hp, f = i - rl * Hp;
            const bool ok = row0 + rl < a.n_nodes && f < H;
            float v = reg[u];
            if (bnf && ok) v = v * bn_lds[f] + bn_lds[Hp + f];
            if (i < kFrRows * Hp) xs_r[rl * L.ldx + f] = ok ? v : 0.f;
        }
    }
    if (tid <= kFrRows) rp_l[tid] = rp_reg;
    if (have_tab && tid >= 32 && tid < 34) hdr_l[tid - 32] = tab_reg;
    __syncthreads();
    // With the window table: the tile's col slice and the x rows of the first window pass are requested HERE and land behind
    // the k projection below (one batch each; longer slices / wider rows keep the loops further down)
    const bool pre_x = have_tab && kFrWin * (Hp >> 2) <= 2 * kFrThreads;
    int pre_col[4] = {0, 0, 0, 0};
    f32x4 pre_v4[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (have_tab) {
        const int sb = rp_l[0], sl = rp_l[kFrRows] - sb;
        if (sl > 0 && sl <= kFrColCap) {   // (sl == 0: a tile of isolated receivers - at the batch's end sb == n_edges, one int past col)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = tid + u * kFrThreads;
                pre_col[u] = a.col[sb + (i < sl ? i : 0)];
            }
        }
        if (pre_x) {
            const int w0 = hdr_l[0], nw = hdr_l[1] + 1 - w0 < kFrWin ? hdr_l[1] + 1 - w0 : kFrWin, f4n = Hp >> 2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + u * kFrThreads;
                const int e = i / f4n, f = (i - e * f4n) * 4;
                const float* src = a.x + (int64_t)(w0 + (e < nw ? e : 0)) * a.ldx;   // (an empty tile: nw <= 0, row w0 = 0x7fffffff is never read:
                if (nw > 0) {                                                        //  guarded)
#pragma unroll
                    for (int q = 0; q < 4; ++q) pre_v4[u][q] = src[f + q < H ? f + q : 0];
                }
            }
        }
    }

    // K order inside a group of 16 is permuted identically on both operands (k = 16 g + 4 (lane >> 4) + q), so an A
    // fragment is one 16-byte LDS read and a B fragment one 16-byte global read.
    auto a_frag = [&](const float* base, int ld, int g) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(base + lrow * ld + 16 * g + 4 * lgrp);
    };
    constexpr int kMaxKG = 8;  // H <= 128
    // ---- P1: k = x_r Wk (+ own-row q | v for the stash) -----------------------------------------------------------
    {
        f32x4 av[kMaxKG];
#pragma unroll
        for (int g = 0; g < kMaxKG; ++g)
            if (g < (Hp >> 4)) av[g] = a_frag(xs_r, L.ldx, g);
        int kt_ = 0;
        for (int nt = wn; nt < (nqp >> 4); nt += 4, ++kt_) {
            f32x4 bv[kMaxKG];
#pragma unroll
            for (int g = 0; g < kMaxKG; ++g)
                if (g < (Hp >> 4)) {
                    if (HOIST && wk_hoist && g < kKKG)
                        bv[g] = kt_ == 0 ? bwk[0][g] : bwk[kKNT - 1][g];
                    else
                        bv[g] = b_frag(wk_p, nqp >> 4, g, nt);
                }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < kMaxKG; ++g)
                if (g < (Hp >> 4)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][q], bv[g][q], acc, 0, 0, 0);
                }
            const int c = 16 * nt + lrow;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * lgrp + r;
                k_r[rl * L.ldk + c] = acc[r];
                if (a.qkv[net] && row0 + rl < a.n_nodes && c < nq) a.qkv[net][(int64_t)(row0 + rl) * P + nq + c] = acc[r];
            }
        }
        if (a.qkv[net]) {  // own rows' q | v (the stash keeps per-node projections)
            for (int nt = wn; nt < (PW >> 4); nt += 4) {
                f32x4 bv[kMaxKG];
#pragma unroll
                for (int g = 0; g < kMaxKG; ++g)
                    if (g < (Hp >> 4)) bv[g] = b_frag(wqv_p, PW >> 4, g, nt);
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < kMaxKG; ++g)
                    if (g < (Hp >> 4)) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][q], bv[g][q], acc, 0, 0, 0);
                    }
                const int c = 16 * nt + lrow;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = 4 * lgrp + r;
                    if (row0 + rl >= a.n_nodes) continue;
                    float* dst = a.qkv[net] + (int64_t)(row0 + rl) * P;
                    if (c < nq)
                        dst[c] = acc[r];
                    else if (c >= nqp && c - nqp < vd)
                        dst[2 * nq + (c - nqp)] = acc[r];
                }
            }
        }
    }
    const int seg_beg = rp_l[0], seg_end = rp_l[kFrRows];
    constexpr int kHoistNT = 6, kHoistKG = 2;
    f32x4 bqv[HOIST ? kHoistNT : 1][HOIST ? kHoistKG : 1];
    if (HOIST) {
#pragma unroll
        for (int nt = 0; nt < kHoistNT; ++nt)
#pragma unroll
            for (int g = 0; g < kHoistKG; ++g)
                bqv[nt][g] = (nt < (PW >> 4) && g < (Hp >> 4)) ? b_frag(wqv_p, PW >> 4, g, nt) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // thread = (row, head, parity): the row's edges are split between two threads when heads <= 8 (all 256 threads of the
    // net busy), merged at the end; k row in registers, online-softmax state in registers
    const bool split = 2 * kFrRows * nh <= 256;
    const int t_pair = split ? tn >> 1 : tn, t_par = split ? tn & 1 : 0, t_stride = split ? 2 : 1;
    const bool att = t_pair < kFrRows * nh;
    const int t_rl = att ? t_pair / nh : 0, t_h = att ? t_pair - t_rl * nh : 0;
    const bool v2 = (kq & 1) == 0;  // q rows start 8-byte aligned and have an even length -> 8-byte LDS reads
    float kreg[KQM], ag[VDM];
    float m_run = -INFINITY, z_run = 0.f;
#pragma unroll
    for (int j = 0; j < VDM; ++j) ag[j] = 0.f;
    // ---- the tile's col slice into LDS, its sender window [lo, hi] ----------------------------------------------------
    const int seg_len = seg_end - seg_beg;
    const bool col_staged = seg_len <= kFrColCap;
    if (have_tab) {
        // the prefetched col slice and window rows into LDS (the rows normalised like the own rows when a bijector rides along)
        if (col_staged) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = tid + u * kFrThreads;
                if (i < seg_len) col_l[i] = pre_col[u];
            }
            for (int i = tid + 4 * kFrThreads; i < seg_len; i += kFrThreads) col_l[i] = a.col[seg_beg + i];  // (slices of more than 2048 edges: never staged)
        }
        if (pre_x) {
            const int w0 = hdr_l[0], nw = hdr_l[1] + 1 - w0 < kFrWin ? hdr_l[1] + 1 - w0 : kFrWin, f4n = Hp >> 2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + u * kFrThreads;
                const int e = i / f4n, f = (i - e * f4n) * 4;
                if (e >= kFrWin) continue;
                f32x4 v = pre_v4[u];
                if (TO_LDS && bn_lds != nullptr && a.bn_part != nullptr) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (f + q < H) v[q] = v[q] * bn_lds[f + q] + bn_lds[Hp + f + q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (!(e < nw && f + q < H)) v[q] = 0.f;
                *reinterpret_cast<f32x4*>(xs_e + e * L.ldx + f) = v;
            }
        }
        __syncthreads();  // (also: k_r is complete)
    } else {
        if (tid == 0) {
            hdr_l[0] = 0x7fffffff;
            hdr_l[1] = -1;
        }
        __syncthreads();  // (also: k_r is complete)
        int lo = 0x7fffffff, hi = -1;
        for (int base = 0; base < seg_len; base += kFrThreads * 4) {
            int reg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + tid + u * kFrThreads;
                reg[u] = a.col[seg_beg + (i < seg_len ? i : 0)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + tid + u * kFrThreads;
                if (i < seg_len) {
                    if (col_staged) col_l[i] = reg[u];
                    lo = reg[u] < lo ? reg[u] : lo;
                    hi = reg[u] > hi ? reg[u] : hi;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lane == 0 && hi >= 0) {
            atomicMin(&hdr_l[0], lo);
            atomicMax(&hdr_l[1], hi);
        }
    }
#pragma unroll
    for (int j = 0; j < KQM; ++j) kreg[j] = (att && (EXACT || j < kq)) ? k_r[t_rl * L.ldk + t_h * kq + j] : 0.f;
    const int t_beg = rp_l[t_rl], t_end = att ? rp_l[t_rl + 1] : t_beg;
    if (!have_tab) __syncthreads();
    const int win_lo = hdr_l[0], win_hi = hdr_l[1];  // empty tile: hi = -1 < lo
    int chunk_ = 0;
    (void)chunk_;

    for (int w0 = win_lo; w0 <= win_hi; w0 += kFrWin) {
        const int nw = win_hi + 1 - w0 < kFrWin ? win_hi + 1 - w0 : kFrWin;  // window nodes of this pass
        if (chunk_ > 0) __syncthreads();  // the previous pass's buffers are free
        // ---- C2: the window's x rows (contiguous rows: coalesced) ------------------------------------------------------
        if (!(pre_x && chunk_ == 0)) {   // (the first pass's rows are in LDS already when the window table gave their range up front)
            const int f4n = Hp >> 2;
            for (int base = 0; base < kFrWin * f4n; base += 2 * kFrThreads) {  // two slots per thread in flight
                f32x4 v4[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = base + tid + u * kFrThreads;
                    const int e = i / f4n, f = (i - e * f4n) * 4;
                    const float* src = a.x + (int64_t)(w0 + (e < nw ? e : 0)) * a.ldx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v4[u][q] = src[f + q < H ? f + q : 0];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = base + tid + u * kFrThreads;
                    const int e = i / f4n, f = (i - e * f4n) * 4;
                    if (e >= kFrWin) continue;
                    if (TO_LDS && bn_lds != nullptr && a.bn_part != nullptr) {  // (the bijector, as for the own rows)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (f + q < H) v4[u][q] = v4[u][q] * bn_lds[f + q] + bn_lds[Hp + f + q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (!(e < nw && f + q < H)) v4[u][q] = 0.f;
                    *reinterpret_cast<f32x4*>(xs_e + e * L.ldx + f) = v4[u];
                }
            }
        }
        if (!(pre_x && chunk_ == 0)) __syncthreads();
        // ---- C3: q | v of the window's nodes: wave wn of each net takes M-tiles wn and wn + 4 (16 nodes each) -----------
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = wn + 4 * mt;
            if (16 * m >= nw) break;  // wave-uniform
            const float* abase = xs_e + 16 * m * L.ldx;
            f32x4 av[kMaxKG];
#pragma unroll
            for (int g = 0; g < kMaxKG; ++g)
                if (g < (Hp >> 4)) av[g] = a_frag(abase, L.ldx, g);
            if (HOIST) {
#pragma unroll
                for (int nt = 0; nt < kHoistNT; ++nt)
                    if (nt < (PW >> 4)) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < kHoistKG; ++g)
                            if (g < (Hp >> 4)) {
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][q], bqv[nt][g][q], acc, 0, 0, 0);
                            }
#pragma unroll
                        for (int r = 0; r < 4; ++r) qv_e[(16 * m + 4 * lgrp + r) * L.ldqv + 16 * nt + lrow] = acc[r];
                    }
            } else {
                for (int nt0 = 0; nt0 < (PW >> 4); nt0 += 2) {  // two column tiles per round: their B loads in flight together
                    const int nt1 = nt0 + 1 < (PW >> 4) ? nt0 + 1 : nt0;
                    f32x4 b0[kMaxKG], b1[kMaxKG];
#pragma unroll
                    for (int g = 0; g < kMaxKG; ++g)
                        if (g < (Hp >> 4)) {
                            b0[g] = b_frag(wqv_p, PW >> 4, g, nt0);
                            b1[g] = b_frag(wqv_p, PW >> 4, g, nt1);
                        }
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < kMaxKG; ++g)
                        if (g < (Hp >> 4)) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][q], b0[g][q], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][q], b1[g][q], acc1, 0, 0, 0);
                            }
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* dst = qv_e + (16 * m + 4 * lgrp + r) * L.ldqv + lrow;
                        dst[16 * nt0] = acc0[r];
                        if (nt1 != nt0) dst[16 * nt1] = acc1[r];
                    }
                }
            }
        }
        __syncthreads();
        // ---- C4: this thread's share of the row's edges whose sender lies in the pass's window part, EU at a time (their
        // q | v rows are independent LDS reads; only the softmax recurrence is sequential) -----------------------------------
        if (att) {
            for (int e = t_beg + t_par; e < t_end; e += EU * t_stride) {
                int sidx[EU];
                float qq[EU][KQM], vv[EU][VDM];
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int eu = e + u * t_stride;
                    const int ec = eu < t_end ? eu : t_beg;  // clamped: the load is unconditional, the update is not
                    sidx[u] = (col_staged ? col_l[ec - seg_beg] : a.col[ec]) - w0;
                    if (eu >= t_end) sidx[u] = -1;
                }
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const float* row = qv_e + (sidx[u] >= 0 && sidx[u] < nw ? sidx[u] : 0) * L.ldqv;
                    if (EXACT) {
                        const f32x2_attn* qp = reinterpret_cast<const f32x2_attn*>(row + t_h * KQM);
                        const f32x2_attn* vp = reinterpret_cast<const f32x2_attn*>(row + nqp);
#pragma unroll
                        for (int j = 0; j < KQM / 2; ++j) {
                            const f32x2_attn t = qp[j];
                            qq[u][2 * j] = t[0], qq[u][2 * j + 1] = t[1];
                        }
#pragma unroll
                        for (int j = 0; j < VDM / 2; ++j) {
                            const f32x2_attn t = vp[j];
                            vv[u][2 * j] = t[0], vv[u][2 * j + 1] = t[1];
                        }
                    } else {
                        load_row<KQM>(row + t_h * kq, kq, v2, qq[u]);
                        load_row<VDM>(row + nqp, vd, true, vv[u]);
                    }
                }
                // one rescale per group of EU edges: m' = max(m, logits), state *= exp(m - m'), + sum_u exp(lg_u - m') (.)
                float lg[EU], mn = m_run;
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    float l = 0.f;
#pragma unroll
                    for (int j = 0; j < KQM; ++j) l += qq[u][j] * kreg[j];
                    lg[u] = (sidx[u] >= 0 && sidx[u] < nw) ? l * a.scale : -INFINITY;
                    mn = fmaxf(mn, lg[u]);
                }
                if (mn != -INFINITY) {  // at least one live edge so far
                    const float sc = __expf(m_run - mn);  // exp(-inf) = 0 on the first live group
                    float pe[EU];
                    z_run *= sc;
#pragma unroll
                    for (int u = 0; u < EU; ++u) {
                        pe[u] = __expf(lg[u] - mn);  // masked edge: exp(-inf) = 0
                        z_run += pe[u];
                    }
#pragma unroll
                    for (int j = 0; j < VDM; ++j) {
                        float acc = ag[j] * sc;
#pragma unroll
                        for (int u = 0; u < EU; ++u) acc = fmaf(pe[u], vv[u][j], acc);
                        ag[j] = acc;
                    }
                    m_run = mn;
                }
            }
        }
        ++chunk_;
    }
    // ---- merge the two halves of every row (online-softmax merge with the partner lane) -----------------------------------
    if (split) {
        const float m_o = __shfl_xor(m_run, 1, 64), z_o = __shfl_xor(z_run, 1, 64);
        const float mn = fmaxf(m_run, m_o);
        const float s_me = m_run == -INFINITY ? 0.f : __expf(m_run - mn), s_o = m_o == -INFINITY ? 0.f : __expf(m_o - mn);
        z_run = z_run * s_me + z_o * s_o;
#pragma unroll
        for (int j = 0; j < VDM; ++j) {
            const float a_o = __shfl_xor(ag[j], 1, 64);
            ag[j] = ag[j] * s_me + a_o * s_o;
        }
        m_run = mn;
    }
    // Wo fragments of this wave's output column tiles {wn, wn + 4}: requested before the normalisation and its barrier
    constexpr int kOutNT = 2, kOutKG = 8;  // register form when C <= 128 and heads * v <= 128 (else read in the loop below)
    const bool wo_hoist = (Cp >> 4) <= 4 * kOutNT && (NVp >> 4) <= kOutKG;
    f32x4 bwo[kOutNT][kOutKG];
#pragma unroll
    for (int t = 0; t < kOutNT; ++t)
#pragma unroll
        for (int g = 0; g < kOutKG; ++g)
            bwo[t][g] = (wo_hoist && wn + 4 * t < (Cp >> 4) && g < (NVp >> 4)) ? b_frag(wo_p, Cp >> 4, g, wn + 4 * t)
                                                                            : f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- normalise (no incoming edge -> 0, gnn.py:403) and hand the attended values to the output projection -------
    if (att && t_par == 0) {
        const float inv = t_end > t_beg ? 1.f / z_run : 0.f;
#pragma unroll
        for (int j = 0; j < VDM; ++j)
            if (EXACT || j < vd) agg[t_rl * L.ldagg + t_h * vd + j] = ag[j] * inv;
        if (a.mz_out[net] && row0 + t_rl < a.n_nodes) {
            float* mz = a.mz_out[net] + (int64_t)(row0 + t_rl) * 3 * nh;
            mz[t_h] = m_run;
            mz[nh + t_h] = t_end > t_beg ? z_run : 1.f;
        }
    }
    if (NVp > NV)
        for (int i = tn; i < kFrRows * (NVp - NV); i += 256) {  // zero the pad columns the k-groups of P9 run over
            const int rl = i / (NVp - NV), c = NV + (i - rl * (NVp - NV));
            agg[rl * L.ldagg + c] = 0.f;
        }
    [[maybe_unused]] float xkeep[2] = {0.f, 0.f};  // TO_LDS + concat: this thread's own-row x values (16 x H <= 512 per net)
    if constexpr (TO_LDS) {
        if (a.concat) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tn + 256 * u;
                if (i < kFrRows * H) xkeep[u] = xs_r[(i / H) * L.ldx + (i - (i / H) * H)];
            }
        }
    }
    __syncthreads();
    before_out();
    if (a.agg_out[net])
        for (int i = tn; i < kFrRows * NV; i += 256) {
            const int rl = i / NV, c = i - rl * NV;
            if (row0 + rl < a.n_nodes) a.agg_out[net][(int64_t)(row0 + rl) * NV + c] = agg[rl * L.ldagg + c];
        }
    // ---- P9: new = agg Wo on the matrix cores, then h0 = [x || new] ------------------------------------------------
    {
        const int off = a.concat ? H : 0;
        float* h0 = a.h0[net];
        constexpr int kMaxOG = 16;  // heads * v <= 256
        int t_ = 0;
        for (int nt = wn; nt < (Cp >> 4); nt += 4, ++t_) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int c = 16 * nt + lrow;
            if (wo_hoist) {
#pragma unroll
                for (int g = 0; g < kOutKG; ++g)
                    if (g < (NVp >> 4)) {
                        const f32x4 av = a_frag(agg, L.ldagg, g);
                        const f32x4 bv = t_ == 0 ? bwo[0][g] : bwo[1][g];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);
                    }
            } else {
                f32x4 bv[kMaxOG];
#pragma unroll
                for (int u = 0; u < kMaxOG; ++u)
                    if (u < (NVp >> 4)) bv[u] = b_frag(wo_p, Cp >> 4, u, nt);
#pragma unroll
                for (int u = 0; u < kMaxOG; ++u)
                    if (u < (NVp >> 4)) {
                        const f32x4 av = a_frag(agg, L.ldagg, u);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[u][q], acc, 0, 0, 0);
                    }
            }
            if constexpr (TO_LDS) {  // (columns [C, Cp) come out as zeros: Wo's pad columns are zero)
                float* hl = net == 0 ? h0_lds0 : h0_lds1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = row0 + 4 * lgrp + r;
                    hl[(4 * lgrp + r) * h0_ls + off + c] = rr < a.n_nodes ? acc[r] : 0.f;
                    if (h0 && rr < a.n_nodes && c < d.C) h0[(int64_t)rr * a.in0 + off + c] = acc[r];  // (training forward: the stash keeps h0)
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = row0 + 4 * lgrp + r;
                    if (rr < a.n_nodes && c < d.C) h0[(int64_t)rr * a.in0 + off + c] = acc[r];
                }
            }
        }
        if (a.concat) {
            if constexpr (TO_LDS) {
                float* hl = net == 0 ? h0_lds0 : h0_lds1;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = tn + 256 * u;
                    if (i < kFrRows * H) {
                        const int rl = i / H, f = i - rl * H;
                        hl[rl * h0_ls + f] = xkeep[u];
                        if (h0 && row0 + rl < a.n_nodes) h0[(int64_t)(row0 + rl) * a.in0 + f] = xkeep[u];
                    }
                }
            } else {
                for (int i = tn; i < kFrRows * H; i += 256) {
                    const int rl = i / H, f = i - rl * H;
                    if (row0 + rl < a.n_nodes) h0[(int64_t)(row0 + rl) * a.in0 + f] = xs_r[rl * L.ldx + f];
                }
            }
        }
    }
}

}  // namespace gnf
