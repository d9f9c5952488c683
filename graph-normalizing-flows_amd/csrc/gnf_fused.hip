// Fused coupling half-step for gfx950 (MI355X): ONE launch does, for a tile of TM = 16*MT nodes,
//   A  CSR segmented reduce of neighbour rows (coalesced row reads, sum | sum/max(deg,1))   gnn.py:103-104,117-118,151-156
//      + combine  eps*x+agg | [x || agg]  straight into LDS                                 gnn.py:123 | 108-109
//   B  the s-net and the t-net MLPs, all K layers, activations resident in LDS, on the      gnn.py:159-180
//      exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain), weights
//      streamed from L2 in pre-packed fragment order (one global_load_dwordx4 per 4 MFMAs)
//   C  x_upd <- x_upd*exp(s)+t  |  (x_upd-t)*exp(-s)  and a block-reduced fp64 sum(s)        gnn.py:322-323,337-338 | 359,372
// so the [E,H] edge tensor, the aggregated tensor and every MLP activation of the reference's TF
// graph never touch HBM.
//
// Work decomposition (MI355X-first, not a tiling borrowed from a 32-wide-warp design):
//   * 512 threads = 8 wave64; waves 0-3 run the s-net, waves 4-7 the t-net, concurrently, two
//     waves per SIMD so one wave's MFMA chain covers the other's loads.
//   * inside a net, wave w owns output column tiles {w, w+4, w+8, ...} (16 columns each) and
//     processes them four at a time: one ds_read_b128 of the A fragment (16 nodes x 16 k) feeds
//     16 MFMAs per M-tile.  K order inside a group of 16 is permuted identically on both operands
//     (k = 16*kg + 4*(lane>>4) + q) so that A is one 16-byte LDS read and B one 16-byte global read.
//   * layer outputs go back to LDS in the accumulator layout (row = 4*(lane>>4)+r, col = lane&15)
//     through a ping-pong pair of [TM][LS] buffers per net, LS = widest padded layer + 4 floats
//     (keeps rows 16-byte aligned and spreads rows over LDS banks).
//   * blockIdx -> tile mapping is XCD-aware (consecutive tiles = neighbouring nodes of the same
//     graphs stay on one XCD's L2).
//
// Zero padding: every layer width is padded to a multiple of 16 in the packed weights (pad rows,
// pad columns and pad biases are 0), so padded activations are act(0) = 0 and never contribute.
#include "gnf_common.h"

namespace gnf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kFusedThreads = 512;
static constexpr int kLdsLimit = 160 * 1024;

static inline int pad16(int v) { return (v + 15) & ~15; }

// ------------------------------------------------------------------------------------------------
// packed weight layout of one MLP (floats):  for each layer j:  Wp[Op/16][Ip/16][64 lanes][4] | bias[Op]
//   Wp[nt][kg][lane][q] = W[16*kg + 4*(lane>>4) + q][16*nt + (lane&15)]   (0 outside [I,O))
// ------------------------------------------------------------------------------------------------
int64_t packed_floats(const GnfMlp* m) {
    int64_t tot = 0;
    for (int j = 0; j < m->num_layers; ++j) {
        const int64_t ip = pad16(m->dims[j]), op = pad16(m->dims[j + 1]);
        tot += ip * op + op;
    }
    return tot;
}

__global__ __launch_bounds__(256) void k_pack_layer(const float* __restrict__ W,
                                                    const float* __restrict__ b, int I, int O, int Ip,
                                                    int Op, float* __restrict__ out) {
    const int64_t nw = (int64_t)Ip * Op;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nw) {
        const int q = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        const int64_t blk = i >> 8;  // nt * (Ip/16) + kg
        const int kgs = Ip >> 4;
        const int nt = (int)(blk / kgs), kg = (int)(blk % kgs);
        const int k = 16 * kg + 4 * (lane >> 4) + q;
        const int c = 16 * nt + (lane & 15);
        out[i] = (k < I && c < O) ? W[(int64_t)k * O + c] : 0.f;
    } else if (i < nw + Op) {
        const int c = (int)(i - nw);
        out[i] = c < O ? b[c] : 0.f;
    }
}

int launch_pack_mlp(const GnfMlp* m, float* packed, hipStream_t st) {
    int64_t off = 0;
    for (int j = 0; j < m->num_layers; ++j) {
        const int I = m->dims[j], O = m->dims[j + 1], Ip = pad16(I), Op = pad16(O);
        const int64_t tot = (int64_t)Ip * Op + Op;
        hipLaunchKernelGGL(k_pack_layer, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, m->W[j],
                           m->b[j], I, O, Ip, Op, packed + off);
        GNF_LAUNCH_CHECK("k_pack_layer");
        off += tot;
    }
    return GNF_OK;
}

// ------------------------------------------------------------------------------------------------
struct FusedArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* x_cond;
    float* x_upd;
    double* partials;
    const float* wp[2][GNF_MAX_LAYERS];  // [net][layer] packed weights
    const float* bp[2][GNF_MAX_LAYERS];  // [net][layer] padded bias
    int32_t ipg[GNF_MAX_LAYERS];         // padded input width / 16 of layer j
    int32_t ont[GNF_MAX_LAYERS];         // padded output width / 16 of layer j
    int64_t ld;
    int32_t n_nodes;
    int32_t H;
    int32_t in0;      // true layer-0 input width (H or 2H)
    int32_t K;
    int32_t LS;       // LDS row stride (floats)
    int32_t mean, concat, act, inverse;
    float eps, alpha;
};

template <int MT, int NV>
__device__ __forceinline__ void mlp_chunk(const float* __restrict__ in_lds, int LS,
                                          const f32x4* __restrict__ wp, const float* __restrict__ bp,
                                          int ipg, int nt0, float* __restrict__ out_lds, bool last,
                                          int act, float alpha, int lane) {
    const int lrow = lane & 15, lgrp = lane >> 4;
    f32x4 acc[MT][NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) {
        const float bias = bp[16 * (nt0 + 4 * b) + lrow];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][b] = f32x4{bias, bias, bias, bias};
    }
    const f32x4* wtile[NV];
#pragma unroll
    for (int b = 0; b < NV; ++b) wtile[b] = wp + ((int64_t)(nt0 + 4 * b) * ipg) * 64 + lane;
    const float* arow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) arow[m] = in_lds + (16 * m + lrow) * LS + 4 * lgrp;

    f32x4 a_cur[MT], b_cur[NV];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = *reinterpret_cast<const f32x4*>(arow[m]);
#pragma unroll
    for (int b = 0; b < NV; ++b) b_cur[b] = wtile[b][0];

    for (int kg = 0; kg < ipg; ++kg) {
        f32x4 a_nxt[MT], b_nxt[NV];
        const int kn = (kg + 1 < ipg) ? kg + 1 : kg;  // last iteration re-reads (harmless)
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nxt[m] = *reinterpret_cast<const f32x4*>(arow[m] + 16 * kn);
#pragma unroll
        for (int b = 0; b < NV; ++b) b_nxt[b] = wtile[b][(int64_t)kn * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < NV; ++b)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[m][q], b_cur[b][q],
                                                                      acc[m][b], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
#pragma unroll
        for (int b = 0; b < NV; ++b) b_cur[b] = b_nxt[b];
    }
    // accumulator layout: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int b = 0; b < NV; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[m][b][r];
                if (!last) v = (act == GNF_ACT_RELU) ? fmaxf(v, 0.f) : fmaxf(v, alpha * v);
                out_lds[(16 * m + 4 * lgrp + r) * LS + 16 * (nt0 + 4 * b) + lrow] = v;
            }
}

template <int MT>
__global__ __launch_bounds__(kFusedThreads) void k_half_fused(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 16 * MT;
    const int LS = a.LS;
    // [net][pingpong][TM][LS]
    auto buf = [&](int net_, int pp_) -> float* { return smem + (2 * net_ + pp_) * TM * LS; };
    double* red = reinterpret_cast<double*>(smem + 4 * TM * LS);  // 8 doubles

    // XCD-aware, bijective blockIdx -> tile map (block b is dispatched to XCD b % 8)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int row0 = tile * TM;
    const int tid = threadIdx.x;
    const int H = a.H;

    // ---- A: aggregate + combine into both nets' layer-0 input --------------------------------
    {
        const int in0p = a.ipg[0] * 16;
        for (int idx = tid; idx < TM * in0p; idx += kFusedThreads) {
            const int rl = idx / in0p, c = idx - rl * in0p;
            const int r = row0 + rl;
            float v = 0.f;
            if (r < a.n_nodes && c < a.in0) {
                const int f = c < H ? c : c - H;
                if (a.concat && c < H) {
                    v = a.x_cond[(int64_t)r * a.ld + f];
                } else {
                    const int beg = a.rowptr[r], end = a.rowptr[r + 1];
                    float s = 0.f;
                    for (int e = beg; e < end; ++e) s += a.x_cond[(int64_t)a.col[e] * a.ld + f];
                    if (a.mean) {
                        const int cnt = end - beg;
                        s = s / (float)(cnt > 1 ? cnt : 1);
                    }
                    v = a.concat ? s : a.eps * a.x_cond[(int64_t)r * a.ld + f] + s;
                }
            }
            buf(0, 0)[rl * LS + c] = v;
            buf(1, 0)[rl * LS + c] = v;
        }
    }
    __syncthreads();

    // ---- B: K layers, s-net on waves 0-3, t-net on waves 4-7 ----------------------------------
    const int wave = tid >> 6, lane = tid & 63;
    const int net = wave >> 2, wl = wave & 3;
    int pp = 0;
    for (int j = 0; j < a.K; ++j) {
        const float* in_lds = buf(net, pp);
        float* out_lds = buf(net, pp ^ 1);
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.wp[net][j]);
        const float* bp = a.bp[net][j];
        const int ipg = a.ipg[j], ont = a.ont[j];
        const bool last = (j == a.K - 1);
        for (int nt0 = wl; nt0 < ont; nt0 += 16) {
            const int nv = (ont - nt0 + 3) >> 2;  // tiles nt0, nt0+4, ... still < ont
            if (nv >= 4)
                mlp_chunk<MT, 4>(in_lds, LS, wp, bp, ipg, nt0, out_lds, last, a.act, a.alpha, lane);
            else if (nv == 3)
                mlp_chunk<MT, 3>(in_lds, LS, wp, bp, ipg, nt0, out_lds, last, a.act, a.alpha, lane);
            else if (nv == 2)
                mlp_chunk<MT, 2>(in_lds, LS, wp, bp, ipg, nt0, out_lds, last, a.act, a.alpha, lane);
            else
                mlp_chunk<MT, 1>(in_lds, LS, wp, bp, ipg, nt0, out_lds, last, a.act, a.alpha, lane);
        }
        pp ^= 1;
        __syncthreads();
    }

    // ---- C: coupling update + block-reduced sum(s) ---------------------------------------------
    const float* s_lds = buf(0, pp);
    const float* t_lds = buf(1, pp);
    double local = 0.0;
    for (int idx = tid; idx < TM * H; idx += kFusedThreads) {
        const int rl = idx / H, f = idx - rl * H;
        const int r = row0 + rl;
        if (r < a.n_nodes) {
            const float sv = s_lds[rl * LS + f], tv = t_lds[rl * LS + f];
            float* px = a.x_upd + (int64_t)r * a.ld + f;
            const float xv = *px;
            *px = a.inverse ? (xv - tv) * expf(-sv) : xv * expf(sv) + tv;
            local += (double)sv;
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
}

// ------------------------------------------------------------------------------------------------
static int max_padded_width(const GnfMlp* m) {
    int w = 16;
    for (int j = 0; j <= m->num_layers; ++j) w = w > pad16(m->dims[j]) ? w : pad16(m->dims[j]);
    return w;
}

static size_t fused_lds_bytes(const GnfMlp* m, int MT) {
    const int LS = max_padded_width(m) + 4;
    return (size_t)4 * 16 * MT * LS * sizeof(float) + 8 * sizeof(double);
}

bool fused_supported(const HalfStep& hs) {
    const GnfMlp *s = hs.s_net, *t = hs.t_net;
    if (!s->packed || !t->packed) return false;
    if (s->num_layers != t->num_layers) return false;
    for (int j = 0; j <= s->num_layers; ++j)
        if (s->dims[j] != t->dims[j]) return false;
    return fused_lds_bytes(s, 1) <= (size_t)kLdsLimit;
}

int launch_half_fused(const HalfStep& hs, hipStream_t st) {
    const GnfMlp *s = hs.s_net, *t = hs.t_net;
    *hs.n_partials = 0;
    if (hs.n_nodes == 0) return GNF_OK;
    FusedArgs a;
    a.rowptr = hs.rowptr;
    a.col = hs.col;
    a.x_cond = hs.x_cond;
    a.x_upd = hs.x_upd;
    a.partials = hs.partials;
    int64_t off = 0;
    for (int j = 0; j < s->num_layers; ++j) {
        const int ip = pad16(s->dims[j]), op = pad16(s->dims[j + 1]);
        a.wp[0][j] = s->packed + off;
        a.wp[1][j] = t->packed + off;
        a.bp[0][j] = s->packed + off + (int64_t)ip * op;
        a.bp[1][j] = t->packed + off + (int64_t)ip * op;
        a.ipg[j] = ip / 16;
        a.ont[j] = op / 16;
        off += (int64_t)ip * op + op;
    }
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

    // 16-node tiles keep every CU busy on small batches; 32-node tiles halve the weight traffic
    // per node once there are enough tiles to fill 256 CUs twice over.
    const int64_t tiles16 = (hs.n_nodes + 15) / 16;
    const bool big = tiles16 > 1024 && fused_lds_bytes(s, 2) <= (size_t)kLdsLimit;
    const int MT = big ? 2 : 1;
    const int64_t tiles = (hs.n_nodes + 16 * MT - 1) / (16 * MT);
    const size_t lds = fused_lds_bytes(s, MT);
    if (MT == 2) {
        static bool attr2 = false;
        if (!attr2) {
            GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_fused<2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit));
            attr2 = true;
        }
        hipLaunchKernelGGL(k_half_fused<2>, dim3((unsigned)tiles), dim3(kFusedThreads), lds, st, a);
    } else {
        static bool attr1 = false;
        if (!attr1) {
            GNF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_half_fused<1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit));
            attr1 = true;
        }
        hipLaunchKernelGGL(k_half_fused<1>, dim3((unsigned)tiles), dim3(kFusedThreads), lds, st, a);
    }
    GNF_LAUNCH_CHECK("k_half_fused");
    *hs.n_partials = (int32_t)tiles;
    return GNF_OK;
}

}  // namespace gnf
