// Layer-by-layer ("layered") HIP path for gfx950: shape-generic kernels that run one coupling
// half-step as aggregate -> K x linear -> coupling epilogue through a global-memory scratch.
// It serves every shape the LDS-resident fused kernel (gnf_fused.hip) cannot hold, and the
// stand-alone kernel-A / kernel-D entry points.  Everything here is fp32 FMA in k order.
//
// Reference arithmetic being replaced (see include/gnf.h for the per-entry citations):
//   aggregate : gnn.py:103-104,117-118,151-156 (gather senders + unsorted_segment_{sum,mean})
//   combine   : gnn.py:108-109 (concat) / gnn.py:123 (eps*x + agg)
//   linear    : gnn.py:159-180 (snt.nets.MLP)
//   coupling  : gnn.py:322-323,337-338 (forward) / gnn.py:359,372 (inverse)
//   gauss     : run_grevnet.py:292-294
#include <cstring>
#include "gnf_common.h"

namespace gnf {

// ------------------------------------------------------------------------------------------------
// Kernel A: CSR segmented reduce of neighbour rows (gnn.py:103-104,117-118,151-156).
// A group of G lanes (G = power of two <= 64) owns one receiver row; lanes run along the feature axis
// VEC floats each, so every neighbour row is one coalesced read (16 B per lane when VEC = 4) and a
// wave covers 64/G rows.  Eight neighbour rows are in flight per lane; the adds stay in edge order.
//   mode 0: out[r, f] = eps * x[r, f] + agg        (AggThenMLPBlock)
//   mode 1: out[r, f] = x[r, f]; out[r, H+f] = agg (ConcatThenMLPBlock)
//   mode 2: out[r, f] = agg                        (aggregator alone)
//
// Hub rows.  A row's loop is a chain of dependent round trips (8 neighbour rows each, ~0.6 us under load): the ego
// hub of a config-5 graph (in-degree up to 194) took 25 of them, and the launch ended when the last graph's hub did
// (17.3 us on that batch against 10.3 with every row cut to 8 edges; tools/probes/agg_probe.hip).  The grid is
// therefore [front | regular]: front workgroup i, dispatched first, scans the rowptr slice of rows [256 i, 256 i + 256)
// and takes the first kAggFrontRows rows of more than kAggLong edges there - each row's edges split into contiguous
// segments over the workgroup's lane groups (a group adds its segment up in edge order, 8 rows in flight), the
// partial sums added up in group order: a fixed order, but not the sequential one (such a row differs from the
// sequential sum in rounding only).  A regular wave that meets a long row reads the same slice and skips the rows a
// front workgroup has.  A slice with more than kAggDense long rows (complete graphs: nothing to gain) is left alone.
// Config-5 batch, H = 128: 17.3 -> 12.5 us alone; H = 32: 12.3 -> 6.1 us; batches without long rows: unchanged.
// ------------------------------------------------------------------------------------------------
static constexpr int kAggLong = 32;
static constexpr int kAggFrontRows = 4;
static constexpr int kAggDense = 16;

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    typedef float type;
};
template <>
struct VecT<4> {
    typedef float type __attribute__((ext_vector_type(4)));
};

template <int VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_aggregate(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_nodes, const float* __restrict__ x,
    int64_t ldx, int H, int mean, int mode, float eps, float* __restrict__ out, int64_t ldo, int G, int n_front) {
    typedef typename VecT<VEC>::type V;
    __shared__ __attribute__((aligned(16))) V part[kAggFrontRows * 256];  // front: [row][lane group][lane]
    __shared__ int s_rp[257];
    __shared__ unsigned long long s_mask[4];
    __shared__ int s_list[kAggFrontRows];
    const int t = threadIdx.x;
    const int rows_wg = 256 / G;  // (divides 256: a regular workgroup's rows lie in one slice)
    const int gq = t / G, gl = t & (G - 1);
    // a lane group's sum over edges [beg, end) of feature slice f, in edge order, 8 neighbour rows in flight
    auto seg_sum = [&](int beg, int end, int f) {
        V acc = V(0.f);
        int e = beg;
        for (; e + 8 <= end; e += 8) {
            int ci[8];
            V vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ci[q] = col[e + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) vv[q] = *reinterpret_cast<const V*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += vv[q];
        }
        if (e < end) {  // up to 7 left, still issued together (indices clamped, adds predicated)
            int ci[7];
            V vv[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) ci[q] = col[e + q < end ? e + q : end - 1];
#pragma unroll
            for (int q = 0; q < 7; ++q) vv[q] = *reinterpret_cast<const V*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (e + q < end) acc += vv[q];
        }
        return acc;
    };
    auto finish = [&](int64_t r, int f, V acc, int deg) {
        if (mean) acc = acc / (float)(deg > 1 ? deg : 1);  // unsorted_segment_mean: max(count, 1)
        if (mode == 0) {
            const V xs = *reinterpret_cast<const V*>(x + r * ldx + f);
            *reinterpret_cast<V*>(out + r * ldo + f) = eps * xs + acc;
        } else if (mode == 1) {
            *reinterpret_cast<V*>(out + r * ldo + f) = *reinterpret_cast<const V*>(x + r * ldx + f);
            *reinterpret_cast<V*>(out + r * ldo + H + f) = acc;
        } else {
            *reinterpret_cast<V*>(out + r * ldo + f) = acc;
        }
    };
    auto clampi = [&](int64_t i) { return rowptr[i < n_nodes ? i : n_nodes]; };

    if ((int)blockIdx.x >= n_front) {
        // ---- regular workgroup.  XCD-aware bijective block remap (block b runs on XCD b % 8): consecutive row blocks -
        // the nodes of one graph and their neighbours - share one XCD's L2 instead of being sprayed over all eight.
        const int64_t nwg = gridDim.x - n_front, bid = blockIdx.x - n_front;
        const int64_t xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
        const int64_t blk = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
        const int64_t r = blk * rows_wg + gq;
        const bool live = r < n_nodes;
        int beg = 0, end = 0;
        if (live) {
            beg = rowptr[r];
            end = rowptr[r + 1];
        }
        bool taken = false;
        if (__ballot(end - beg > kAggLong)) {  // (wave-uniform, rare) does a front workgroup have the row?
            const int64_t sbase = (blk * rows_wg) & ~(int64_t)255;
            const int lane = t & 63;
            unsigned long long m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = __ballot(clampi(sbase + 64 * k + lane + 1) - clampi(sbase + 64 * k + lane) > kAggLong);
            const int total = __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]);
            if (live && end - beg > kAggLong && total <= kAggDense) {
                const int tl = (int)(r - sbase), w = tl >> 6;
                int rank = __popcll((w == 0 ? m[0] : w == 1 ? m[1] : w == 2 ? m[2] : m[3]) & ((1ull << (tl & 63)) - 1ull));
                rank += (w > 0 ? __popcll(m[0]) : 0) + (w > 1 ? __popcll(m[1]) : 0) + (w > 2 ? __popcll(m[2]) : 0);
                taken = rank < kAggFrontRows;
            }
        }
        if (!live || taken) return;
        for (int f = gl * VEC; f < H; f += G * VEC) finish(r, f, seg_sum(beg, end, f), end - beg);
        return;
    }

    // ---- front workgroup: the hub rows of one 256-row slice --------------------------------------------------------
    const int64_t sbase = (int64_t)blockIdx.x * 256;
    s_rp[t] = clampi(sbase + t);
    if (t == 0) s_rp[256] = clampi(sbase + 256);
    __syncthreads();
    const bool is_long = s_rp[t + 1] - s_rp[t] > kAggLong;
    const unsigned long long bal = __ballot(is_long);
    if ((t & 63) == 0) s_mask[t >> 6] = bal;
    __syncthreads();
    const int c0 = __popcll(s_mask[0]), c1 = __popcll(s_mask[1]), c2 = __popcll(s_mask[2]), c3 = __popcll(s_mask[3]);
    const int total = c0 + c1 + c2 + c3;
    if (total == 0 || total > kAggDense) return;
    if (is_long) {
        const int w = t >> 6;
        const int rank = (w > 0 ? c0 : 0) + (w > 1 ? c1 : 0) + (w > 2 ? c2 : 0) + __popcll(s_mask[w] & ((1ull << (t & 63)) - 1ull));
        if (rank < kAggFrontRows) s_list[rank] = t;
    }
    __syncthreads();
    const int nl = total < kAggFrontRows ? total : kAggFrontRows;
    for (int f0 = 0; f0 < H; f0 += G * VEC) {
        const int f = f0 + gl * VEC;
        const bool livef = f < H;
        if (f0 > 0) __syncthreads();  // (the previous slice's partial sums have been read)
        for (int i = 0; i < nl; ++i) {
            const int tl = s_list[i];
            const int b2 = s_rp[tl], e2 = s_rp[tl + 1];
            const int seg = (e2 - b2 + rows_wg - 1) / rows_wg;
            const int sb = b2 + gq * seg < e2 ? b2 + gq * seg : e2, se = sb + seg < e2 ? sb + seg : e2;
            part[(i * rows_wg + gq) * G + gl] = livef ? seg_sum(sb, se, f) : V(0.f);
        }
        __syncthreads();
        if (gq < nl && livef) {  // lane group i adds up row i's partial sums in group order
            const int tl = s_list[gq];
            V acc = part[(gq * rows_wg) * G + gl];
            for (int g = 1; g < rows_wg; ++g) acc += part[(gq * rows_wg + g) * G + gl];
            finish(sbase + tl, f, acc, s_rp[tl + 1] - s_rp[tl]);
        }
    }
}

int launch_aggregate(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, const float* x,
                     int64_t ldx, int32_t H, int32_t mean, int32_t mode, float eps, float* out,
                     int64_t ldo, hipStream_t st) {
    if (n_nodes == 0) return GNF_OK;
    const bool vec4 = (H % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int per_row = vec4 ? H / 4 : H;  // lanes a row can use
    int G = 1;
    while (G < per_row && G < 64) G <<= 1;
    const int64_t n_front = (n_nodes + 255) / 256;
    const int64_t blocks = (n_nodes * G + 255) / 256 + n_front;
    if (vec4)
        hipLaunchKernelGGL(k_aggregate<4>, dim3((unsigned)blocks), dim3(256), 0, st, rowptr, col, n_nodes, x,
                           ldx, H, mean, mode, eps, out, ldo, G, (int)n_front);
    else
        hipLaunchKernelGGL(k_aggregate<1>, dim3((unsigned)blocks), dim3(256), 0, st, rowptr, col, n_nodes, x,
                           ldx, H, mean, mode, eps, out, ldo, G, (int)n_front);
    GNF_LAUNCH_CHECK("k_aggregate");
    return GNF_OK;
}

// ------------------------------------------------------------------------------------------------
// Linear layer Y[N,O] = act(X[N,I] @ W[I,O] + b): 64x64 output tile, BK = 16, 4x4 micro-tile per
// thread, operands staged through LDS; k accumulated in order with fmaf.
// ------------------------------------------------------------------------------------------------
static constexpr int LT = 64;
static constexpr int LK = 16;

__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ X, int64_t ldx,
                                                const float* __restrict__ W,
                                                const float* __restrict__ bias,
                                                float* __restrict__ Y, int64_t ldy, int64_t n_rows,
                                                int I, int O, int act, float alpha, int apply_act) {
    __shared__ float Xs[LK][LT + 4];  // transposed: Xs[k][row]
    __shared__ float Ws[LK][LT + 4];  // Ws[k][col]
    const int64_t row0 = (int64_t)blockIdx.y * LT;
    const int col0 = blockIdx.x * LT;
    const int tx = threadIdx.x & 15;   // column group
    const int ty = threadIdx.x >> 4;   // row group
    float acc[4][4] = {};
    for (int k0 = 0; k0 < I; k0 += LK) {
        // stage X tile (64 rows x 16 k): 1024 elements, 4 per thread
        for (int i = threadIdx.x; i < LT * LK; i += 256) {
            const int r = i >> 4, k = i & 15;
            const int64_t gr = row0 + r;
            Xs[k][r] = (gr < n_rows && k0 + k < I) ? X[gr * ldx + k0 + k] : 0.f;
        }
        for (int i = threadIdx.x; i < LT * LK; i += 256) {
            const int k = i >> 6, c = i & 63;
            Ws[k][c] = (k0 + k < I && col0 + c < O) ? W[(int64_t)(k0 + k) * O + col0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LK; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Xs[k][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Ws[k][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t gr = row0 + ty * 4 + i;
        if (gr >= n_rows) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gc = col0 + tx * 4 + j;
            if (gc >= O) continue;
            float v = acc[i][j] + bias[gc];
            if (apply_act) v = (act == GNF_ACT_RELU) ? fmaxf(v, 0.f) : fmaxf(v, alpha * v);
            Y[gr * ldy + gc] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Coupling epilogue: x_upd <- x_upd*exp(s)+t  or  (x_upd-t)*exp(-s); one fp64 partial of sum(s)
// per workgroup (fixed in-block order -> bitwise reproducible).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) tot += sh[i];
    }
    return tot;  // valid on thread 0
}

// V4: every row is whole, aligned float4s (H, ld, every pointer): a thread takes four consecutive features of a row per step -
// a quarter of the loads, index divisions and stores (with the last layer's slabs a scalar element is 2 x n_slab + 2 loads:
// 9.8 us per half-step on the 2 718 x 100 wide_fc batch)
template <bool V4>
__global__ __launch_bounds__(256) void k_coupling(const float* __restrict__ s,
                                                  const float* __restrict__ t, int64_t lds_,
                                                  float* __restrict__ x_upd, int64_t ld,
                                                  int64_t n_nodes, int H, int inverse,
                                                  double* __restrict__ partials,
                                                  const float* __restrict__ xres, const SlabSrc sl) {
    typedef float vf __attribute__((ext_vector_type(V4 ? 4 : 1)));
    constexpr int W = V4 ? 4 : 1;
    __shared__ double sh[4];
    const int Hw = H / W;
    const int64_t total = n_nodes * Hw;
    const int64_t per_block = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t beg = (int64_t)blockIdx.x * per_block;
    int64_t end = beg + per_block;
    if (end > total) end = total;
    double local = 0.0;
    auto ldv = [](const float* p) { return *reinterpret_cast<const vf*>(p); };
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
        const int64_t r = i / Hw;
        const int f = (int)(i - r * Hw) * W;
        vf sv, tv;
        if (sl.n_slab) {  // s, t = the last layer's bias + its partial products (launch_linear_big_fused), added in slab order
            sv = ldv(sl.bias_s + f), tv = ldv(sl.bias_t + f);
#pragma unroll 8
            for (int k = 0; k < sl.n_slab; ++k) sv += ldv(s + k * sl.stride + r * lds_ + f), tv += ldv(t + k * sl.stride + r * lds_ + f);
            if (sl.s_out) *reinterpret_cast<vf*>(sl.s_out + r * lds_ + f) = sv, *reinterpret_cast<vf*>(sl.t_out + r * lds_ + f) = tv;
        } else {
            sv = ldv(s + r * lds_ + f), tv = ldv(t + r * lds_ + f);
        }
        if (xres) {  // attention block with residual (gnn.py:547-548): both nets' outputs += x_cond
            const vf xr = ldv(xres + r * ld + f);
            sv += xr;
            tv += xr;
        }
        const vf xv = ldv(x_upd + r * ld + f);
        vf y;
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const float se = V4 ? sv[q] : sv[0], te = V4 ? tv[q] : tv[0], xe = V4 ? xv[q] : xv[0];
            const float ye = inverse ? (xe - te) * expf(-se) : xe * expf(se) + te;
            if (V4) y[q] = ye; else y[0] = ye;
            local += (double)se;
        }
        *reinterpret_cast<vf*>(x_upd + r * ld + f) = y;
    }
    const double tot = block_sum_256(local, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// The same update for a flow with batch-norm bijectors (forward): the half this launch writes is what the NEXT bijector
// normalises, so its column sums ride along as one [H][2] fp64 partial row per workgroup (sum x, sum x^2 of the new values,
// k_bn_stats' arithmetic: fixed order) - a pass over [N, H] of its own took 22.9 us per half-step on the data driver's batch.
// A workgroup owns `rows` consecutive nodes; thread (rs, c) walks column c of rows rs, rs + lanes, ...  (H <= 256)
__global__ __launch_bounds__(256) void k_coupling_rows(const float* __restrict__ s, const float* __restrict__ t, int64_t lds_,
                                                       float* __restrict__ x_upd, int64_t ld, int64_t n_nodes, int H, int rows,
                                                       double* __restrict__ partials, const float* __restrict__ xres,
                                                       double* __restrict__ bn_part, const SlabSrc sl) {
    __shared__ double sh[4];
    __shared__ double cs[2][256];
    const int tid = threadIdx.x;
    const int lanes = 256 / H, c = tid % H, rs = tid / H;
    const int64_t r0 = (int64_t)blockIdx.x * rows;
    const int64_t r1 = r0 + rows < n_nodes ? r0 + rows : n_nodes;
    double local = 0.0, sx = 0.0, sq = 0.0;
    if (rs < lanes)
        for (int64_t r = r0 + rs; r < r1; r += 4 * lanes) {  // four rows in flight per thread
            float sv[4], tv[4], xv[4], xr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + u * lanes < r1 ? r + u * lanes : r;
                sv[u] = s[rr * lds_ + c], tv[u] = t[rr * lds_ + c], xv[u] = x_upd[rr * ld + c];
                xr[u] = xres ? xres[rr * ld + c] : 0.f;
            }
            if (sl.n_slab) {  // (see k_coupling)
                const float bs = sl.bias_s[c], bt = sl.bias_t[c];
#pragma unroll
                for (int u = 0; u < 4; ++u) sv[u] += bs, tv[u] += bt;
#pragma unroll 4
                for (int k = 1; k < sl.n_slab; ++k) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t rr = r + u * lanes < r1 ? r + u * lanes : r;
                        sv[u] += s[k * sl.stride + rr * lds_ + c], tv[u] += t[k * sl.stride + rr * lds_ + c];
                    }
                }
                if (sl.s_out) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r + u * lanes < r1) sl.s_out[(r + u * lanes) * lds_ + c] = sv[u], sl.t_out[(r + u * lanes) * lds_ + c] = tv[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r + u * lanes >= r1) continue;
                const float se = sv[u] + xr[u], te = tv[u] + xr[u];   // (attention block with residual, gnn.py:547-548)
                const float y = xv[u] * expf(se) + te;
                x_upd[(r + u * lanes) * ld + c] = y;
                local += (double)se;
                sx += (double)y;
                sq += (double)y * (double)y;
            }
        }
    cs[0][tid] = sx;
    cs[1][tid] = sq;
    const double tot = block_sum_256(local, sh);   // (its barriers also publish cs)
    if (tid == 0) partials[blockIdx.x] = tot;
    __syncthreads();
    if (tid < H) {
        double a0 = 0.0, a1 = 0.0;
        for (int k = 0; k < lanes; ++k) a0 += cs[0][k * H + tid], a1 += cs[1][k * H + tid];
        bn_part[((int64_t)blockIdx.x * H + tid) * 2 + 0] = a0;
        bn_part[((int64_t)blockIdx.x * H + tid) * 2 + 1] = a1;
    }
}

// Kernel D: per-workgroup fp64 partials of sum(z^2).
__global__ __launch_bounds__(256) void k_gauss(const float* __restrict__ z, int64_t n_nodes, int D,
                                               int64_t ld, double* __restrict__ partials) {
    __shared__ double sh[4];
    const int64_t total = n_nodes * D;
    const int64_t per_block = ((total + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const int64_t beg = (int64_t)blockIdx.x * per_block;
    int64_t end = beg + per_block;
    if (end > total) end = total;
    double local = 0.0;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
        const int64_t r = i / D;
        const int f = (int)(i - r * D);
        const double v = (double)z[r * ld + f];
        local += v * v;
    }
    const double tot = block_sum_256(local, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

int launch_gauss_partials(const float* z, int64_t n_nodes, int32_t D, int64_t ld, double* partials,
                          int32_t* n_partials, hipStream_t st) {
    const int64_t total = n_nodes * D;
    int64_t blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > kMaxGaussBlocks) blocks = kMaxGaussBlocks;
    hipLaunchKernelGGL(k_gauss, dim3((unsigned)blocks), dim3(256), 0, st, z, n_nodes, D, ld, partials);
    GNF_LAUNCH_CHECK("k_gauss");
    *n_partials = (int32_t)blocks;
    return GNF_OK;
}

// Final fixed-order fp64 reduction of the per-workgroup partials (single workgroup).
__global__ __launch_bounds__(kFinalizeBlock) void k_finalize(const double* __restrict__ a, int64_t na,
                                                            const double* __restrict__ b, int64_t nb,
                                                            double* __restrict__ out,
                                                            int accumulate_a, int write_b) {
    __shared__ double sh[4];
    // a thread's partials are requested eight at a time and both lists before the first block reduction (a plain
    // "load, add" loop is one memory round trip per element: 11 in a row for the 2720 log-det partials of a config-2
    // flow); the order of a thread's additions is unchanged
    auto strided_sum = [&](const double* __restrict__ p, int64_t cnt) {
        double acc = 0.0;
        for (int64_t i0 = threadIdx.x; i0 < cnt; i0 += 8 * kFinalizeBlock) {
            double r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t i = i0 + (int64_t)k * kFinalizeBlock;
                r[k] = p[i < cnt ? i : 0];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + (int64_t)k * kFinalizeBlock < cnt) acc += r[k];
        }
        return acc;
    };
    const double la = a != nullptr ? strided_sum(a, na) : 0.0;
    const double lb = write_b ? strided_sum(b, nb) : 0.0;
    const double ta = block_sum_256(la, sh);
    __syncthreads();
    const double tb = block_sum_256(lb, sh);
    if (threadIdx.x == 0) {
        if (a != nullptr) out[0] = accumulate_a ? out[0] + ta : ta;
        if (write_b) out[1] = tb;
    }
}

__global__ __launch_bounds__(256) void k_copy_rows2(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                    int64_t ldd, int64_t n, int W) {
    const int64_t total = n * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / W;
        const int f = (int)(i - r * W);
        dst[r * ldd + f] = src[r * lds_ + f];
    }
}

int launch_copy_rows(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t n, int32_t W, hipStream_t st) {
    if (n == 0) return GNF_OK;
    int64_t blocks = (n * W + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_copy_rows2, dim3((unsigned)blocks), dim3(256), 0, st, src, lds_, dst, ldd, n, W);
    GNF_LAUNCH_CHECK("k_copy_rows");
    return GNF_OK;
}

int launch_finalize(const double* a, int64_t na, const double* b, int64_t nb, double* out,
                    int accumulate_a, int write_b, hipStream_t st) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(kFinalizeBlock), 0, st, a, na, b, nb, out,
                       accumulate_a, write_b);
    GNF_LAUNCH_CHECK("k_finalize");
    return GNF_OK;
}

// ------------------------------------------------------------------------------------------------
// One half-step through the scratch buffer.
// scratch layout (floats): h0 [N, in0] | s-net bufA, bufB [N, Lmax] | t-net bufA, bufB | s [N, H] | t [N, H]
// ------------------------------------------------------------------------------------------------
// nets[0..nj): MLPs with identical layer shapes (nj = 2: the s- and t-net of a half-step), each with its
// own input h0[q], ping-pong buffers bufA[q] / bufB[q] (row stride ldbuf) and output outp[q].
// keep (nullable): keep[q * GNF_MAX_LAYERS + j] = where layer j's output of net q goes instead of the ping-pong buffers
// (j < K - 1: the training forward's stash of the hidden activations, row stride ldbuf)
static int run_mlps(const GnfMlp* const* nets, int nj, const float* const* h0, int64_t ld0, float* const* bufA,
                    float* const* bufB, int64_t ldbuf, float* const* outp, int64_t ldout, int64_t n,
                    const GnfGnnSpec& g, hipStream_t st, float* const* keep = nullptr, SlabSrc* fuse = nullptr) {
    const GnfMlp* m = nets[0];
    const float* in[2] = {h0[0], h0[nj - 1]};
    int64_t ldin = ld0;
    if (fuse) fuse->n_slab = 0;
    for (int j = 0; j < m->num_layers; ++j) {
        const bool last = (j == m->num_layers - 1);
        if (fuse && nj == 2 && j + 2 == m->num_layers && linear_big_fused_last(m, j + 1) &&
            (int64_t)linear_big_fused_slabs(m->dims[j + 1]) * m->dims[j + 2] <= ldbuf) {
            // the wide layer in front of a thin last one takes it along (gnf_linear_big.hip): the partial products land in
            // the ping-pong buffer this layer's output would have taken, the caller's coupling kernel adds them up
            float* yq[2];
            float* sq[2];
            for (int q = 0; q < 2; ++q) yq[q] = keep ? keep[q * GNF_MAX_LAYERS + j] : nullptr, sq[q] = (j & 1) ? bufB[q] : bufA[q];
            int32_t ns = 0;
            const int rc = launch_linear_big_fused(nets, 2, j, in, ldin, yq, ldbuf, sq, &ns, n, g.activation, g.alpha, st);
            if (rc == GNF_OK) {
                fuse->n_slab = ns;
                fuse->stride = n * (int64_t)m->dims[j + 2];
                fuse->bias_s = nets[0]->b[j + 1], fuse->bias_t = nets[1]->b[j + 1];
                fuse->s_out = sq[0], fuse->t_out = sq[1];  // (callers read: the slabs' bases)
                return GNF_OK;
            }
            if (rc != 1) return rc;
        }
        float* dst[2];
        for (int q = 0; q < 2; ++q) {
            const int qq = q < nj ? q : nj - 1;
            dst[q] = last ? outp[qq] : (keep ? keep[qq * GNF_MAX_LAYERS + j] : ((j & 1) ? bufB[qq] : bufA[qq]));
        }
        const int64_t lddst = last ? ldout : ldbuf;
        const int I = m->dims[j], O = m->dims[j + 1];
        if ((I >= 32 && O >= 32) || (last && j >= 1 && I >= 512)) {
            // matrix-core layers run through the generic GEMM tile (gnf_train.hip); a thin output behind a wide layer
            // (2048 -> 100 on a few hundred nodes: 16 tiles x 64 k-steps) is split over the reduction there, its
            // slabs in the ping-pong buffer the LAST layer leaves free
            const float* xin[2];
            const float *wq[2], *bq[2];
            float* yq[2];
            float* sk[2] = {nullptr, nullptr};
            for (int q = 0; q < nj; ++q) {
                xin[q] = in[q], wq[q] = nets[q]->W[j], bq[q] = nets[q]->b[j], yq[q] = dst[q];
                if (last && j >= 1) sk[q] = (j & 1) ? bufB[q] : bufA[q];
            }
            // wide layers of a pair of nets with packed weights: the large-batch kernel's inner loop (gnf_linear_big.hip)
            int rc = launch_linear_big(nets, nj, j, xin, ldin, yq, lddst, n, g.activation, g.alpha, last ? 0 : 1, st);
            if (rc == 1) rc = launch_linear_short(nets, nj, j, xin, ldin, yq, lddst, n, g.activation, g.alpha, last ? 0 : 1, st);
            if (rc == 1)
                rc = launch_linear_splitk(xin, ldin, wq, bq, yq, lddst, nj, n, I, O, g.activation, g.alpha, last ? 0 : 1, sk,
                                          (size_t)n * (size_t)ldbuf, st);
            if (rc) return rc;
            in[0] = dst[0];
            in[1] = dst[1];
            ldin = lddst;
            continue;
        }
        {
            dim3 grid((O + LT - 1) / LT, (unsigned)((n + LT - 1) / LT));
            for (int q = 0; q < nj; ++q) {
                hipLaunchKernelGGL(k_linear, grid, dim3(256), 0, st, in[q], ldin, nets[q]->W[j], nets[q]->b[j],
                                   dst[q], lddst, n, I, O, g.activation, g.alpha, last ? 0 : 1);
                GNF_LAUNCH_CHECK("k_linear");
            }
        }
        in[0] = dst[0];
        in[1] = dst[1];
        ldin = lddst;
    }
    return GNF_OK;
}

__global__ __launch_bounds__(256) void k_add_rows(float* __restrict__ dst, int64_t ldd,
                                                  const float* __restrict__ src, int64_t lds_, int64_t n, int W) {
    const int64_t total = n * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / W;
        const int f = (int)(i - r * W);
        dst[r * ldd + f] += src[r * lds_ + f];
    }
}

int launch_add_rows(float* dst, int64_t ldd, const float* src, int64_t lds_, int64_t n, int32_t W, hipStream_t st) {
    if (n == 0) return GNF_OK;
    int64_t blocks = (n * W + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_add_rows, dim3((unsigned)blocks), dim3(256), 0, st, dst, ldd, src, lds_, n, W);
    GNF_LAUNCH_CHECK("k_add_rows");
    return GNF_OK;
}

// One wave per row, lanes along the features (three passes over a row that stays in L1/L2: mean, centred variance,
// normalise - the two-pass variance is what tf.nn.moments computes).
__global__ __launch_bounds__(256) void k_layer_norm(const LnArgs a) {
    const LnJob j = a.job[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int W = a.W;
    const float inv_w = 1.f / (float)W;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < a.n; r += (int64_t)gridDim.x * 4) {
        const float* in = j.in + r * a.ldin;
        const float* xr = a.xres ? a.xres + r * a.ldx : nullptr;
        float sum = 0.f;
        for (int f = lane; f < W; f += 64) sum += in[f] + (xr ? xr[f] : 0.f);
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float mean = sum * inv_w;
        float sq = 0.f;
        for (int f = lane; f < W; f += 64) {
            const float d = in[f] + (xr ? xr[f] : 0.f) - mean;
            sq = fmaf(d, d, sq);
        }
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        const float rstd = 1.f / sqrtf(sq * inv_w + GNF_LN_EPS);
        for (int f = lane; f < W; f += 64) {
            const float u = in[f] + (xr ? xr[f] : 0.f);
            j.y[r * a.ldy + f] = (u - mean) * rstd * j.gamma[f] + j.beta[f];
            if (j.u) j.u[r * a.ldin + f] = u;
        }
    }
}

int launch_layer_norm(const LnArgs& a, int nets, hipStream_t st) {
    if (a.n == 0) return GNF_OK;
    int64_t blocks = (a.n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_layer_norm, dim3((unsigned)blocks, (unsigned)nets), dim3(256), 0, st, a);
    GNF_LAUNCH_CHECK("k_layer_norm");
    return GNF_OK;
}

static int lmax_of(const GnfMlp* m) {
    int lmax = 1;
    for (int j = 1; j < m->num_layers; ++j) lmax = lmax > m->dims[j] ? lmax : m->dims[j];
    return lmax;
}

// One GNN module call: scratch = h0 [N, in0] | bufA [N, Lmax] | bufB [N, Lmax] | (s, t unused) | attention region
int launch_gnn_layered(const int32_t* rowptr, const int32_t* col, int64_t n, const float* x,
                       int64_t ldx, int32_t H, const GnfGnnSpec& g, const GnfMlp* mlp, float* out,
                       int64_t ldo, float* scratch, hipStream_t st) {
    const int in0 = mlp->dims[0];
    const int lmax = lmax_of(mlp);
    float* h0 = scratch;
    float* bufA = h0 + n * in0;
    float* bufB = bufA + n * lmax;
    int rc;
    if (mlp->attn) {
        float* attn_scratch = scratch + (size_t)n * (size_t)(in0 + kLayeredActBufs * lmax + 2 * H);
        const GnfAttn* at[1] = {mlp->attn};
        float* h0s[1] = {h0};
        rc = launch_attn_front(rowptr, col, n, x, ldx, H, at, 1, in0, attn_scratch, h0s, st);
    } else {
        rc = launch_aggregate(rowptr, col, n, x, ldx, H, g.agg == GNF_AGG_MEAN,
                              g.combine == GNF_COMBINE_CONCAT ? 1 : 0, g.epsilon, h0, in0, st);
    }
    if (rc) return rc;
    {
        const float* h0c = h0;
        rc = run_mlps(&mlp, 1, &h0c, in0, &bufA, &bufB, lmax, &out, ldo, n, g, st);
    }
    if (rc) return rc;
    if (mlp->attn && mlp->attn->layer_norm) {  // residual add folded into the normalisation's first read
        LnArgs a;
        memset(&a, 0, sizeof(a));
        a.job[0] = LnJob{out, out, nullptr, mlp->attn->ln_gamma, mlp->attn->ln_beta};
        a.ldin = a.ldy = ldo;
        a.xres = mlp->attn->residual ? x : nullptr;
        a.ldx = ldx;
        a.n = n;
        a.W = mlp->dims[mlp->num_layers];
        return launch_layer_norm(a, 1, st);
    }
    if (mlp->attn && mlp->attn->residual) return launch_add_rows(out, ldo, x, ldx, n, H, st);
    return GNF_OK;
}

// s, t [N, H] (dense) of a half-step whose blocks end in snt.LayerNorm: normalised in place, both nets in one launch
int launch_half_layer_norm(const HalfStep& hs, float* sbuf, float* tbuf, const float* xres, hipStream_t st) {
    LnArgs a;
    memset(&a, 0, sizeof(a));
    a.job[0] = LnJob{sbuf, sbuf, nullptr, hs.s_net->attn->ln_gamma, hs.s_net->attn->ln_beta};
    a.job[1] = LnJob{tbuf, tbuf, nullptr, hs.t_net->attn->ln_gamma, hs.t_net->attn->ln_beta};
    a.ldin = a.ldy = hs.H;
    a.xres = xres;
    a.ldx = hs.ld;
    a.n = hs.n_nodes;
    a.W = hs.H;
    return launch_layer_norm(a, 2, st);
}

// scratch layout (floats): h0 [N,in0] | bufA,bufB (s) | bufA,bufB (t) | s [N,H] | t [N,H] | attention region: qkv x2, h0_s, h0_t
int launch_half_layered(const HalfStep& hs, float* scratch, hipStream_t st) {
    const int64_t n = hs.n_nodes;
    *hs.n_partials = 0;
    if (n == 0) return GNF_OK;
    const int H = hs.H;
    const int in0 = hs.s_net->dims[0];
    int lmax = lmax_of(hs.s_net);
    const int lt = lmax_of(hs.t_net);
    lmax = lmax > lt ? lmax : lt;
    float* h0 = scratch;
    float* bufA[2] = {h0 + n * in0, h0 + n * in0 + 2 * n * lmax};   // per net: A | B
    float* bufB[2] = {bufA[0] + n * lmax, bufA[1] + n * lmax};
    float* sbuf = bufA[0] + n * lmax * kLayeredActBufs;
    float* tbuf = sbuf + n * H;
    // training forward with the MLP-row stash in its layered mode (layered_stash_mode, gnf_train.hip): the layer-0 rows of a
    // message-passing net, every hidden activation and s, t are written straight into the half-step's slot
    float* keep[2 * GNF_MAX_LAYERS];
    const bool stash = hs.mlp_stash != nullptr && hs.direction == GNF_FORWARD;
    if (stash) {
        const MlpStashLayout L = mlp_stash_layout(hs.s_net, n, H);
        if (!hs.s_net->attn) h0 = hs.mlp_stash + L.h0;
        for (int q = 0; q < 2; ++q)
            for (int j = 0; j + 1 < hs.s_net->num_layers; ++j)
                keep[q * GNF_MAX_LAYERS + j] = hs.mlp_stash + L.act + ((size_t)q * (hs.s_net->num_layers - 1) + j) * L.act_each;
        sbuf = hs.mlp_stash + L.st;
        tbuf = sbuf + L.st_each;
    }
    const float* h0s = h0;
    const float* h0t = h0;
    SlabSrc fuse;
    memset(&fuse, 0, sizeof(fuse));
    int rc;
    if (hs.s_net->attn) {
        float* h0_pair[2];
        rc = launch_attn_pair(hs, scratch, h0_pair, st);
        h0s = h0_pair[0];
        h0t = h0_pair[1];
    } else {
        rc = launch_aggregate(hs.rowptr, hs.col, n, hs.x_cond, hs.ld, H, hs.gnn.agg == GNF_AGG_MEAN,
                              hs.gnn.combine == GNF_COMBINE_CONCAT ? 1 : 0, hs.gnn.epsilon, h0, in0, st);
    }
    if (rc) return rc;
    {
        const GnfMlp* nets[2] = {hs.s_net, hs.t_net};
        const float* h0p[2] = {h0s, h0t};
        float* outs[2] = {sbuf, tbuf};
        const bool ln = hs.s_net->attn && hs.s_net->attn->layer_norm;  // (normalises finished rows in place)
        rc = run_mlps(nets, 2, h0p, in0, bufA, bufB, lmax, outs, H, n, hs.gnn, st, stash ? keep : nullptr, ln ? nullptr : &fuse);
    }
    if (rc) return rc;
    const bool res = hs.s_net->attn && hs.s_net->attn->residual;
    if (fuse.n_slab) {
        const float *ss = fuse.s_out, *ts = fuse.t_out;
        fuse.s_out = stash ? sbuf : nullptr, fuse.t_out = stash ? tbuf : nullptr;
        return launch_coupling(ss, ts, hs, res ? hs.x_cond : nullptr, st, &fuse);
    }
    if (hs.s_net->attn && hs.s_net->attn->layer_norm) {
        rc = launch_half_layer_norm(hs, sbuf, tbuf, res ? hs.x_cond : nullptr, st);
        if (rc) return rc;
        return launch_coupling(sbuf, tbuf, hs, nullptr, st);
    }
    return launch_coupling(sbuf, tbuf, hs, res ? hs.x_cond : nullptr, st);
}

// Attention front-end of BOTH nets of a half-step into the scratch's attention region; returns the two
// h0 pointers ([N, in0] each).
int launch_attn_pair(const HalfStep& hs, float* scratch, float** h0_pair, hipStream_t st) {
    const int64_t n = hs.n_nodes;
    const int in0 = hs.s_net->dims[0];
    int lmax = lmax_of(hs.s_net);
    const int lt = lmax_of(hs.t_net);
    lmax = lmax > lt ? lmax : lt;
    float* region = hs.attn_region ? hs.attn_region
                                   : scratch + (size_t)n * (size_t)(in0 + kLayeredActBufs * lmax + 2 * hs.H);
    const GnfAttn* a0 = hs.s_net->attn;
    const size_t P = 2 * (size_t)a0->num_heads * a0->kq_dim + a0->v_dim;
    h0_pair[0] = region + 2 * (size_t)n * P;
    h0_pair[1] = h0_pair[0] + (size_t)n * in0;
    const GnfAttn* at[2] = {hs.s_net->attn, hs.t_net->attn};
    // a stash slot also keeps the attended values and the softmax statistics (attn_scratch_floats' layout)
    const size_t NV = (size_t)a0->num_heads * a0->v_dim;
    float* agg0 = h0_pair[1] + (size_t)n * in0;
    float* mz0 = agg0 + 2 * (size_t)n * NV;
    float* agg_out[2] = {agg0, agg0 + (size_t)n * NV};
    float* mz_out[2] = {mz0, mz0 + (size_t)n * 3 * a0->num_heads};
    const bool keep = hs.attn_region != nullptr;
    return launch_attn_front(hs.rowptr, hs.col, n, hs.x_cond, hs.ld, hs.H, at, 2, in0, region, h0_pair, st, hs.n_edges,
                             /*need_qkv=*/keep, hs.attn_packed, keep ? agg_out : nullptr, keep ? mz_out : nullptr);
}

int launch_coupling(const float* sbuf, const float* tbuf, const HalfStep& hs, const float* xres, hipStream_t st, const SlabSrc* slabs) {
    const int64_t n = hs.n_nodes;
    const int H = hs.H;
    SlabSrc sl;
    memset(&sl, 0, sizeof(sl));
    if (slabs) sl = *slabs;
    if (hs.bn_part && hs.n_bn && hs.direction == GNF_FORWARD && H <= 256 && n > 0) {
        // (16 rows per workgroup: (n + 15) / 16 partial rows - the caller sized both partial buffers for exactly that)
        const int rows = 16;
        const int64_t blocks = (n + rows - 1) / rows;
        hipLaunchKernelGGL(k_coupling_rows, dim3((unsigned)blocks), dim3(256), 0, st, sbuf, tbuf, (int64_t)H, hs.x_upd, hs.ld, n, H,
                           rows, hs.partials, xres, hs.bn_part, sl);
        GNF_LAUNCH_CHECK("k_coupling_rows");
        *hs.n_partials = (int32_t)blocks;
        *hs.n_bn = (int32_t)blocks;
        return GNF_OK;
    }
    int64_t blocks = (n * H + 256 * 4 - 1) / (256 * 4);
    const int64_t cap = coupling_blocks_max(n);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    uintptr_t al = reinterpret_cast<uintptr_t>(sbuf) | reinterpret_cast<uintptr_t>(tbuf) | reinterpret_cast<uintptr_t>(hs.x_upd) |
                   reinterpret_cast<uintptr_t>(xres) | reinterpret_cast<uintptr_t>(sl.bias_s) | reinterpret_cast<uintptr_t>(sl.bias_t) |
                   reinterpret_cast<uintptr_t>(sl.s_out) | reinterpret_cast<uintptr_t>(sl.t_out);
    const bool v4 = (H & 3) == 0 && (hs.ld & 3) == 0 && (sl.stride & 3) == 0 && (al & 15) == 0;
    if (v4)
        hipLaunchKernelGGL(k_coupling<true>, dim3((unsigned)blocks), dim3(256), 0, st, sbuf, tbuf, (int64_t)H,
                           hs.x_upd, hs.ld, n, H, hs.direction == GNF_INVERSE ? 1 : 0, hs.partials, xres, sl);
    else
        hipLaunchKernelGGL(k_coupling<false>, dim3((unsigned)blocks), dim3(256), 0, st, sbuf, tbuf, (int64_t)H,
                           hs.x_upd, hs.ld, n, H, hs.direction == GNF_INVERSE ? 1 : 0, hs.partials, xres, sl);
    GNF_LAUNCH_CHECK("k_coupling");
    *hs.n_partials = (int32_t)blocks;
    return GNF_OK;
}

}  // namespace gnf
