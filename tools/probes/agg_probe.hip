// Developer probe (not part of the library): variants of kernel A (CSR segmented reduce of neighbour rows) on a
// CSR dumped by tools/dump_csr.py.  Every variant must reproduce variant 0 (the shipped kernel's arithmetic: adds in
// edge order) bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/agg_probe.hip -o /tmp/agg_probe
//   /tmp/agg_probe /tmp/ego128.csr 128 256
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float V4 __attribute__((ext_vector_type(4)));

__device__ inline int64_t remap_block(int64_t nwg, int64_t bid) {
    const int64_t xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
}

// ---- variant 0: the shipped kernel (VEC = 4 instance) -------------------------------------------------------------
__global__ __launch_bounds__(256) void k_v0(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                            int64_t n_nodes, const float* __restrict__ x, int64_t ldx, int H, int mean,
                                            float eps, float* __restrict__ out, int64_t ldo, int G) {
    const int64_t blk = remap_block(gridDim.x, blockIdx.x);
    const int64_t gid = (blk * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (gid >= n_nodes) return;
    const int64_t r = gid;
    const int beg = rowptr[r], end = rowptr[r + 1];
    const float cnt = (float)((end - beg) > 1 ? (end - beg) : 1);
    for (int f = gl * 4; f < H; f += G * 4) {
        V4 acc = V4(0.f);
        int e = beg;
        for (; e + 8 <= end; e += 8) {
            int ci[8];
            V4 vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ci[q] = col[e + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += vv[q];
        }
        if (e < end) {
            int ci[7];
            V4 vv[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) ci[q] = col[e + q < end ? e + q : end - 1];
#pragma unroll
            for (int q = 0; q < 7; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (e + q < end) acc += vv[q];
        }
        if (mean) acc = acc / cnt;
        const V4 xs = *reinterpret_cast<const V4*>(x + r * ldx + f);
        *reinterpret_cast<V4*>(out + r * ldo + f) = eps * xs + acc;
    }
}

// ---- variant 1/2: rows of more than LONG edges are worked off by the whole workgroup afterwards: every lane group
// fetches other neighbour rows of the chunk into LDS, then group 0 adds them up IN EDGE ORDER (same bits).
// COOP: the short rows' col entries come from one load per group (lane j holds col[beg + j]) and a shuffle.
template <int LONG, int CHUNK, bool COOP>
__global__ __launch_bounds__(256) void k_v2(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                            int64_t n_nodes, const float* __restrict__ x, int64_t ldx, int H, int mean,
                                            float eps, float* __restrict__ out, int64_t ldo, int G) {
    extern __shared__ float stage[];  // [CHUNK][G * 4]
    __shared__ int s_long[256];
    __shared__ int s_nlong;
    const int64_t blk = remap_block(gridDim.x, blockIdx.x);
    const int rows_wg = 256 / G;
    const int gq = threadIdx.x / G;
    const int gl = threadIdx.x & (G - 1);
    const int64_t r = blk * rows_wg + gq;
    if (threadIdx.x == 0) s_nlong = 0;
    __syncthreads();
    int beg = 0, end = 0;
    if (r < n_nodes) {
        beg = rowptr[r];
        end = rowptr[r + 1];
    }
    const int deg = end - beg;
    if (deg > LONG) {
        if (gl == 0) s_long[atomicAdd(&s_nlong, 1)] = gq;
    } else if (r < n_nodes) {
        const float cnt = (float)(deg > 1 ? deg : 1);
        if (COOP && G >= 8) {
            for (int f = gl * 4; f < H; f += G * 4) {
                V4 acc = V4(0.f);
                for (int e = beg; e < end; e += G) {  // G >= 8 edges per col load
                    const int mine = col[e + gl < end ? e + gl : end - 1];
                    const int m = end - e < G ? end - e : G;
                    for (int q0 = 0; q0 < m; q0 += 8) {
                        int ci[8];
                        V4 vv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) ci[q] = __shfl(mine, q0 + q, G);
#pragma unroll
                        for (int q = 0; q < 8; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (q0 + q < m) acc += vv[q];
                    }
                }
                if (mean) acc = acc / cnt;
                const V4 xs = *reinterpret_cast<const V4*>(x + r * ldx + f);
                *reinterpret_cast<V4*>(out + r * ldo + f) = eps * xs + acc;
            }
        } else {
            for (int f = gl * 4; f < H; f += G * 4) {
                V4 acc = V4(0.f);
                int e = beg;
                for (; e + 8 <= end; e += 8) {
                    int ci[8];
                    V4 vv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) ci[q] = col[e + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc += vv[q];
                }
                if (e < end) {
                    int ci[7];
                    V4 vv[7];
#pragma unroll
                    for (int q = 0; q < 7; ++q) ci[q] = col[e + q < end ? e + q : end - 1];
#pragma unroll
                    for (int q = 0; q < 7; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
                    for (int q = 0; q < 7; ++q)
                        if (e + q < end) acc += vv[q];
                }
                if (mean) acc = acc / cnt;
                const V4 xs = *reinterpret_cast<const V4*>(x + r * ldx + f);
                *reinterpret_cast<V4*>(out + r * ldo + f) = eps * xs + acc;
            }
        }
    }
    __syncthreads();
    const int nl = s_nlong;
    if (nl == 0) return;
    const int ngroups = rows_wg < CHUNK ? rows_wg : CHUNK;  // groups that fetch
    constexpr int PER = 8;                                   // rows a group fetches per chunk at most
    for (int i = 0; i < nl; ++i) {
        const int64_t rr = blk * rows_wg + s_long[i];
        const int b2 = rowptr[rr], e2 = rowptr[rr + 1];
        const float cnt = (float)(e2 - b2);
        for (int f0 = 0; f0 < H; f0 += G * 4) {
            const int f = f0 + gl * 4;
            V4 acc = V4(0.f);
            for (int e0 = b2; e0 < e2; e0 += CHUNK) {
                const int m = e2 - e0 < CHUNK ? e2 - e0 : CHUNK;
                if (gq < ngroups && f < H) {
                    int ci[PER];
                    V4 vv[PER];
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int j = gq + q * ngroups;
                        ci[q] = col[e0 + (j < m ? j : m - 1)];
                    }
#pragma unroll
                    for (int q = 0; q < PER; ++q)
                        if (q * ngroups < CHUNK) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int j = gq + q * ngroups;
                        if (j < m) *reinterpret_cast<V4*>(stage + (j * G + gl) * 4) = vv[q];
                    }
                }
                __syncthreads();
                if (gq == 0 && f < H) {
                    for (int j = 0; j < m; ++j) acc += *reinterpret_cast<const V4*>(stage + (j * G + gl) * 4);
                }
                __syncthreads();
            }
            if (gq == 0 && f < H) {
                if (mean) acc = acc / cnt;
                const V4 xs = *reinterpret_cast<const V4*>(x + rr * ldx + f);
                *reinterpret_cast<V4*>(out + rr * ldo + f) = eps * xs + acc;
            }
        }
    }
}


// ---- variant 3: hub rows on workgroups of their own, dispatched FIRST.  The grid is [n_front | regular]: front
// workgroup i scans the rowptr slice of rows [256 i, 256 i + 256), and works off the first KMAX rows of more than LONG
// edges there with all of its lane groups (every group fetches other neighbour rows of a chunk into LDS, group 0 adds
// them up in edge order: same bits as the sequential loop); the regular workgroups read the same slice, so they know
// which rows are taken and skip them.  A slice with more than DENSE long rows (complete graphs) is left alone.
template <int LONG, int KMAX, int DENSE, bool PIPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_v3(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                            int64_t n_nodes, const float* __restrict__ x, int64_t ldx, int H, int mean,
                                            float eps, float* __restrict__ out, int64_t ldo, int G, int n_front) {
    __shared__ __attribute__((aligned(16))) float stage[4096];  // [CHUNK][G * 4], CHUNK * G = 1024
    __shared__ int s_rp[257];
    __shared__ unsigned long long s_mask[4];
    __shared__ int s_list[KMAX];
    __shared__ int s_col[256];
    const bool front = (int)blockIdx.x < n_front;
    const int rows_wg = 256 / G;
    const int gq = threadIdx.x / G, gl = threadIdx.x & (G - 1);
    const int t = threadIdx.x;
    int64_t blk = 0, sbase;
    if (front) {
        sbase = (int64_t)blockIdx.x * 256;
    } else {
        blk = remap_block(gridDim.x - n_front, blockIdx.x - n_front);
        sbase = (blk * rows_wg) & ~(int64_t)255;
    }
    {
        const int64_t rr = sbase + t;
        s_rp[t] = rowptr[rr < n_nodes ? rr : n_nodes];
        if (t == 0) s_rp[256] = rowptr[sbase + 256 < n_nodes ? sbase + 256 : n_nodes];
    }
    __syncthreads();
    const bool is_long = s_rp[t + 1] - s_rp[t] > LONG;
    const unsigned long long bal = __ballot(is_long);
    if ((t & 63) == 0) s_mask[t >> 6] = bal;
    __syncthreads();
    const int c0 = __popcll(s_mask[0]), c1 = __popcll(s_mask[1]), c2 = __popcll(s_mask[2]), c3 = __popcll(s_mask[3]);
    const int total = c0 + c1 + c2 + c3;
    const bool dense = total > DENSE;
    auto rank_of = [&](int tl) {
        const int w = tl >> 6, l = tl & 63;
        const int before = (w > 0 ? c0 : 0) + (w > 1 ? c1 : 0) + (w > 2 ? c2 : 0);
        return before + __popcll(s_mask[w] & ((1ull << l) - 1ull));
    };
    if (!front) {
        const int64_t r = blk * rows_wg + gq;
        if (r >= n_nodes) return;
        const int tl = (int)(r - sbase);
        const int beg = s_rp[tl], end = s_rp[tl + 1];
        if (!dense && end - beg > LONG && rank_of(tl) < KMAX) return;  // a front workgroup has it
        const float cnt = (float)((end - beg) > 1 ? (end - beg) : 1);
        for (int f = gl * 4; f < H; f += G * 4) {
            V4 acc = V4(0.f);
            int e = beg;
            for (; e + 8 <= end; e += 8) {
                int ci[8];
                V4 vv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) ci[q] = col[e + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += vv[q];
            }
            if (e < end) {
                int ci[7];
                V4 vv[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) ci[q] = col[e + q < end ? e + q : end - 1];
#pragma unroll
                for (int q = 0; q < 7; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (e + q < end) acc += vv[q];
            }
            if (mean) acc = acc / cnt;
            const V4 xs = *reinterpret_cast<const V4*>(x + r * ldx + f);
            *reinterpret_cast<V4*>(out + r * ldo + f) = eps * xs + acc;
        }
        return;
    }
    // ---- front workgroup ----
    if (dense || total == 0) return;
    if (is_long) {
        const int rk = rank_of(t);
        if (rk < KMAX) s_list[rk] = t;
    }
    __syncthreads();
    const int nl = total < KMAX ? total : KMAX;
    const int chunk = 1024 / G < 32 ? 1024 / G : 32;
    const int ngroups = rows_wg < chunk ? rows_wg : chunk;
    const int per = chunk / ngroups;  // <= 8
    for (int i = 0; i < nl; ++i) {
        const int tl = s_list[i];
        const int64_t rr = sbase + tl;
        const int b2 = s_rp[tl], e2 = s_rp[tl + 1];
        const float cnt = (float)(e2 - b2);
        for (int f0 = 0; f0 < H; f0 += G * 4) {
            const int f = f0 + gl * 4;
            const bool livef = f < H;
            V4 acc = V4(0.f);
            for (int es = b2; es < e2; es += 256) {
                const int ms = e2 - es < 256 ? e2 - es : 256;
                __syncthreads();  // (s_col / stage of the previous pass are done with)
                if (t < ms) s_col[t] = col[es + t];
                __syncthreads();
                V4 vv[8];
                auto fetch = [&](int e0) {
                    const int m = ms - e0 < chunk ? ms - e0 : chunk;
                    if (gq < ngroups && livef) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (q < per) {
                                const int j = gq + q * ngroups;
                                vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)s_col[e0 + (j < m ? j : m - 1)] * ldx + f);
                            }
                    }
                };
                auto put = [&](int e0) {
                    const int m = ms - e0 < chunk ? ms - e0 : chunk;
                    if (gq < ngroups && livef) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (q < per) {
                                const int j = gq + q * ngroups;
                                if (j < m) *reinterpret_cast<V4*>(stage + (j * G + gl) * 4) = vv[q];
                            }
                    }
                };
                if (PIPE) fetch(0);
                for (int e0 = 0; e0 < ms; e0 += chunk) {
                    const int m = ms - e0 < chunk ? ms - e0 : chunk;
                    if (!PIPE) fetch(e0);
                    put(e0);
                    __syncthreads();
                    if (PIPE && e0 + chunk < ms) fetch(e0 + chunk);
                    if (gq == 0 && livef)
                        for (int j = 0; j < m; ++j) acc += *reinterpret_cast<const V4*>(stage + (j * G + gl) * 4);
                    __syncthreads();
                }
            }
            if (gq == 0 && livef) {
                if (mean) acc = acc / cnt;
                const V4 xs = *reinterpret_cast<const V4*>(x + rr * ldx + f);
                *reinterpret_cast<V4*>(out + rr * ldo + f) = eps * xs + acc;
            }
        }
    }
}

// ---- variant 4: as variant 3, but a hub row's edges are split into contiguous segments over the front workgroup's lane
// groups (each adds its segment up in edge order, 8 rows in flight) and the per-group partial sums are added up in
// group order: a fixed, deterministic order, but not the sequential one (the row differs from variant 0 in rounding).
template <int LONG, int KMAX, int DENSE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_v4(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_nodes, const float* __restrict__ x,
    int64_t ldx, int H, int mean, float eps, float* __restrict__ out, int64_t ldo, int G, int n_front) {
    __shared__ __attribute__((aligned(16))) float part[KMAX * 1024];  // [KMAX][256 / G groups][G * 4]
    __shared__ int s_rp[257];
    __shared__ unsigned long long s_mask[4];
    __shared__ int s_list[KMAX];
    const bool front = (int)blockIdx.x < n_front;
    const int rows_wg = 256 / G;
    const int gq = threadIdx.x / G, gl = threadIdx.x & (G - 1);
    const int t = threadIdx.x;
    int64_t blk = 0, sbase;
    if (front) {
        sbase = (int64_t)blockIdx.x * 256;
    } else {
        blk = remap_block(gridDim.x - n_front, blockIdx.x - n_front);
        sbase = (blk * rows_wg) & ~(int64_t)255;
    }
    {
        const int64_t rr = sbase + t;
        s_rp[t] = rowptr[rr < n_nodes ? rr : n_nodes];
        if (t == 0) s_rp[256] = rowptr[sbase + 256 < n_nodes ? sbase + 256 : n_nodes];
    }
    __syncthreads();
    const bool is_long = s_rp[t + 1] - s_rp[t] > LONG;
    const unsigned long long bal = __ballot(is_long);
    if ((t & 63) == 0) s_mask[t >> 6] = bal;
    __syncthreads();
    const int c0 = __popcll(s_mask[0]), c1 = __popcll(s_mask[1]), c2 = __popcll(s_mask[2]), c3 = __popcll(s_mask[3]);
    const int total = c0 + c1 + c2 + c3;
    const bool dense = total > DENSE;
    auto rank_of = [&](int tl) {
        const int w = tl >> 6, l = tl & 63;
        const int before = (w > 0 ? c0 : 0) + (w > 1 ? c1 : 0) + (w > 2 ? c2 : 0);
        return before + __popcll(s_mask[w] & ((1ull << l) - 1ull));
    };
    // a lane group's sequential sum over edges [beg, end) of feature slice f: 8 neighbour rows in flight, adds in edge order
    auto seg_sum = [&](int beg, int end, int f) {
        V4 acc = V4(0.f);
        int e = beg;
        for (; e + 8 <= end; e += 8) {
            int ci[8];
            V4 vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ci[q] = col[e + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += vv[q];
        }
        if (e < end) {
            int ci[7];
            V4 vv[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) ci[q] = col[e + q < end ? e + q : end - 1];
#pragma unroll
            for (int q = 0; q < 7; ++q) vv[q] = *reinterpret_cast<const V4*>(x + (int64_t)ci[q] * ldx + f);
#pragma unroll
            for (int q = 0; q < 7; ++q)
                if (e + q < end) acc += vv[q];
        }
        return acc;
    };
    if (!front) {
        const int64_t r = blk * rows_wg + gq;
        if (r >= n_nodes) return;
        const int tl = (int)(r - sbase);
        const int beg = s_rp[tl], end = s_rp[tl + 1];
        if (!dense && end - beg > LONG && rank_of(tl) < KMAX) return;  // a front workgroup has it
        const float cnt = (float)((end - beg) > 1 ? (end - beg) : 1);
        for (int f = gl * 4; f < H; f += G * 4) {
            V4 acc = seg_sum(beg, end, f);
            if (mean) acc = acc / cnt;
            const V4 xs = *reinterpret_cast<const V4*>(x + r * ldx + f);
            *reinterpret_cast<V4*>(out + r * ldo + f) = eps * xs + acc;
        }
        return;
    }
    // ---- front workgroup ----
    if (dense || total == 0) return;
    if (is_long) {
        const int rk = rank_of(t);
        if (rk < KMAX) s_list[rk] = t;
    }
    __syncthreads();
    const int nl = total < KMAX ? total : KMAX;
    for (int f0 = 0; f0 < H; f0 += G * 4) {
        const int f = f0 + gl * 4;
        const bool livef = f < H;
        if (f0 > 0) __syncthreads();
        for (int i = 0; i < nl; ++i) {
            const int tl = s_list[i];
            const int b2 = s_rp[tl], e2 = s_rp[tl + 1];
            const int seg = (e2 - b2 + rows_wg - 1) / rows_wg;
            const int sb = b2 + gq * seg < e2 ? b2 + gq * seg : e2, se = sb + seg < e2 ? sb + seg : e2;
            V4 acc = V4(0.f);
            if (livef) acc = seg_sum(sb, se, f);
            *reinterpret_cast<V4*>(part + ((i * rows_wg + gq) * G + gl) * 4) = acc;
        }
        __syncthreads();
        if (gq < nl && livef) {  // group i adds up row i's partial sums in group order
            const int tl = s_list[gq];
            const int64_t rr = sbase + tl;
            V4 acc = *reinterpret_cast<const V4*>(part + ((gq * rows_wg + 0) * G + gl) * 4);
            for (int g = 1; g < rows_wg; ++g) acc += *reinterpret_cast<const V4*>(part + ((gq * rows_wg + g) * G + gl) * 4);
            if (mean) acc = acc / (float)(s_rp[tl + 1] - s_rp[tl]);
            const V4 xs = *reinterpret_cast<const V4*>(x + rr * ldx + f);
            *reinterpret_cast<V4*>(out + rr * ldo + f) = eps * xs + acc;
        }
    }
}

struct Csr {
    int64_t n, e;
    std::vector<int32_t> rowptr, col;
};

static constexpr int NVAR = 19;
int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: agg_probe file.csr H ldx [trunc_deg]\n");
        return 2;
    }
    const int H = atoi(argv[2]);
    const int64_t ldx = atoll(argv[3]);
    const int trunc = argc > 4 ? atoi(argv[4]) : 0;  // > 0: keep only the first trunc edges of every row (what-if: no hub rows)
    Csr c;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int64_t hdr[2];
    if (fread(hdr, 8, 2, f) != 2) return 3;
    c.n = hdr[0];
    c.e = hdr[1];
    c.rowptr.resize(c.n + 1);
    c.col.resize(c.e);
    if (fread(c.rowptr.data(), 4, c.n + 1, f) != (size_t)c.n + 1) return 3;
    if (fread(c.col.data(), 4, c.e, f) != (size_t)c.e) return 3;
    fclose(f);
    if (trunc > 0) {
        std::vector<int32_t> rp(c.n + 1, 0), cl;
        for (int64_t r = 0; r < c.n; ++r) {
            int d = c.rowptr[r + 1] - c.rowptr[r];
            if (d > trunc) d = trunc;
            for (int q = 0; q < d; ++q) cl.push_back(c.col[c.rowptr[r] + q]);
            rp[r + 1] = (int32_t)cl.size();
        }
        c.rowptr = rp;
        c.col = cl;
        c.e = (int64_t)cl.size();
    }
    int32_t *d_rp, *d_col;
    float *d_x, *d_ref, *d_out;
    CK(hipMalloc(&d_rp, 4 * (c.n + 1)));
    CK(hipMalloc(&d_col, 4 * c.e));
    CK(hipMalloc(&d_x, 4 * c.n * ldx));
    CK(hipMalloc(&d_ref, 4 * c.n * H));
    CK(hipMalloc(&d_out, 4 * c.n * H));
    CK(hipMemcpy(d_rp, c.rowptr.data(), 4 * (c.n + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, c.col.data(), 4 * c.e, hipMemcpyHostToDevice));
    std::vector<float> hx(c.n * ldx);
    uint32_t s = 12345;
    for (auto& v : hx) {
        s = s * 1664525u + 1013904223u;
        v = ((int)(s >> 8) % 20001 - 10000) * 1e-4f;
    }
    CK(hipMemcpy(d_x, hx.data(), 4 * c.n * ldx, hipMemcpyHostToDevice));
    int G = 1;
    while (G < H / 4 && G < 64) G <<= 1;
    const int rows_wg = 256 / G;
    const unsigned blocks0 = (unsigned)((c.n * G + 255) / 256);
    const unsigned blocks2 = (unsigned)((c.n + rows_wg - 1) / rows_wg);
    const unsigned nfront = (unsigned)((c.n + 255) / 256);
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    std::vector<float> href(c.n * H), hout(c.n * H);
    const double alg = 8.0 * c.n * H + 4.0 * c.e + 4.0 * c.n, gath = 4.0 * c.e * H + 8.0 * c.n * H + 4.0 * c.e + 4.0 * c.n;
    printf("# %s N=%lld E=%lld H=%d ldx=%lld G=%d trunc=%d\n", argv[1], (long long)c.n, (long long)c.e, H, (long long)ldx, G, trunc);
    for (int mean = 0; mean < 2; ++mean) {
        for (int v = 0; v < NVAR; ++v) {
            if (v >= 1 && v <= 13 && v != 9 && v != 12) continue;
            auto launch = [&](float* o) {
                switch (v) {
                    case 0: hipLaunchKernelGGL(k_v0, dim3(blocks0), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                    case 1: hipLaunchKernelGGL((k_v2<32, 32, false>), dim3(blocks2), dim3(256), 32 * G * 16, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                    case 2: hipLaunchKernelGGL((k_v2<16, 32, false>), dim3(blocks2), dim3(256), 32 * G * 16, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                    case 3: hipLaunchKernelGGL((k_v2<32, 64, false>), dim3(blocks2), dim3(256), 64 * G * 16, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                    case 4: hipLaunchKernelGGL((k_v2<32, 32, true>), dim3(blocks2), dim3(256), 32 * G * 16, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                    case 5: hipLaunchKernelGGL((k_v2<1000000, 32, false>), dim3(blocks2), dim3(256), 32 * G * 16, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                    case 6: hipLaunchKernelGGL((k_v2<1000000, 32, true>), dim3(blocks2), dim3(256), 32 * G * 16, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                    case 8: hipLaunchKernelGGL((k_v3<32, 4, 16, false>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 9: hipLaunchKernelGGL((k_v3<32, 4, 16, true>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 10: hipLaunchKernelGGL((k_v3<16, 8, 24, true>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 11: hipLaunchKernelGGL((k_v3<64, 4, 16, true>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 12: hipLaunchKernelGGL((k_v3<1000000, 4, 16, true>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 13: hipLaunchKernelGGL((k_v3<24, 6, 16, true>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 14: hipLaunchKernelGGL((k_v4<32, 4, 16>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 15: hipLaunchKernelGGL((k_v4<16, 4, 16>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 16: hipLaunchKernelGGL((k_v4<24, 4, 16>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 17: hipLaunchKernelGGL((k_v4<48, 4, 16>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 18: hipLaunchKernelGGL((k_v4<32, 2, 16>), dim3(blocks2 + nfront), dim3(256), 0, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G, (int)nfront); break;
                    case 7: hipLaunchKernelGGL((k_v2<64, 32, false>), dim3(blocks2), dim3(256), 32 * G * 16, 0, d_rp, d_col, c.n, d_x, ldx, H, mean, 1.0f, o, (int64_t)H, G); break;
                }
            };
            static const char* names[] = {"v0 shipped", "v2 LONG32 CHUNK32", "v2 LONG16 CHUNK32", "v2 LONG32 CHUNK64", "v2 LONG32 CHUNK32 coop-col",
                                          "v2 no-long (structure only)", "v2 no-long coop-col", "v2 LONG64 CHUNK32",
                                          "v3 front LONG32 K4", "v3 front LONG32 K4 pipelined", "v3 front LONG16 K8 pipelined", "v3 front LONG64 K4 pipelined",
                                          "v3 no-long (structure only)", "v3 front LONG24 K6 pipelined",
                                          "v4 partials LONG32 K4", "v4 partials LONG16 K4", "v4 partials LONG24 K4", "v4 partials LONG48 K4", "v4 partials LONG32 K2"};
            CK(hipMemset(v == 0 ? d_ref : d_out, 0xff, 4 * c.n * H));
            launch(v == 0 ? d_ref : d_out);
            CK(hipDeviceSynchronize());
            CK(hipGetLastError());
            size_t bad = 0;
            float maxd = 0.f;
            if (v == 0)
                CK(hipMemcpy(href.data(), d_ref, 4 * c.n * H, hipMemcpyDeviceToHost));
            else {
                CK(hipMemcpy(hout.data(), d_out, 4 * c.n * H, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hout.size(); ++i) {
                    bad += memcmp(&hout[i], &href[i], 4) != 0;
                    const float d = fabsf(hout[i] - href[i]);
                    maxd = d > maxd || d != d ? d : maxd;
                }
            }
            for (int i = 0; i < 10; ++i) launch(d_out);
            float best = 1e30f, sum = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(ea, 0));
                for (int i = 0; i < 100; ++i) launch(d_out);
                CK(hipEventRecord(eb, 0));
                CK(hipEventSynchronize(eb));
                float ms;
                CK(hipEventElapsedTime(&ms, ea, eb));
                best = ms < best ? ms : best;
                sum += ms;
            }
            const double us = best * 10.0;
            printf("%s %-30s %8.2f us (mean %.2f)  algorithmic %7.1f GB/s  gathered %7.1f GB/s  mismatches %zu max|d| %.2e\n", mean ? "mean" : "sum ", names[v], us,
                   sum * 2.0, alg / us / 1e3, gath / us / 1e3, bad, maxd);
            fflush(stdout);
        }
    }
    return 0;
}
