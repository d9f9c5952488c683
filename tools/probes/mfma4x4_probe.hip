// Developer probe: (1) semantics and rate of v_mfma_f32_4x4x1_16b_f32 used as a [4 x K] x [K x 64] tile
// (A lane l supplies X[l%4][k], B lane l supplies W[k][l], D reg i of lane l = out[i][l]);
// (2) time of the 5-layer (32-256-256-256-256-32) two-net MLP phase of a 12-row node tile built on it:
// 8 waves (4 per net), wave w of a net owns output columns [64w, 64w+64), weights streamed from global memory
// in [k/4][col tile][lane][4] fragment order through a register ring, activations in LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_sem(const float* X, const float* W, float* out, int K) {  // X [4][K], W [K][64], out [4][64]
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(X[(l & 3) * K + k], W[k * 64 + l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[i * 64 + l] = acc[i];
}

__global__ void k_rate(float* out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

#ifndef SUBT
#define SUBT 3  // 4-row sub-tiles per node tile
#endif
constexpr int R = 8, PF = R - 1;  // register ring of weight fragments
constexpr int LS = 260;

struct Layers {
    const float* w[2][5];
    int nk4[5];   // k-groups of 4
    int ont[5];   // 64-column tiles
};

__global__ __launch_bounds__(512) void k_mlp(Layers L, float* sink) {
    __shared__ __attribute__((aligned(16))) float act[2][2][16][LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int net = wave >> 2, wl = wave & 3;
    for (int i = tid; i < 2 * 2 * 16 * LS; i += 512) (&act[0][0][0][0])[i] = 0.001f * (i & 255);
    __syncthreads();
    int pp = 0;
    float keep = 0.f;
    for (int j = 0; j < 5; ++j) {
        const int nk = L.nk4[j], ont = L.ont[j];
        if (wl < ont) {
            const float* wb = L.w[net][j];
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wb), 0, nk * ont * 1024, 0x00020000);
            const int voff = lane * 16;
            f32x4 acc[SUBT];
#pragma unroll
            for (int s = 0; s < SUBT; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 ring[R];
#pragma unroll
            for (int u = 0; u < PF; ++u)
                ring[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((u < nk ? u : nk - 1) * ont + wl) * 1024, 0));
            const float* arow = &act[net][pp][lane & 3][0];
            for (int k0 = 0; k0 < nk; k0 += R) {
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const int kk = k0 + u;
                    const int kn = kk + PF < nk ? kk + PF : nk - 1;
                    ring[(u + PF) % R] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (kn * ont + wl) * 1024, 0));
                    f32x4 a[SUBT];
#pragma unroll
                    for (int s = 0; s < SUBT; ++s) a[s] = *reinterpret_cast<const f32x4*>(arow + 4 * s * LS + 4 * (kk < nk ? kk : nk - 1));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int s = 0; s < SUBT; ++s) acc[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s][q], ring[u][q], acc[s], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int s = 0; s < SUBT; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[s][i];
                    act[net][pp ^ 1][4 * s + i][64 * wl + lane] = fmaxf(v, 0.2f * v) * 1e-3f;
                    keep += v;
                }
        }
        pp ^= 1;
        __syncthreads();
    }
    if (keep == 123.456f) sink[tid] = keep;
}

int main() {
    // ---- semantics
    const int K = 37;
    std::vector<float> hx(4 * K), hw(K * 64), ho(256), ref(256, 0.f);
    for (auto& v : hx) v = (rand() % 1000) * 1e-3f - 0.5f;
    for (auto& v : hw) v = (rand() % 1000) * 1e-3f - 0.5f;
    for (int i = 0; i < 4; ++i) for (int c = 0; c < 64; ++c) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(hx[i * K + k], hw[k * 64 + c], s); ref[i * 64 + c] = s; }
    float *dx, *dw, *dout;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&dout, 1 << 22);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dx, dw, dout, K);
    hipMemcpy(ho.data(), dout, 1024, hipMemcpyDeviceToHost);
    double maxerr = 0; int bitexact = 1;
    for (int i = 0; i < 256; ++i) { maxerr = fmax(maxerr, fabs(ho[i] - ref[i])); if (ho[i] != ref[i]) bitexact = 0; }
    printf("4x4x1 as [4xK]x[Kx64]: max err vs fmaf chain %.3e, bitwise equal: %d\n", maxerr, bitexact);
    // ---- rate
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 8192;
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(threads), 0, 0, dout, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_rate, dim3(256), dim3(threads), 0, 0, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mf = 4.0 * iters * (threads / 256);
        printf("4x4x1 rate, %d threads x 256 blocks: %.2f ns per MFMA per SIMD (8 cycles at 2.2 GHz = 3.64 ns); %.1f TFLOP/s chip\n",
               threads, ms * 1e6 / mf, 256.0 * 4 * mf * 512 / (ms * 1e-3) / 1e12);
    }
    // ---- MLP phase
    Layers L;
    const int dims[6] = {32, 256, 256, 256, 256, 32};
    size_t tot = 0;
    for (int j = 0; j < 5; ++j) { L.nk4[j] = dims[j] / 4; L.ont[j] = (dims[j + 1] + 63) / 64; tot += (size_t)L.nk4[j] * L.ont[j] * 256; }
    float* wbuf; hipMalloc(&wbuf, tot * 4 * 2 * 16);  // 16 different half-steps worth, so that weights are not L2-hot across launches
    std::vector<float> hwb(tot * 2 * 16, 0.01f);
    hipMemcpy(wbuf, hwb.data(), hwb.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {170, 227, 256}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int hs) {
            size_t off = (size_t)hs * tot * 2;
            for (int n = 0; n < 2; ++n) for (int j = 0; j < 5; ++j) { L.w[n][j] = wbuf + off; off += (size_t)L.nk4[j] * L.ont[j] * 256; }
            hipLaunchKernelGGL(k_mlp, dim3(blocks), dim3(512), 0, 0, L, dout);
        };
        for (int hs = 0; hs < 16; ++hs) launch(hs);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) for (int hs = 0; hs < 16; ++hs) launch(hs);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("MLP phase, SUBT=%d (%d rows/tile), %d blocks: %.2f us per launch (incl. ~2 us launch+LDS init)\n", SUBT, 4 * SUBT, blocks, ms * 1e3 / 80);
    }
    return 0;
}
